"""A/B of the native training step's knobs: rows per workgroup (train_rows) and LDS-resident workspace (train_lds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils

eng = _native.Engine.get(0)

def run(tag, make, L, alpha, n, configs):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 3))
    y = np.random.default_rng(0).random(n)
    for rows, lds, thr in configs:
        eng.set_option("train_rows", rows); eng.set_option("train_lds", lds); eng.set_option("train_threads", thr)
        model = make()
        model.train(seqs, y)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); model.train(seqs, y); ts.append(time.perf_counter() - t0)
        print(f"{tag} n={n} rows={rows} lds={lds} threads={thr}: {min(ts) * 1e3:.2f} ms", flush=True)
    eng.set_option("train_rows", 0); eng.set_option("train_lds", 2); eng.set_option("train_threads", 0)

cfg = [(0, 2, 0), (8, 2, 1024), (16, 2, 1024), (4, 2, 1024), (8, 2, 512), (16, 2, 512)]
run("Ensemble 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1000, cfg)
run("CNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 1000, cfg)
run("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 1000, cfg)
run("Ensemble 8xGE L=90", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=m) for m in range(8)]), 90, s_utils.AAS, 1000, cfg[:6])
run("CNN L=90 A=20", lambda: bm.CNN(90, 32, 100, s_utils.AAS, seed=0), 90, s_utils.AAS, 1000, [(0, 2, 0), (1, 2, 256), (1, 2, 1024), (2, 2, 1024), (1, 0, 1024)])
run("Ensemble 3xCNN L=237 A=20", lambda: flexs_amd.Ensemble([bm.CNN(237, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 237, s_utils.AAS, 500, [(0, 2, 0), (1, 2, 256), (1, 2, 512)])
