"""Rows per slice of the training step (engine option train_rows): 8 (what a 256-row mini-batch gets) against 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()

def run(tag, make, L, alpha, n):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 3))
    y = np.random.default_rng(0).random(n)
    out = []
    for rows in (8, 16, 0):
        eng.set_option("train_rows", rows)
        model = make()
        model.train(seqs, y); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); model.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out.append(min(ts) * 1e3)
    eng.set_option("train_rows", 0)
    print(f"{tag} n={n}: 8 rows per slice {out[0]:.2f} ms, 16 rows {out[1]:.2f} ms, auto {out[2]:.2f} ms", flush=True)

run("Ensemble 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1000)
run("CNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 1000)
run("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 1000)
