"""Fit time (one member, 500 sequences, 20 epochs of two 256-row steps) across shapes: looks for cliffs where a layout stops fitting LDS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from flexs_amd import synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
rows = [("cnn", L, "TGCA") for L in (8, 10, 14, 20, 24, 30, 50, 100, 200)] + [("cnn", L, s_utils.AAS) for L in (10, 20, 50, 90, 150, 237, 260, 261, 300, 500)] + \
       [("mlp", L, a) for L, a in ((14, "UGCA"), (50, "UGCA"), (90, s_utils.AAS), (237, s_utils.AAS))] + [("ge", L, s_utils.AAS) for L in (14, 90, 237)]
for kind, L, alpha in rows:
    m = bm.CNN(L, 32, 100, alpha, seed=0) if kind == "cnn" else (bm.MLP(L, 100, alpha, seed=0) if kind == "mlp" else bm.GlobalEpistasisModel(L, 100, alpha, seed=0))
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(500, L, alpha, 3)); y = np.random.default_rng(0).random(500)
    m.train(seqs, y, seed=5); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    macs = synth.algorithmic_macs(kind, L, len(alpha), 100, 32 if kind == "cnn" else 0, 5 if kind == "cnn" else 0)
    tf = 3 * 2.0 * macs * 500 * 20 / min(ts) / 1e12
    print(f"{kind} L={L:3d} A={len(alpha):2d}: {min(ts) * 1e3:7.2f} ms per fit  ({min(ts) / 40 * 1e6:6.1f} us per step, {tf:6.2f} TFLOP/s algorithmic)", flush=True)
