import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
for kind, L in (("mlp", 237), ("mlp", 90), ("ge", 237), ("mlp", 14)):
    alpha = s_utils.AAS if L > 14 else "UGCA"
    for rows in (0, 16, 32, 64):
        eng.set_option("train_rows", rows)
        m = bm.MLP(L, 100, alpha, seed=0) if kind == "mlp" else bm.GlobalEpistasisModel(L, 100, alpha, seed=0)
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(500, L, alpha, 3)); y = np.random.default_rng(0).random(500)
        m.train(seqs, y, seed=5); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); m.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{kind} L={L} train_rows={rows}: {min(ts)*1e3:.2f} ms", flush=True)
eng.set_option("train_rows", 0)
