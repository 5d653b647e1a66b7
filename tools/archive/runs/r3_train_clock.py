"""Is the training step clock-bound?  Time Ensemble.train cold vs right after 200 ms of MFMA-bound scoring launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get(0)
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
big = synth.random_sequence_bytes(1_000_000, 8, "TGCA", 1)
d_in = torch.from_numpy(big).cuda()
d_pl = torch.empty((3, 1_000_000), dtype=torch.float32, device="cuda")
nat = [m.native() for m in ens.models]
ens.train(seqs, y)
for rep in range(3):
    time.sleep(0.5)
    t0 = time.perf_counter(); ens.train(seqs, y); cold = time.perf_counter() - t0
    eng.time_score_planes(nat, d_in.data_ptr(), 1_000_000, 8, ens.models[0]._lut, d_pl.data_ptr(), 1_000_000, 100)   # ~190 ms of MFMA work
    t0 = time.perf_counter(); ens.train(seqs, y); hot = time.perf_counter() - t0
    t0 = time.perf_counter(); ens.train(seqs, y); hot2 = time.perf_counter() - t0
    print(f"train after 0.5 s idle {cold * 1e3:.2f} ms; right after 190 ms of scoring {hot * 1e3:.2f} ms; again {hot2 * 1e3:.2f} ms", flush=True)
