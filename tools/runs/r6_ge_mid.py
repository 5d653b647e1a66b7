#!/usr/bin/env python3
"""Round 6: GlobalEpistasis / MLP at a few hundred to a few thousand sequences: the small-launch form (dense_small = 1: one workgroup per tile,
weights from L2) against the persistent kernel (0: weights in LDS), kernel time.  -> profiles/r6_ge_mid.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, time_launches  # noqa: E402

eng = _native.Engine.get(0)
for kind, L, alpha, H in (("ge", 237, AAS, 100), ("ge", 90, AAS, 100), ("ge", 100, "UGCA", 100), ("ge", 14, "UGCA", 100), ("mlp", 14, "UGCA", 100), ("mlp", 50, "UGCA", 100), ("mlp", 14, "UGCA", 200)):
    mods = build_members(kind, L, alpha, 1, 0, Hx=H)
    for n in (16, 256, 1000, 2000, 4096):
        d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
        stride = (n + 63) // 64 * 64
        planes = {q: torch.zeros((1, stride), dtype=torch.float32, device="cuda") for q in (0, 1)}
        res = {}
        for q in (1, 0):
            eng.set_option("dense_small", q)
            res[q] = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=5.0)[0] * 1e3 for _ in range(3)]))
        same = bool(torch.equal(planes[0][:, :n], planes[1][:, :n]))
        print(f"{kind} L={L:3d} A={len(alpha):2d} H={H} N={n:5d}: small-launch form {res[1]:7.2f} us   persistent kernel {res[0]:7.2f} us   same bits {same}", flush=True)
eng.set_option("dense_small", 1)
