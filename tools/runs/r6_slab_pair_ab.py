#!/usr/bin/env python3
"""Round 6 A/B: slab form (H > 128) of the MLP on a 4-letter alphabet with the first layer gathered from pre-summed PAIR rows in LDS
(mlp_pair = 1) against one row per position (mlp_pair = 2: pairs for H <= 128 only), interleaved; kernel time from fx_debug_time_score.
The two differ by one float32 rounding per pair of positions (both are held to the oracle in tests/test_gpu_forms.py).
-> profiles/r6_slab_pair_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("mlp H=200 L=14 N=1e5 (DynaPPO member)", 14, "UGCA", 200, 1, 100_000), ("mlp H=200 L=14 N=5e4", 14, "UGCA", 200, 1, 50_000),
         ("mlp H=200 L=14 N=1e6", 14, "UGCA", 200, 1, 1_000_000), ("mlp H=200 L=8 N=1e5", 8, "TGCA", 200, 1, 100_000),
         ("mlp H=256 L=8 N=1e5", 8, "TGCA", 256, 1, 100_000), ("3 x mlp H=200 L=14 N=1e5", 14, "UGCA", 200, 3, 100_000),
         ("mlp H=200 L=14 N=2000", 14, "UGCA", 200, 1, 2_000), ("mlp H=160 L=14 N=1e5", 14, "UGCA", 160, 1, 100_000)]
for name, L, alpha, H, M, n in CASES:
    mods = build_members("mlp", L, alpha, M, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    opts = (2, 1)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in opts}
    res = {q: [] for q in opts}
    for rep in range(3):
        for q in opts:
            eng.set_option("mlp_pair", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    a, b = planes[2][:, :n].cpu().numpy(), planes[1][:, :n].cpu().numpy()
    med = {q: float(np.median(res[q])) for q in opts}
    fr = {q: roofline_block("mlp", L, len(alpha), H, 0, 0, M, n, med[q] * 1e-3, "k")["frac"] for q in opts}
    print(f"{name:40s} plain rows {med[2]:8.2f} us ({fr[2]:.3f})  pair rows {med[1]:8.2f} us ({fr[1]:.3f})  ({(med[1] / med[2] - 1) * 100:+.1f} %)   "
          f"max |diff| {np.abs(a - b).max():.2e} (max |score| {np.abs(a).max():.2f})", flush=True)
eng.set_option("mlp_pair", 1)
