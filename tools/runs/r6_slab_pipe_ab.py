#!/usr/bin/env python3
"""Round 6 A/B: slab form (H > 128), MLP: the next round's first layer gathered inside this round's last H x H layer
(dense_slab_pipe = 1) against in front of the round (0), interleaved; kernel time from fx_debug_time_score.
-> profiles/r6_slab_pipe_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("mlp H=200 L=14 N=1e5 (DynaPPO member)", "mlp", 14, "UGCA", 200, 1, 100_000), ("mlp H=200 L=14 N=5e4", "mlp", 14, "UGCA", 200, 1, 50_000),
         ("mlp H=200 L=14 N=2e5", "mlp", 14, "UGCA", 200, 1, 200_000), ("mlp H=200 L=14 N=1e6", "mlp", 14, "UGCA", 200, 1, 1_000_000),
         ("mlp H=256 L=14 N=1e5", "mlp", 14, "UGCA", 256, 1, 100_000), ("3 x mlp H=200 L=14 N=1e5", "mlp", 14, "UGCA", 200, 3, 100_000),
         ("mlp H=200 L=8 N=1e5", "mlp", 8, "TGCA", 200, 1, 100_000), ("mlp H=200 L=28 N=1e5", "mlp", 28, "UGCA", 200, 1, 100_000),
         ("mlp H=200 L=14 N=118784 (r=5)", "mlp", 14, "UGCA", 200, 1, 118_784)]
for name, kind, L, alpha, H, M, n in CASES:
    mods = build_members(kind, L, alpha, M, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    opts = (0, 1)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in opts}
    res = {q: [] for q in opts}
    for rep in range(3):
        for q in opts:
            eng.set_option("dense_slab_pipe", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(planes[0][:, :n], planes[q][:, :n])) for q in opts)
    med = {q: float(np.median(res[q])) for q in opts}
    fr = {q: roofline_block(kind, L, len(alpha), H, 0, 0, M, n, med[q] * 1e-3, "k")["frac"] for q in opts}
    print(f"{name:40s} " + "  ".join(f"pipe={q}: {med[q]:8.2f} us ({fr[q]:.3f})" for q in opts) + f"   same bits {same}", flush=True)
eng.set_option("dense_slab_pipe", 1)
