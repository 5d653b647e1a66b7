#!/usr/bin/env python3
"""Round 6 A/B (experiment build -DFX_DENSE_W12, FLEXS_AMD_LIB=.../libflexs_amd_v_w12.so): the PAIR (MLP) / byte-table (GE) dense kernels in
12-wave workgroups (three waves per SIMD: 6 tiles per SIMD go 2-2-2) against 16 (2-2-1-1), interleaved; kernel time from fx_debug_time_score.
-> profiles/r6_dense_w12_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("mlp L=14 N=1e5 (C3)", "mlp", 14, "UGCA", 1, 100_000), ("mlp L=14 N=5e4", "mlp", 14, "UGCA", 1, 50_000), ("mlp L=14 N=73728 (18/WG)", "mlp", 14, "UGCA", 1, 73_728),
         ("mlp L=14 N=2e5", "mlp", 14, "UGCA", 1, 200_000), ("mlp L=14 N=1e6", "mlp", 14, "UGCA", 1, 1_000_000), ("3 x mlp L=14 N=1e5", "mlp", 14, "UGCA", 3, 100_000),
         ("ge L=90 N=1e5", "ge", 90, AAS, 1, 100_000), ("8 x ge L=90 N=1e5 (C4)", "ge", 90, AAS, 8, 100_000), ("mlp L=8 N=1e5", "mlp", 8, "TGCA", 1, 100_000)]
for name, kind, L, alpha, M, n in CASES:
    mods = build_members(kind, L, alpha, M, 0)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    opts = (0, 12)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in opts}
    res = {q: [] for q in opts}
    for rep in range(3):
        for q in opts:
            eng.set_option("dense_waves", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=30.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = bool(torch.equal(planes[0][:, :n], planes[12][:, :n]))
    med = {q: float(np.median(res[q])) for q in opts}
    fr = {q: roofline_block(kind, L, len(alpha), 100, 0, 0, M, n, med[q] * 1e-3, "k")["frac"] for q in opts}
    print(f"{name:28s} 16 waves {med[0]:8.2f} us ({fr[0]:.3f})   12 waves {med[12]:8.2f} us ({fr[12]:.3f})  ({(med[12] / med[0] - 1) * 100:+.1f} %)  same bits {same}", flush=True)
eng.set_option("dense_waves", 0)
