#!/usr/bin/env python3
"""Round 6 A/B: K1 (unrolled seq_len = 8 form) with the (tiles mod 4) last tiles of a workgroup walked by wave quads (cnn_quad_tail = 1) against
one wave per tile throughout (0), interleaved on one box; kernel time from fx_debug_time_score, bits compared.  -> profiles/r6_quad_tail_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("3xCNN L=8 N=1e5 (headline)", 3, 100_000), ("1xCNN L=8 N=1e5", 1, 100_000), ("3xCNN L=8 N=5e4", 3, 50_000), ("3xCNN L=8 N=2e5", 3, 200_000),
         ("3xCNN L=8 N=1e6", 3, 1_000_000), ("8xCNN L=8 N=1e5", 8, 100_000), ("3xCNN L=8 N=3e4", 3, 30_000), ("1xCNN L=8 N=65536", 1, 65_536),
         ("3xCNN L=8 N=98304 (18 per SIMD)", 3, 98_304), ("2xCNN L=8 N=1e5", 2, 100_000)]
for name, M, n in CASES:
    L, alpha = 8, "TGCA"
    mods = build_members("cnn", L, alpha, M, 0)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in (0, 1)}
    res = {0: [], 1: []}
    for rep in range(4):
        for q in (0, 1):
            eng.set_option("cnn_quad_tail", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = bool(torch.equal(planes[0][:, :n], planes[1][:, :n]))
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    fr = [roofline_block("cnn", L, 4, 100, 32, 5, M, n, t * 1e-3, "k")["frac"] for t in (a, b)]
    print(f"{name:34s} one wave per tile {a:8.2f} us ({fr[0]:.3f})  quad tail {b:8.2f} us ({fr[1]:.3f})  ({(b / a - 1) * 100:+.1f} %)  same bits {same}   runs {[round(x, 1) for x in res[0]]} / {[round(x, 1) for x in res[1]]}", flush=True)
eng.set_option("cnn_quad_tail", 1)
