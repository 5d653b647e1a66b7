#!/usr/bin/env python3
"""Round 6 A/B: K1's work split in whole rounds of four tiles per workgroup (engine option cnn_unit_quant = 4) against the plain cut (1),
interleaved on one box; kernel time from fx_debug_time_score (launches issued from C), bits compared.  -> profiles/r6_unit_quant_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("3xCNN L=8 N=1e5 (headline)", 8, "TGCA", 3, 100_000), ("1xCNN L=8 N=1e5", 8, "TGCA", 1, 100_000), ("3xCNN L=8 N=5e4", 8, "TGCA", 3, 50_000),
         ("3xCNN L=8 N=1e6", 8, "TGCA", 3, 1_000_000), ("8xCNN L=8 N=1e5", 8, "TGCA", 8, 100_000), ("3xCNN L=14 N=1e5", 14, "UGCA", 3, 100_000),
         ("3xCNN L=8 N=3e4", 8, "TGCA", 3, 30_000), ("1xCNN L=14 N=7e4", 14, "UGCA", 1, 70_001)]
for name, L, alpha, M, n in CASES:
    mods = build_members("cnn", L, alpha, M, 0)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in (1, 4)}
    res = {1: [], 4: []}
    for rep in range(4):
        for q in (1, 4):
            eng.set_option("cnn_unit_quant", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = bool(torch.equal(planes[1][:, :n], planes[4][:, :n]))
    a, b = float(np.median(res[1])), float(np.median(res[4]))
    fr = roofline_block("cnn", L, len(alpha), 100, 32, 5, M, n, b * 1e-3, "k")["frac"]
    print(f"{name:30s} plain {a:8.2f} us  quads {b:8.2f} us  ({(b / a - 1) * 100:+.1f} %)  issued frac {fr:.3f}  same bits {same}   runs plain {[round(x, 1) for x in res[1]]} quads {[round(x, 1) for x in res[4]]}", flush=True)
eng.set_option("cnn_unit_quant", 4)
