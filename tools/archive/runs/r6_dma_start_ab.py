#!/usr/bin/env python3
"""(ARCHIVED: the cnn_dma_start option this script toggles was measured slower and removed; profiles/r6_dma_start_ab.log)
Round 6 A/B: K1 (unrolled seq_len = 8 form) with the (tiles mod 4) last tiles of a workgroup walked by wave quads (cnn_quad_tail = 1) and with
the staged start (cnn_dma_start = 1: image by direct global -> LDS copies, first tiles start when the conv part has landed) against
one wave per tile throughout / the whole image before anybody starts (0), interleaved on one box; kernel time from fx_debug_time_score, bits compared.  -> profiles/r6_quad_tail_ab.log"""
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
    legs = ((0, 0), (1, 0), (0, 1), (1, 1))            # (cnn_quad_tail, cnn_dma_start); (0, 0) = the kernel of rounds 2-5 (+ the 8-byte row load)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in legs}
    res = {q: [] for q in legs}
    for rep in range(4):
        for q in legs:
            eng.set_option("cnn_quad_tail", q[0]); eng.set_option("cnn_dma_start", q[1])
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(planes[legs[0]][:, :n], planes[q][:, :n])) for q in legs)
    med = {q: float(np.median(res[q])) for q in legs}
    fr = {q: roofline_block("cnn", L, 4, 100, 32, 5, M, n, med[q] * 1e-3, "k")["frac"] for q in legs}
    print(f"{name:34s} " + "  ".join(f"tail={q[0]} dma={q[1]}: {med[q]:8.2f} us ({fr[q]:.3f})" for q in legs) + f"   same bits {same}", flush=True)
eng.set_option("cnn_quad_tail", 1); eng.set_option("cnn_dma_start", 1)
