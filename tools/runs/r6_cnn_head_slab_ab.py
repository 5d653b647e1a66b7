#!/usr/bin/env python3
"""Round 6 A/B: 4-letter CNN with more than 128 hidden units at batch size as conv-only kernel + slab head kernel (cnn_head_slab = 1) against
the fused kernel (0), interleaved; time of the whole scoring call's launches from fx_debug_time_score (both kernels of the two-kernel path).
-> profiles/r6_cnn_head_slab_ab.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("cnn H=200 L=8 N=1e5", 8, "TGCA", 200, 1, 100_000), ("cnn H=200 L=8 N=66e3", 8, "TGCA", 200, 1, 66_000), ("cnn H=200 L=8 N=2e5", 8, "TGCA", 200, 1, 200_000),
         ("cnn H=200 L=8 N=1e6", 8, "TGCA", 200, 1, 1_000_000), ("3 x cnn H=200 L=8 N=1e5", 8, "TGCA", 200, 3, 100_000), ("cnn H=256 L=8 N=1e5", 8, "TGCA", 256, 1, 100_000),
         ("cnn H=200 L=14 N=1e5", 14, "UGCA", 200, 1, 100_000), ("cnn H=160 L=8 N=1e5", 8, "TGCA", 160, 1, 100_000)]
for name, L, alpha, H, M, n in CASES:
    mods = build_members("cnn", L, alpha, M, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    opts = (0, 1)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in opts}
    res = {q: [] for q in opts}
    for rep in range(3):
        for q in opts:
            eng.set_option("cnn_head_slab", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=40.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = bool(torch.equal(planes[0][:, :n], planes[1][:, :n]))
    med = {q: float(np.median(res[q])) for q in opts}
    fr = {q: roofline_block("cnn", L, len(alpha), H, 32, 5, M, n, med[q] * 1e-3, "k")["frac"] for q in opts}
    print(f"{name:28s} fused {med[0]:8.2f} us ({fr[0]:.3f})   conv + slab head {med[1]:8.2f} us ({fr[1]:.3f})  ({(med[1] / med[0] - 1) * 100:+.1f} %)  same bits {same}", flush=True)
eng.set_option("cnn_head_slab", 1)
