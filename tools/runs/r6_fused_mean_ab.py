#!/usr/bin/env python3
"""Round 6 A/B: a mean-only batch step (K1 + the NumPy-order ensemble mean) with the mean taken by the scoring kernel itself
(fuse_mean_batch = 1: the last member to finish a tile averages it) against the mean kernel behind it (0), interleaved (A/B build: FLEXS_AMD_LIB=.../libflexs_amd_ab.so); `steps`
back-to-back steps issued through DistributedEnsemble.launch / finish (bench.py's loop), wall time per step.
-> profiles/r6_fused_mean_ab.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from flexs_amd import distributed as fd  # noqa: E402
from tools.bench_common import build_members  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("3xCNN L=8 N=1e5 (headline)", 8, 3, 100_000), ("2xCNN L=8 N=1e5", 8, 2, 100_000), ("7xCNN L=8 N=1e5", 8, 7, 100_000),
         ("3xCNN L=8 N=3e4", 8, 3, 30_000), ("3xCNN L=14 N=1e5", 14, 3, 100_000), ("3xCNN L=8 N=1e6", 8, 3, 1_000_000)]
for name, L, M, n in CASES:
    alpha = "TGCA" if L == 8 else "UGCA"
    ens = fd.DistributedEnsemble(build_members("cnn", L, alpha, M, 0), mode="sequence")
    with torch.cuda.stream(ens.stream):
        d_seq = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, seed=0)).cuda()
    ens.stream.synchronize()
    steps = max(50, int(0.25 / (2e-9 * n * M * (L / 8))))

    def run(k):
        for i in range(k):
            ens.launch(d_seq, n, slot=i & 1, want="mean")
            if i:
                ens.finish((i - 1) & 1)
        out = ens.finish((k - 1) & 1)
        ens.stream.synchronize()
        torch.cuda.synchronize()
        return out

    res, outs = {0: [], 1: []}, {}
    for rep in range(4):
        for q in (0, 1):
            eng.set_option("fuse_mean_batch", q)
            run(10)
            t0 = time.perf_counter()
            out = run(steps)
            res[q].append((time.perf_counter() - t0) / steps * 1e6)
            outs[q] = out.cpu().numpy().copy()
    same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    m0, m1 = float(np.median(res[0])), float(np.median(res[1]))
    print(f"{name:30s} mean kernel {m0:9.2f} us/step   fused {m1:9.2f} us/step  ({(m1 / m0 - 1) * 100:+.1f} %)   same bits {same}   "
          f"runs {[round(x, 1) for x in res[0]]} / {[round(x, 1) for x in res[1]]}", flush=True)
eng.set_option("fuse_mean_batch", 0)
