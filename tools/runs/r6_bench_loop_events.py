#!/usr/bin/env python3
"""Round 6: what the HIP event pair per step costs bench.py's timed loop (tools/bench_common.run_pipelined): wall time per step of the
headline step (3 x CNN L=8, N=1e5 resident in HBM) with an event pair around every K1 launch, every 4th, and none; K = 20 / 100 / 1000.
-> profiles/r6_bench_loop_events.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from flexs_amd import distributed as fd  # noqa: E402
from tools import bench_common as bc  # noqa: E402

eng = _native.Engine.get(0)
L, M, n, alpha = 8, 3, 100_000, "TGCA"
ens = fd.DistributedEnsemble(bc.build_members("cnn", L, alpha, M, 0), mode="sequence")
with torch.cuda.stream(ens.stream):
    d_seq = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, seed=0)).cuda()
ens.stream.synchronize()
bc.run_pipelined(ens, d_seq, n, 200, 20, torch, dist, False)           # clocks up
for K in (20, 100, 1000):
    for every in (1, 4, 0):
        rows = []
        for rep in range(5):
            el, _, kern = bc.run_pipelined(ens, d_seq, n, K, 5, torch, dist, False, want_events=every)
            rows.append((el / K * 1e6, kern * 1e3 if kern else float("nan")))
        a = np.array(rows)
        print(f"K={K:5d} events every {every}: {np.median(a[:, 0]):8.2f} us/step (runs {[round(x, 1) for x in a[:, 0]]})  kernel_ms by events {np.median(a[:, 1]):7.2f} us", flush=True)
