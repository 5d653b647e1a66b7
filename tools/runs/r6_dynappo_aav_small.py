#!/usr/bin/env python3
"""Round 6: DyNA-PPO's default native members on AAV (dyna_ppo.py:52-56: GlobalEpistasis(L, 100), MLP(L, 200), CNN(L, 32, 100); L = 90, 20 letters)
as explorer-size get_fitness(list[str]) calls: per member and as one ensemble.  -> profiles/r6_dynappo_aav_small.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import flexs_amd  # noqa: E402
from flexs_amd import synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

L = 90
ge = build_members("ge", L, AAS, 1, 0, Hx=100)[0]
mlp = build_members("mlp", L, AAS, 1, 0, Hx=200)[0]
cnn = build_members("cnn", L, AAS, 1, 0, Hx=100)[0]
ens = flexs_amd.Ensemble([ge, mlp, cnn])
for n in (1, 10, 100, 1000, 4000):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, n))
    row = []
    for name, m in (("GE", ge), ("MLP200", mlp), ("CNN", cnn), ("ensemble", ens)):
        for _ in range(20):
            m.get_fitness(seqs)
        ts = []
        for _ in range(200 if n <= 100 else 50):
            t0 = time.perf_counter(); m.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        row.append(f"{name} {np.median(ts) * 1e6:7.1f} us")
    print(f"N={n:5d}: " + "   ".join(row), flush=True)
