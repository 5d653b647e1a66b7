#!/usr/bin/env python3
"""Round 6: explorer-size get_fitness(list[str]) -> ndarray (host strings in, host scores out) for every (model family, FLEXS landscape shape)
pair, one member and a 3-member ensemble of the family: looking for a slow corner.  -> profiles/r6_small_call_survey.log"""
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

LAND = [("TF-binding", 8, "TGCA"), ("RNA14", 14, "UGCA"), ("RNA50", 50, "UGCA"), ("RNA100", 100, "UGCA"), ("AAV", 90, AAS), ("GFP", 237, AAS)]
MODELS = [("cnn", "cnn", 100), ("mlp H100", "mlp", 100), ("mlp H200", "mlp", 200), ("ge", "ge", 100)]


def med(fn, reps):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


for mname, kind, H in MODELS:
    for lname, L, alpha in LAND:
        mods = build_members(kind, L, alpha, 3, 0, Hx=H)
        ens = flexs_amd.Ensemble(mods)
        row = []
        for n in (1, 20, 1000):
            seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
            row.append(f"N={n}: 1 member {med(lambda: mods[0].get_fitness(seqs), 100):6.1f} us, 3 members {med(lambda: ens.get_fitness(seqs), 100):6.1f} us")
        print(f"{mname:9s} {lname:10s} L={L:3d} A={len(alpha):2d}  " + "   ".join(row), flush=True)
