#!/usr/bin/env python3
"""Round 6 A/B: results of launched-first calls handed out in place (registered anonymous memory, the default) against copied into np.empty
(FLEXS_AMD_RESULTS_IN_PLACE = 0), interleaved; get_fitness(list[str]) wall time, median of 15 per leg, three rounds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flexs_amd  # noqa: E402
from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip() if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "n/a")
for name, kind, L, alpha, M, n in (("C2 3xCNN L=8", "cnn", 8, "TGCA", 3, 100_000), ("C3 MLP L=14", "mlp", 14, "UGCA", 1, 100_000),
                                   ("C4 8xGE L=90", "ge", 90, AAS, 8, 100_000), ("3xCNN L=8 matrix", "cnn", 8, "TGCA", 3, 100_000)):
    mods = build_members(kind, L, alpha, M, 0)
    model = (flexs_amd.Ensemble(mods, combine_with=(lambda x: x)) if "matrix" in name else flexs_amd.Ensemble(mods)) if M > 1 else mods[0]
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    res = {0: [], 1: []}
    for rnd in range(3):
        for leg in (1, 0):
            _native.RESULTS_IN_PLACE = leg
            for _ in range(3):
                model.get_fitness(seqs)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
            res[leg].append(float(np.median(ts)) * 1e6)
    _native.RESULTS_IN_PLACE = 1
    print(f"{name:20s} in place {[round(x, 1) for x in res[1]]} us   copied {[round(x, 1) for x in res[0]]} us", flush=True)
