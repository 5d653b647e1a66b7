#!/usr/bin/env python3
"""Round 6: end to end -- MLP.get_fitness(list[str]) -> ndarray on a protein landscape (host strings in, host array out), position-major first
layer on (mlp_l1_pos = 1: such calls are packed first and copied to the device) against off (0: launched first, rows gathered from L2 per
sequence).  -> profiles/r6_protein_mlp_wide.log (third table)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

eng = _native.Engine.get(0)
for L, H, n in ((90, 200, 100_000), (90, 200, 20_000), (90, 100, 100_000), (237, 200, 50_000)):
    mod = build_members("mlp", L, AAS, 1, 0, Hx=H)[0]
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, 0))
    res, outs = {}, {}
    for q in (0, 1, 0, 1):
        eng.set_option("mlp_l1_pos", q)
        for _ in range(3):
            outs[q] = mod.get_fitness(seqs)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); mod.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        res[q] = float(np.median(ts)) * 1e6
    print(f"MLP(L={L}, H={H}, 20 letters).get_fitness({n} str): gather form {res[0]:9.1f} us   position-major {res[1]:9.1f} us  ({(res[1] / res[0] - 1) * 100:+.0f} %)  "
          f"same bits {np.array_equal(outs[0], outs[1])}", flush=True)
eng.set_option("mlp_l1_pos", 1)
