#!/usr/bin/env python3
"""Round 6: protein CNN at mid size (units between half and all of the CUs, and just beyond): kernel time per call for N x M.
-> profiles/r6_pair_mid.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, time_launches  # noqa: E402

eng = _native.Engine.get(0)
tag = os.environ.get("FX_TAG", "")
for L in (90, 237):
    for M, n in ((1, 1000), (1, 2500), (1, 4000), (3, 1000), (3, 1300), (3, 2000), (3, 4000), (8, 500)):
        mods = build_members("cnn", L, AAS, M, 0)
        d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 0)).cuda()
        stride = (n + 63) // 64 * 64
        planes = torch.zeros((M, stride), dtype=torch.float32, device="cuda")
        us = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=10.0, reps0=5)[0] * 1e3 for _ in range(3)]))
        print(f"{tag} cnn L={L} M={M} N={n:5d} (units {M * ((n + 15) // 16):4d}): {us:9.1f} us", flush=True)
