#!/usr/bin/env python3
"""Round 6: from which batch size the position-major first layer (mlp_l1_pos) pays: protein MLP, mlp_l1_pos = 1 (position-major above mlp_l1_pos_tiles = 2 tiles per CU, the small-launch
form below) against 0 (the small-launch form to 4 tiles per CU, the gather form beyond), N = 2e3 ... 3e4.  -> profiles/r6_protein_mlp_wide.log (second table)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, time_launches  # noqa: E402

eng = _native.Engine.get(0)
for L, H in ((90, 200), (90, 100), (237, 100)):
    mods = build_members("mlp", L, AAS, 1, 0, Hx=H)
    for n in (1_000, 2_000, 4_096, 8_192, 12_000, 16_384, 30_000):
        d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 0)).cuda()
        stride = (n + 63) // 64 * 64
        planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
        res = {}
        for q in (0, 1):
            eng.set_option("mlp_l1_pos", q)
            xs = [time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=15.0)[0] * 1e3 for _ in range(3)]
            res[q] = float(np.median(xs))
        print(f"mlp L={L} H={H} N={n:6d} ({(n + 15) // 16 / 256:5.2f} tiles per CU): gather {res[0]:8.2f} us   position-major {res[1]:8.2f} us  ({(res[1] / res[0] - 1) * 100:+.0f} %)", flush=True)
eng.set_option("mlp_l1_pos", 1)
