#!/usr/bin/env python3
"""Round 6: issued-MFMA fraction of every (model family, FLEXS landscape shape) pair at virtual-screen size, one member, inputs in HBM --
looking for shape families the bench list never covered (the protein MLP was at 0.07-0.16).  -> profiles/r6_shape_survey.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
LAND = [("TF-binding", 8, "TGCA"), ("RNA14", 14, "UGCA"), ("RNA50", 50, "UGCA"), ("RNA100", 100, "UGCA"), ("AAV", 90, AAS), ("GFP", 237, AAS),
        ("Rosetta 3MSI", 66, AAS), ("binary NK-like", 40, "01")]
MODELS = [("cnn F32 H100 K5", "cnn", 100, 32, 5), ("mlp H100", "mlp", 100, 0, 0), ("mlp H200", "mlp", 200, 0, 0), ("ge H100", "ge", 100, 0, 0), ("cnn F32 H200 K5", "cnn", 200, 32, 5)]
for mname, kind, H, F, K in MODELS:
    for lname, L, alpha in LAND:
        for n in (100_000, 20_000):
            try:
                mods = build_members(kind, L, alpha, 1, 0, Hx=H, Fx=F or 32, Kx=K or 5)
                d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
                stride = (n + 63) // 64 * 64
                planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
                us = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=10.0)[0] * 1e3 for _ in range(2)]))
                fr = roofline_block(kind, L, len(alpha), H, F, K, 1, n, us * 1e-3, "k")["frac"]
                print(f"{mname:16s} {lname:15s} L={L:3d} A={len(alpha):2d} N={n:6d}  {us:9.2f} us  issued {fr:.3f}  {n / us:8.1f} seq/us", flush=True)
            except Exception as ex:  # noqa: BLE001
                print(f"{mname:16s} {lname:15s} L={L:3d} A={len(alpha):2d} N={n:6d}  {type(ex).__name__}: {str(ex)[:80]}", flush=True)
