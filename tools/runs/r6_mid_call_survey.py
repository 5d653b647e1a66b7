#!/usr/bin/env python3
"""Round 6: get_fitness(list[str]) -> ndarray at 5e3 / 2e4 / 1e5 sequences per (model family, FLEXS landscape shape) pair against the kernel time on
resident bytes: where does a host call lose most?  -> profiles/r6_mid_call_survey.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, time_launches  # noqa: E402

eng = _native.Engine.get(0)
LAND = [("TF-binding", 8, "TGCA"), ("RNA14", 14, "UGCA"), ("RNA100", 100, "UGCA"), ("AAV", 90, AAS), ("GFP", 237, AAS)]
MODELS = [("cnn", "cnn", 100), ("mlp H100", "mlp", 100), ("mlp H200", "mlp", 200), ("ge", "ge", 100)]
for mname, kind, H in MODELS:
    for lname, L, alpha in LAND:
        mod = build_members(kind, L, alpha, 1, 0, Hx=H)[0]
        row = []
        for n in (5_000, 20_000, 100_000):
            if kind == "cnn" and L == 237 and n > 20_000:
                continue
            b = synth.random_sequence_bytes(n, L, alpha, n)
            seqs = synth.bytes_to_strings(b)
            for _ in range(3):
                mod.get_fitness(seqs)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); mod.get_fitness(seqs); ts.append(time.perf_counter() - t0)
            d_in = torch.from_numpy(b).cuda()
            stride = (n + 63) // 64 * 64
            planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
            k_us = time_launches(eng, [mod], d_in.data_ptr(), n, L, mod._lut, planes, stride, min_ms=10.0)[0] * 1e3
            e_us = float(np.median(ts)) * 1e6
            row.append(f"N={n}: call {e_us:8.1f} us, kernel {k_us:8.1f} us ({k_us / e_us:4.2f})")
        print(f"{mname:9s} {lname:10s} L={L:3d} A={len(alpha):2d}  " + "   ".join(row), flush=True)
