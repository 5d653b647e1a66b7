#!/usr/bin/env python3
"""Round 6: KerasModel.train (keras_model.py:49-67: 20 epochs, batch 256) per model family on protein / long shapes -- looking for a
pathological case like the scoring side's protein MLP.  -> profiles/r6_train_protein_survey.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from flexs_amd import synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

for name, kind, L, alpha, H in (("mlp H200 AAV", "mlp", 90, AAS, 200), ("mlp H100 AAV", "mlp", 90, AAS, 100), ("ge H100 AAV", "ge", 90, AAS, 100),
                                ("cnn AAV", "cnn", 90, AAS, 100), ("mlp H200 GFP", "mlp", 237, AAS, 200), ("ge H100 GFP", "ge", 237, AAS, 100),
                                ("mlp H100 RNA100", "mlp", 100, "UGCA", 100), ("mlp H100 RNA14", "mlp", 14, "UGCA", 100), ("mlp H200 RNA14", "mlp", 14, "UGCA", 200)):
    for n in (500, 2000):
        mod = build_members(kind, L, alpha, 1, 0, Hx=H)[0]
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
        y = np.random.default_rng(0).normal(size=n)
        mod.train(seqs, y)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); mod.train(seqs, y); ts.append(time.perf_counter() - t0)
        steps = 20 * -(-n // 256)
        print(f"{name:18s} n={n:5d}: {np.median(ts) * 1e3:8.2f} ms per fit ({steps} steps, {np.median(ts) * 1e6 / steps:7.1f} us per step)", flush=True)
