#!/usr/bin/env python3
"""Launches for a kernel trace of one protein-MLP fit (rocprofv3 --kernel-trace --stats -- python tools/runs/r6_train_protein_trace.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from flexs_amd import synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

for kind, L, H in (("mlp", 90, 200), ("mlp", 90, 100)):
    mod = build_members(kind, L, AAS, 1, 0, Hx=H)[0]
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(512, L, AAS, 1))
    y = np.random.default_rng(0).normal(size=512)
    mod.train(seqs, y)
    mod.train(seqs, y)
