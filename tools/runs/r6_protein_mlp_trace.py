#!/usr/bin/env python3
"""Launches for a kernel trace of the protein MLP path (rocprofv3 --kernel-trace --stats -- python tools/runs/r6_protein_mlp_trace.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members  # noqa: E402

eng = _native.Engine.get(0)
for L, H, n in ((90, 200, 100_000), (90, 100, 100_000), (237, 200, 100_000)):
    mods = build_members("mlp", L, AAS, 1, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
    for _ in range(5):
        eng.score_planes_dev([m.native() for m in mods], d_in.data_ptr(), n, L, mods[0]._lut, planes.data_ptr(), stride)
    eng.sync()
