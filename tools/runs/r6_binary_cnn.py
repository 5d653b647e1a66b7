import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flexs_amd import _native, synth
from tools.bench_common import build_members, roofline_block, time_launches
eng = _native.Engine.get(0)
for H in (200, 100, 64):
    mods = build_members("cnn", 40, "01", 1, 0, Hx=H)
    n = 100_000
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, 40, "01", 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
    us = time_launches(eng, mods, d_in.data_ptr(), n, 40, mods[0]._lut, planes, stride, min_ms=10.0)[0] * 1e3
    print(f"binary cnn L=40 H={H} N=1e5: {us:.1f} us")
