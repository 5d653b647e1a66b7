import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flexs_amd import _native, synth
from tools.bench_common import AAS, build_members, time_launches
eng = _native.Engine.get(0)
for L, alpha in ((14, "UGCA"), (30, "UGCA"), (50, "UGCA"), (100, "UGCA"), (90, AAS)):
    for M in (1, 3):
        mods = build_members("cnn", L, alpha, M, 0)
        for n in (16, 100, 1000):
            d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
            stride = (n + 63) // 64 * 64
            planes = torch.zeros((M, stride), dtype=torch.float32, device="cuda")
            us = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=5.0)[0] * 1e3 for _ in range(3)]))
            print(f"cnn L={L:3d} A={len(alpha):2d} M={M} N={n:5d}: kernel(s) {us:7.2f} us", flush=True)
