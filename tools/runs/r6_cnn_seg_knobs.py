import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flexs_amd import _native, synth
from tools.bench_common import build_members, time_launches
eng = _native.Engine.get(0)
for L in (20, 30, 50, 100):
    mods = build_members("cnn", L, "UGCA", 1, 0)
    for n in (16, 200):
        d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, "UGCA", 0)).cuda()
        stride = (n + 63) // 64 * 64
        planes = torch.zeros((1, stride), dtype=torch.float32, device="cuda")
        row = []
        for name, opts in (("default", {}), ("seg_multi=0", {"cnn_seg_multi": 0}), ("seg=0", {"cnn_seg": 0}), ("quad=0", {"cnn_quad": 0})):
            for k, v in opts.items():
                eng.set_option(k, v)
            try:
                us = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=5.0)[0] * 1e3 for _ in range(3)]))
                row.append(f"{name} {us:6.2f}")
            except Exception as ex:
                row.append(f"{name} ERR")
            for k in opts:
                eng.set_option(k, {"cnn_seg_multi": 1, "cnn_seg": -1, "cnn_quad": 1}[k])
        print(f"cnn L={L:3d} N={n:4d}: " + "   ".join(row), flush=True)
