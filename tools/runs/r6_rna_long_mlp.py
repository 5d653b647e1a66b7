import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flexs_amd import _native, synth
from tools.bench_common import build_members, roofline_block, time_launches
eng = _native.Engine.get(0)
for name, L, H, n in (("rna L=100 H=100", 100, 100, 100_000), ("rna L=100 H=200", 100, 200, 100_000), ("rna L=50 H=200", 50, 200, 100_000), ("rna L=50 H=100", 50, 100, 100_000), ("rna L=100 H=100 N=2e4", 100, 100, 20_000)):
    mods = build_members("mlp", L, "UGCA", 1, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, "UGCA", 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = {q: torch.zeros((1, stride), dtype=torch.float32, device="cuda") for q in (0, 1)}
    res = {}
    for q in (0, 1):
        eng.set_option("mlp_l1_pos", q)
        res[q] = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=20.0)[0] * 1e3 for _ in range(3)]))
    fr = {q: roofline_block("mlp", L, 4, H, 0, 0, 1, n, res[q] * 1e-3, "k")["frac"] for q in (0, 1)}
    print(f"{name:24s} mlp_l1_pos=0 {res[0]:8.2f} us ({fr[0]:.3f})   =1 {res[1]:8.2f} us ({fr[1]:.3f})  same bits {bool(torch.equal(planes[0], planes[1]))}", flush=True)
