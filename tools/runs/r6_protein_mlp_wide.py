#!/usr/bin/env python3
"""Round 6: DyNA-PPO's default MLP(seq_len, 200, alphabet) and the MLP surrogate on PROTEIN landscapes (dyna_ppo.py:54 with AAV's L = 90, 20
letters): the first layer's rows (1.5 MB) do not fit LDS.  mlp_l1_pos = 0: every sequence gathers its seq_len rows from L2 (rounds 1-5);
1: the first layer position-major by k_mlp_l1_pos (rows L2 -> LDS once per 16-32 tiles) + the dense kernel from a scratch (H > 128: H x H
layers through LDS slabs).  Time of the call's launches from fx_debug_time_score.  -> profiles/r6_protein_mlp_wide.log"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
CASES = [("mlp H=200 L=90 A=20 N=1e5", 90, 200, 1, 100_000), ("mlp H=200 L=90 A=20 N=1e4", 90, 200, 1, 10_000), ("mlp H=200 L=90 A=20 N=1e6", 90, 200, 1, 1_000_000),
         ("mlp H=200 L=237 A=20 N=1e5", 237, 200, 1, 100_000), ("mlp H=100 L=90 A=20 N=1e5", 90, 100, 1, 100_000), ("mlp H=200 L=30 A=20 N=1e5", 30, 200, 1, 100_000),
         ("3 x mlp H=200 L=90 A=20 N=1e5", 90, 200, 3, 100_000), ("mlp H=200 L=90 A=20 N=3e4", 90, 200, 1, 30_000), ("mlp H=100 L=90 A=20 N=2e4", 90, 100, 1, 20_000)]
for name, L, H, M, n in CASES:
    mods = build_members("mlp", L, AAS, M, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 0)).cuda()
    stride = (n + 63) // 64 * 64
    opts = (0, 1, 2) if os.environ.get('FX_L1_EXP') else (0, 1)
    planes = {q: torch.zeros((M, stride), dtype=torch.float32, device="cuda") for q in opts}
    res = {q: [] for q in opts}
    for rep in range(3):
        for q in opts:
            eng.set_option("mlp_l1_pos", q)
            ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes[q], stride, min_ms=30.0)
            res[q].append(ms * 1e3)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(planes[0][:, :n], planes[q][:, :n])) for q in opts)
    med = {q: float(np.median(res[q])) for q in opts}
    fr = {q: roofline_block("mlp", L, 20, H, 0, 0, M, n, med[q] * 1e-3, "k")["frac"] for q in opts}
    print(f"{name:30s} " + "   ".join(f"mlp_l1_pos={q}: {med[q]:9.2f} us ({fr[q]:.3f})" for q in opts) + f"   same bits {same}", flush=True)
eng.set_option("mlp_l1_pos", 1)
