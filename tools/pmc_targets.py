#!/usr/bin/env python3
"""Launch the kernels whose counters are quoted in DESIGN.md (for rocprofv3 --pmc passes): the BASELINE configs other
than the bench kernel, one workload per kernel / grid so that summarize_pmc.py can tell them apart, and K4 (min_dist)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from flexs_amd.baselines.models.keras_model import Architecture  # noqa: E402

AAS = "ILVAGMFYWEDQNHCRKSTP"
eng = _native.Engine.get(0)


def run(kind, L, alpha, M, N, F=0, K=0, reps=5, blocks=0, H=100):
    # `blocks` (workgroups) only tags the workload: rocprofv3 reports the grid size per dispatch, and two workloads of one
    # kernel instantiation would otherwise be averaged together (254 or 252 of 256 CUs: < 2 % off the full-grid figures)
    eng.set_option("grid_blocks", blocks)
    arch = Architecture(kind, L, len(alpha), H, num_filters=F, kernel_size=K)
    ms = []
    for m in range(M):
        nm = _native.NativeModel(eng, {"cnn": 0, "mlp": 1, "ge": 2}[kind], L, len(alpha), F, H, K)
        nm.set_weights(synth.synthetic_weights(arch.shapes(), 1000 + m))
        ms.append(nm)
    d_in = torch.from_numpy(synth.random_sequence_bytes(N, L, alpha, 0)).cuda()
    stride = (N + 63) // 64 * 64
    d_pl = torch.empty((M, stride), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    lut = _native.make_lut(alpha)
    for _ in range(reps):
        eng.score_planes_dev(ms, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride)
    eng.sync()


run("cnn", 8, "TGCA", 1, 10_000, 32, 5)                       # C1                 grid 256 x 512
run("cnn", 8, "TGCA", 3, 10_000, 32, 5, blocks=254)           # C2 at 1e4          grid 254 x 512
run("mlp", 14, "UGCA", 1, 100_000)                            # C3                 grid 256 x 1024
run("mlp", 14, "UGCA", 1, 1_000_000, blocks=254)              #                    grid 254 x 1024
run("ge", 90, AAS, 8, 100_000)                                # C4                 grid 256 x 1024
run("ge", 90, AAS, 8, 1_000_000, blocks=254)
run("ge", 90, AAS, 1, 100_000, blocks=252)
run("cnn", 237, AAS, 3, 16_384, 32, 5, reps=2)                # C5 kernel (pair form)
run("cnn", 8, "TGCA", 3, 1_000_000, 32, 5, reps=3)            # bench kernel, long launch
# round 6: the C5 launch at one GPU's share of the 5e5 batch, and the wide hidden layers (dyna_ppo.py:54: MLP(seq_len, 200, alphabet))
run("cnn", 237, AAS, 3, 62_500, 32, 5, reps=2, blocks=254)    # C5 at 62 500 rows     grid 254 x ...
run("mlp", 14, "UGCA", 1, 100_000, H=200)                     # slab form, leftover tiles walked cooperatively (dense_slab_coop = 3)
eng.set_option("dense_slab_coop", 0)
run("mlp", 14, "UGCA", 1, 100_000, H=200, blocks=254)         # ... and as whole lockstep rounds (round 5's form), tagged by the grid
eng.set_option("dense_slab_coop", 3)
run("cnn", 8, "TGCA", 1, 100_000, 32, 5, H=200)               # CNN with a 200-unit head
run("ge", 90, AAS, 1, 100_000, H=200, blocks=250)
run("mlp", 90, AAS, 1, 100_000, H=200, blocks=248)          # protein MLP (dyna_ppo.py:54 on AAV): k_mlp_l1_pos + the dense kernel from its scratch (round 6)
eng.set_option("grid_blocks", 0)
# K4: NoisyAbstractModel neighbour search, Levenshtein, RNA L = 14 and protein L = 90
rng = np.random.default_rng(0)
for L, nsym, Q, C in ((14, 4, 2000, 20000), (90, 20, 200, 20000)):
    cache = rng.integers(65, 65 + nsym, (C, L)).astype(np.uint8)
    q = cache[rng.integers(0, C, Q)].copy()
    mut = rng.random(q.shape) < 0.1
    q[mut] = rng.integers(65, 65 + nsym, mut.sum())
    dc = _native.NativeCache(eng, L)
    dc.append(cache)
    for _ in range(3):
        dc.min_dist(q, 0)
