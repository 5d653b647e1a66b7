"""Round-2 probe: wall time of ONE explorer round on the GPU box -- train the 3-CNN ensemble on the measured sequences, then
an Adalead round (roll-outs of 1-20 sequence model calls, query budget 2000) -- split into its parts."""
import random
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import rollouts

L, alphabet = 8, "TGCA"
rng = np.random.default_rng(0)
ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alphabet, seed=m) for m in range(3)])
for n_meas in (100, 500, 1000):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n_meas, L, alphabet, 3))
    y = rng.random(n_meas)
    ens.train(seqs, y); torch.cuda.synchronize()
    t0 = time.perf_counter(); ens.train(seqs, y); torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    for fuse in (False, True):
        random.seed(1); np.random.seed(1)
        c0 = ens.cost
        t0 = time.perf_counter()
        new, preds = rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet=alphabet, fuse=fuse)
        t_prop = time.perf_counter() - t0
        print({"what": f"round n_measured={n_meas} fuse={fuse}", "train_3_members_ms": round(t_train * 1e3, 1), "adalead_round_ms": round(t_prop * 1e3, 1),
               "model_queries": ens.cost - c0, "proposed": len(new)}, flush=True)
