"""Wall time of `train` (native HIP path): the whole fit as ONE launch (train_persistent = 1, round 4) against two launches per
mini-batch step (round 3), and the split of a call into host preparation and device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()

def run(tag, make, L, alpha, n):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 3))
    y = np.random.default_rng(0).random(n)
    out = []
    for persistent in (1, 0):
        eng.set_option("train_persistent", persistent)
        model = make()
        model.train(seqs, y); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); model.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out.append(min(ts) * 1e3)
    eng.set_option("train_persistent", 0)
    print(f"{tag} n={n}: one launch per fit {out[0]:.2f} ms, two launches per step {out[1]:.2f} ms", flush=True)

run("Ensemble 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1000)
run("CNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 1000)
run("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 1000)
run("Ensemble 8xGE L=90", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=m) for m in range(8)]), 90, s_utils.AAS, 1000)
run("CNN L=90 A=20", lambda: bm.CNN(90, 32, 100, s_utils.AAS, seed=0), 90, s_utils.AAS, 1000)
run("Ensemble 3xCNN L=237 A=20", lambda: flexs_amd.Ensemble([bm.CNN(237, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 237, s_utils.AAS, 500)
