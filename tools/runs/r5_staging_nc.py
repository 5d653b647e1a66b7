"""The input staging area as coherent (default) or non-coherent pinned host memory: end-to-end time of get_fitness(list[str])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
def med(f, n=21):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
AAS = s_utils.AAS
cases = [("C2 3xCNN L=8", lambda s: bm.CNN(8, 32, 100, "TGCA", seed=s), 3, 8, "TGCA", 100_000),
         ("C3 MLP L=14", lambda s: bm.MLP(14, 100, "UGCA", seed=s), 1, 14, "UGCA", 100_000),
         ("C4 8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS, 100_000),
         ("3xMLP L=14", lambda s: bm.MLP(14, 100, "UGCA", seed=s), 3, 14, "UGCA", 100_000)]
for tag, make, M, L, alpha, n in cases:
    mods = [make(s) for s in range(M)]
    model = mods[0] if M == 1 else flexs_amd.Ensemble(mods)
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
    res = {}
    for nc in (0, 1, 0, 1):
        eng.set_option("staging_noncoherent", nc)
        got = np.asarray(model.get_fitness(seqs)).copy()
        res.setdefault(nc, []).append((got, med(lambda: model.get_fitness(seqs))))
    same = all(np.array_equal(res[0][0][0].view(np.uint32), g.view(np.uint32)) for v in res.values() for g, _ in v)
    print(f"{tag} n={n}: same bits {same}; coherent {res[0][0][1]:.0f} / {res[0][1][1]:.0f} us, non-coherent {res[1][0][1]:.0f} / {res[1][1][1]:.0f} us", flush=True)
eng.set_option("staging_noncoherent", 0)
