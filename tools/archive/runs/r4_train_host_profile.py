"""Where the host side of Ensemble.train goes (3 x CNN L=8, 1000 sequences): cProfile of one call."""
import cProfile, pstats, sys, time; sys.path.insert(0, ".")
import numpy as np, torch, flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
for _ in range(3): ens.train(seqs, y)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); ens.train(seqs, y); ts.append((time.perf_counter() - t0) * 1e3)
print("Ensemble.train ms:", [round(t, 2) for t in ts])
pr = cProfile.Profile(); pr.enable(); ens.train(seqs, y); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
