import cProfile, pstats, sys, time; sys.path.insert(0, ".")
import numpy as np, torch, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
for _ in range(3): ens.train(seqs, y)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): ens.train(seqs, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
