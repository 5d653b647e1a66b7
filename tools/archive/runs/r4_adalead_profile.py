"""Where one Adalead round (2000 model queries, 3 x CNN L=8 trained on the measured sequences, as in bench.py's explorer_round)
goes: wall time, calls by batch size, time inside get_fitness, cProfile."""
import cProfile, pstats, random, sys, time, collections; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import rollouts
eng = _native.Engine.get()
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3))
y = np.random.default_rng(0).random(1000)
ens.train(seqs, y, seed=0) if "seed" in ens.train.__code__.co_varnames else ens.train(seqs, y)
def once():
    random.seed(1)
    return rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet="TGCA")
for _ in range(3): once()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); once(); ts.append((time.perf_counter() - t0) * 1e3)
print("round ms:", [round(t, 2) for t in ts])
sizes = collections.Counter(); inside = [0.0]
orig = ens.get_fitness
def spy(s):
    sizes[len(s)] += 1
    t0 = time.perf_counter(); r = orig(s); inside[0] += time.perf_counter() - t0
    return r
ens.get_fitness = spy
t0 = time.perf_counter(); once(); total = time.perf_counter() - t0
ens.get_fitness = orig
print("get_fitness calls by size:", sorted(sizes.items()), "total calls", sum(sizes.values()), "sequences", sum(k * v for k, v in sizes.items()))
print("round %.2f ms of which inside get_fitness %.2f ms" % (total * 1e3, inside[0] * 1e3))
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
