import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm

class Table(flexs_amd.Landscape):
    def __init__(self):
        super().__init__("table")
    def _fitness_function(self, seqs):
        return np.array([(hash(str(s)) % 1000) / 1000.0 for s in seqs])

np.random.seed(0)
model = bm.NoisyAbstractModel(Table(), 0.9)
alpha, L = "UGCA", 14
model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, L, alpha, 5)), np.random.random(1000))
batches = [synth.bytes_to_strings(synth.random_sequence_bytes(100, L, alpha, 100 + c)) for c in range(40)]
for b in batches[:20]:
    model.get_fitness(b)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for b in batches[20:]:
    model.get_fitness(b)
t = time.perf_counter() - t0
pr.disable()
print("per call us", t / 20 * 1e6)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
