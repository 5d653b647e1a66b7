"""Where the ~80 us of Python in a one-sequence NoisyAbstractModel.get_fitness go (device-table landscape, L = 8): cProfile over 3000 calls."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
L, alpha = 8, "TGCA"
vals = np.random.default_rng(0).random(4 ** L)

class Table(flexs_amd.Landscape):
    batch_safe = True
    def __init__(self):
        super().__init__("Table"); self._t = None; self._L = L
    def _native_table(self):
        if self._t is None:
            self._t = _native.NativeTable(eng, vals, alpha, bits=2)
        return self._t
    def _fitness_function(self, seqs):
        return self._native_table().lookup(_native.sequences_to_bytes([str(s) for s in seqs], L=L))

pool = list(dict.fromkeys(synth.bytes_to_strings(synth.random_sequence_bytes(60000, L, alpha, 9))))
np.random.seed(5)
land = Table(); nam = bm.NoisyAbstractModel(land, 0.9)
nam.train(pool[:3000], land.get_fitness(pool[:3000]))
for i in range(200):
    nam.get_fitness(pool[3000 + i:3001 + i])
ts = []
for i in range(2000):
    t0 = time.perf_counter(); nam.get_fitness(pool[4000 + i:4001 + i]); ts.append(time.perf_counter() - t0)
print(f"one uncached sequence per call: median {np.median(ts) * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for i in range(3000):
    nam.get_fitness(pool[7000 + i:7001 + i])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
