"""Explorer-size calls of the small entry points with their buffers in mapped pinned memory (default) against the copy
path (zero_copy_bytes = 0): population decode+score (CMA-ES / DyNA-PPO), nearest cached neighbour, blend, table look-up."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
def med(f, n=2000):
    for _ in range(200): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6
rng = np.random.default_rng(0)
cases = {}
members = [bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]
nat = [m.native() for m in members]
x = rng.random((40, 8, 4))
cases["decode + score, population 40, 3 x CNN L=8"] = lambda: eng.decode_score(nat, x, "TGCA", members[0]._lut, want_matrix=False, want_mean=True)
cache = _native.NativeCache(eng, 14)
cache.append(synth.random_sequence_bytes(3000, 14, "UGCA", 1))
q = synth.random_sequence_bytes(100, 14, "UGCA", 2)
cases["nearest neighbour of 100 queries in a cache of 3000 (L=14)"] = lambda: cache.min_dist(q, _native.FX_LEVENSHTEIN)
sig, noi, dd, tab = rng.random(100), rng.random(100), rng.integers(0, 5, 100).astype(np.int32), 0.9 ** np.arange(6)
cases["blend of 100 queries"] = lambda: eng.nam_combine(sig, noi, dd, tab)
for name, f in cases.items():
    for rep in range(2):
        for zc in (262144, 0):
            eng.set_option("zero_copy_bytes", zc)
            print(f"{name}: zero_copy_bytes={zc} [{rep}]: {med(f):.1f} us", flush=True)
eng.set_option("zero_copy_bytes", 262144)
