"""NoisyAbstractModel over a device table (TF-binding style, L = 8): wall time of get_fitness for 1 / 10 / 100 uncached queries against a cache of
~3000, and of the C call alone.  Written for round 5's one-launch experiment (engine option nam_one, removed again: csrc/OPTIONS.md); the option
loop is skipped when the library does not know the option."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
L, alpha = 8, "TGCA"
rng = np.random.default_rng(0)
vals = rng.random(4 ** L)

class Table(flexs_amd.Landscape):
    batch_safe = True
    def __init__(self):
        super().__init__("Table"); self._t = None
    def _native_table(self):
        if self._t is None:
            self._t = _native.NativeTable(eng, vals, alpha, bits=2)
        return self._t
    def _fitness_function(self, seqs):
        return self._native_table().lookup(_native.sequences_to_bytes([str(s) for s in seqs], L=L))

pool = list(dict.fromkeys(synth.bytes_to_strings(synth.random_sequence_bytes(60000, L, alpha, 9))))
res = {}
def _set(one):
    try:
        eng.set_option("nam_one", one); return True
    except Exception:
        return one == 0
for one in (0, 1):
    if not _set(one):
        res[(one, "out")] = None
        continue
    np.random.seed(5)
    land = Table()
    nam = bm.NoisyAbstractModel(land, 0.9)
    nam.train(pool[:3000], land.get_fitness(pool[:3000]))
    at = 3000
    outs = []
    for q in (1, 10, 100):
        ts = []
        for rep in range(60):
            batch = pool[at:at + q]; at += q
            t0 = time.perf_counter(); o = nam.get_fitness(batch); ts.append(time.perf_counter() - t0)
            outs.append(o)
        res[(one, q)] = np.median(ts[5:]) * 1e6
    res[(one, "out")] = np.concatenate(outs)
if res.get((1, "out")) is not None:
    print("same floats:", np.array_equal(res[(0, "out")], res[(1, "out")]))
for q in (1, 10, 100):
    print(f"NAM get_fitness of {q:3d} uncached queries, cache ~3000-9000: chain {res[(0, q)]:.1f} us" + (f", one launch {res[(1, q)]:.1f} us" if (1, q) in res else ""))

# the C call alone (fx_cache_nam_query through ctypes): what the kernel chain / the one launch cost without the model's Python around them
cache_rows = synth.random_sequence_bytes(3000, L, alpha, 21)
tab = _native.NativeTable(eng, vals, alpha, bits=2)
alpha_tab = 0.9 ** np.arange(L + 1)
for one in (0, 1):
    if not _set(one):
        continue
    dc = _native.NativeCache(eng, L); dc.append(cache_rows)
    for q in (1, 10, 100):
        qs = synth.random_sequence_bytes(q, L, alpha, 33); Ev = np.random.default_rng(1).standard_exponential(q)
        app = synth.random_sequence_bytes(q, L, alpha, 34)
        ts = []
        for rep in range(300):
            t0 = time.perf_counter(); dc.nam_query(tab, qs, Ev, alpha_tab, 0, append=app if rep % 2 else None); ts.append(time.perf_counter() - t0)
        print(f"fx_cache_nam_query alone, {q:3d} queries, nam_one = {one}: median {np.median(ts[20:]) * 1e6:.1f} us", flush=True)
