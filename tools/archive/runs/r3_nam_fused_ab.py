"""NoisyAbstractModel over a device table landscape (TF-binding style, 8-mers): the fused batch (fx_cache_nam_query) against
the two-batched-look-ups path (the same table behind a landscape without `_native_table`) and the one-by-one reference loop."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines.models import NoisyAbstractModel
L = 8
vals = np.random.default_rng(0).random(4 ** L)
class Table(flexs_amd.Landscape):
    batch_safe = True
    def __init__(self):
        super().__init__("table"); self._L = L; self._t = None
    def _native_table(self):
        if self._t is None: self._t = _native.NativeTable(_native.Engine.get(None), vals, "ACGT", bits=2)
        return self._t
    def _fitness_function(self, seqs):
        return self._native_table().lookup(_native.sequences_to_bytes([str(s) for s in seqs], L=L))
class Batched(flexs_amd.Landscape):
    batch_safe = True
    def __init__(self, inner):
        super().__init__("batched"); self.inner = inner
    def _fitness_function(self, seqs): return self.inner._fitness_function(seqs)
class Plain(Batched):
    batch_safe = False
def run(make, n_per_call, calls):
    res = []
    for rep in range(3):
        np.random.seed(0)
        model = NoisyAbstractModel(make(), 0.9)
        model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, L, "ACGT", 5)), np.random.random(1000))
        batches = [synth.bytes_to_strings(synth.random_sequence_bytes(n_per_call, L, "ACGT", 100 + c)) for c in range(calls)]
        t0 = time.perf_counter()
        out = [model.get_fitness(b) for b in batches]
        res.append((time.perf_counter() - t0) / calls * 1e6)
    return min(res), np.concatenate(out)
for n_per_call, calls in ((100, 20), (20, 100), (1, 300)):
    ref = None
    for name, make in (("fused device batch", Table), ("two batched look-ups", lambda: Batched(Table())), ("one-by-one (reference loop)", lambda: Plain(Table()))):
        us, out = run(make, n_per_call, calls)
        same = "" if ref is None else f", identical to the fused path: {np.array_equal(out, ref)}"
        ref = out if ref is None else ref
        print(f"{n_per_call:4d} sequences per call, {name}: {us:.1f} us per call, {n_per_call / us * 1e6:.3g} sequences/s{same}", flush=True)
