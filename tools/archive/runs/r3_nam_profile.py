"""Where a NoisyAbstractModel.get_fitness call of the CbAS pattern (100 sequences, batch_safe landscape) spends its time."""
import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines.models import NoisyAbstractModel
Lx, alpha = 14, "UGCA"
class _Synth(flexs_amd.Landscape):
    def __init__(self, batch_safe):
        super().__init__("synth"); self.batch_safe = batch_safe
        self._w = (np.arange(1, Lx + 1, dtype=np.int64) * 2654435761) % 1000003
    def _fitness_function(self, seqs):
        b = _native.sequences_to_bytes([str(s_) for s_ in seqs], L=Lx).astype(np.int64)
        return ((b * self._w).sum(axis=1) % 1000) / 1000.0
def run(profile):
    np.random.seed(0)
    model = NoisyAbstractModel(_Synth(True), 0.9)
    model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, Lx, alpha, 5)), np.random.random(1000))
    batches = [synth.bytes_to_strings(synth.random_sequence_bytes(100, Lx, alpha, 100 + c)) for c in range(20)]
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr: pr.enable()
    for bch in batches: model.get_fitness(bch)
    if pr: pr.disable()
    t = time.perf_counter() - t0
    print(f"20 calls x 100: {t * 1e3:.2f} ms = {t / 20 * 1e6:.0f} us per call, {2000 / t:.3g} sequences/s")
    if pr: pstats.Stats(pr).sort_stats("tottime").print_stats(22)
run(False); run(False); run(True)
