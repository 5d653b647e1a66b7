"""Where Ensemble.train (3 x CNN L=8, 1000 sequences) spends its wall time: Python around fx_train_fit, and the call's own timeline."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch, flexs_amd
from flexs_amd import _native, synth, training
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
for _ in range(3): ens.train(seqs, y)
inner = []
orig = _native.train_fit
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); inner.append((time.perf_counter() - t0) * 1e3); return r
_native.train_fit = timed
walls, profs = [], []
for _ in range(15):
    t0 = time.perf_counter(); ens.train(seqs, y); walls.append((time.perf_counter() - t0) * 1e3)
    profs.append([eng.get_option(f"train_prof_{k}") for k in range(5)])
p = np.median(np.array(profs), axis=0) / 1e3
print(f"Ensemble.train wall {np.median(walls):.2f} ms; inside _native.train_fit (ctypes call incl. argument conversion) {np.median(inner):.2f} ms")
print("fx_train_fit timeline, us since entry: image filled %.0f, upload enqueued %.0f, launches enqueued %.0f, synchronised %.0f, results copied out %.0f" % tuple(p))
