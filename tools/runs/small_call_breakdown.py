"""Where the ~39 us of a small `Ensemble.get_fitness` call go: Python layers vs the C entry point."""
import sys, time; sys.path.insert(0, ".")
import ctypes as C
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm

L, alpha = 8, "TGCA"
members = [bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)]
ens = flexs_amd.Ensemble(members)
eng = _native.Engine.get()
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(20, L, alpha, 2))
b = _native.sequences_to_bytes(seqs, L=L)
nat = [m.native() for m in members]
lut = members[0]._lut

def med(f, n=2000):
    for _ in range(50): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6

arr = (C.c_void_p * 3)(*[m.handle for m in nat])
out = np.empty(20, np.float32)
lib, h = eng._lib, eng.handle
lutp = lut.ctypes.data_as(_native._u8p)
bp, op = b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)
print("raw fx_score (ctypes call only)      %.1f us" % med(lambda: lib.fx_score(h, arr, 3, bp, 20, L, lutp, None, op)))
print("Engine.score(natives, bytes)         %.1f us" % med(lambda: eng.score(nat, b, lut, want_matrix=False, want_mean=True)))
print("sequences_to_bytes(list of 20)       %.1f us" % med(lambda: _native.sequences_to_bytes(seqs, L=L)))
print("[m.native() for m in members]        %.1f us" % med(lambda: [m.native() for m in members]))
print("Ensemble.get_fitness(list of 20)     %.1f us" % med(lambda: ens.get_fitness(seqs)))
print("KerasModel.get_fitness(list of 20)   %.1f us" % med(lambda: members[0].get_fitness(seqs)))
