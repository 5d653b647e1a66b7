"""Where the 0.30 ms of Ensemble.get_fitness(list[str]) at N = 1e5 (3 x CNN L=8) go."""
import sys, time; sys.path.insert(0, ".")
import ctypes as C
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm

L, alpha, N = 8, "TGCA", 100_000
members = [bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)]
ens = flexs_amd.Ensemble(members)
eng = _native.Engine.get()
b = synth.random_sequence_bytes(N, L, alpha, 2)
seqs = synth.bytes_to_strings(b)
nat = [m.native() for m in members]
lut = members[0]._lut

def med(f, n=60):
    for _ in range(5): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

arr = (C.c_void_p * 3)(*[m.handle for m in nat])
out = np.empty(N, np.float32)
lib, h = eng._lib, eng.handle
lutp = lut.ctypes.data_as(_native._u8p)
stg = eng.staging_rows(N, L); stg[:] = b
sp = stg.ctypes.data_as(C.c_void_p); op = out.ctypes.data_as(C.c_void_p)
print("raw fx_score, bytes already in the pinned staging area, mean only   %.3f ms" % med(lambda: lib.fx_score(h, arr, 3, sp, N, L, lutp, None, op)))
bp = b.ctypes.data_as(C.c_void_p)
print("raw fx_score from a pageable numpy array (memcpy into staging)      %.3f ms" % med(lambda: lib.fx_score(h, arr, 3, bp, N, L, lutp, None, op)))
print("strpack.pack into the staging area                                  %.3f ms" % med(lambda: _native._strpack.pack(seqs, L, stg)))
print("np.empty(N, float32) + first touch                                  %.3f ms" % med(lambda: np.empty(N, np.float32).fill(0)))
print("Engine.score(natives, staged bytes)                                 %.3f ms" % med(lambda: eng.score(nat, stg, lut, want_matrix=False, want_mean=True)))
print("Ensemble.get_fitness(list[str])                                     %.3f ms" % med(lambda: ens.get_fitness(seqs)))
print("Ensemble.get_fitness(ndarray 'S8')                                  %.3f ms" % med(lambda a=np.array(seqs, dtype='S'): ens.get_fitness(a)))
ms = eng.time_score_planes(nat, __import__('torch').from_numpy(b).cuda().data_ptr(), N, L, lut, __import__('torch').empty((3, N), device='cuda').data_ptr(), N, 200) / 200
print("kernel alone (device-resident, from C)                              %.3f ms" % ms)
