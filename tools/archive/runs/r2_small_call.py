"""Round-2 probe: where do the ~30 us of an explorer-size call go?  Ensemble(3 x CNN, L = 8), N = 20."""
import sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import perf_survey as ps
from flexs_amd import _native, synth
import flexs_amd
from flexs_amd.baselines import models as bm

eng = ps.eng
nat = ps.natives("cnn", 8, 4, 100, 3, 32, 5)
lut = _native.make_lut("TGCA")
b = synth.random_sequence_bytes(20, 8, "TGCA", 0)
for want in ((False, True), (True, False), (True, True)):
    for _ in range(50): eng.score(nat, b, lut, want_matrix=want[0], want_mean=want[1])
    ts = []
    for _ in range(2000):
        t0 = time.perf_counter(); eng.score(nat, b, lut, want_matrix=want[0], want_mean=want[1]); ts.append(time.perf_counter() - t0)
    print({"what": f"eng.score N=20 M=3 want_matrix={want[0]} want_mean={want[1]}", "median_us": round(float(np.median(ts)) * 1e6, 2), "p10_us": round(float(np.percentile(ts, 10)) * 1e6, 2)}, flush=True)
for M in (1, 3):
    ps.time_score("cnn", 8, "TGCA", 100, M, 20, 32, 5, reps=2000, label=f"K1 only (C loop) M={M} N=20")
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(b)
for _ in range(50): ens.get_fitness(seqs)
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
print({"what": "Ensemble.get_fitness(list[str]) N=20", "median_us": round(float(np.median(ts)) * 1e6, 2)}, flush=True)
m0 = ens.models[0]
for _ in range(50): m0.get_fitness(seqs)
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); m0.get_fitness(seqs); ts.append(time.perf_counter() - t0)
print({"what": "CNN.get_fitness(list[str]) N=20 (one member, no mean kernel)", "median_us": round(float(np.median(ts)) * 1e6, 2)}, flush=True)
