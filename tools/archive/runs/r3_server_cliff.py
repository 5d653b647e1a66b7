"""A launched call issued while resident workgroups hold CUs: sizes around num_cus - resident workgroups."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
small = synth.bytes_to_strings(synth.random_sequence_bytes(20, 8, "TGCA", 1))
for n in (400, 1000, 1100, 1150, 1250, 1300, 1360, 1400, 2001):
    big = synth.bytes_to_strings(synth.random_sequence_bytes(n, 8, "TGCA", 2))
    ts = []
    for _ in range(30):
        for _ in range(4): ens.get_fitness(small)          # (resident again)
        t0 = time.perf_counter(); ens.get_fitness(big); ts.append((time.perf_counter() - t0) * 1e6)
    print(f"N={n} ({3 * ((n + 15) // 16)} work units) right after small calls: median {np.median(ts):.0f} us, max {max(ts):.0f} us, resident after: {eng.get_option('server_resident')}", flush=True)
