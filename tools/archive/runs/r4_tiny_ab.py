"""Tiny requests (<= 48 sequence bytes in the request word's own line, serve_tiny = 1) against the byte area (0): get_fitness latency."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
def med_us(fn, reps=500):
    for _ in range(50): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, float(np.percentile(ts, 99)) * 1e6
fams = [("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA"),
        ("1xCNN L=14", lambda: bm.CNN(14, 32, 100, "UGCA", seed=0), 14, "UGCA"),
        ("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA"),
        ("GE+MLP+CNN L=8", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(8, 100, "TGCA", seed=1), bm.MLP(8, 100, "TGCA", seed=2), bm.CNN(8, 32, 100, "TGCA", seed=3)]), 8, "TGCA")]
for name, make, L, alpha in fams:
    model = make()
    pool = synth.bytes_to_strings(synth.random_sequence_bytes(64, L, alpha, 3))
    print(f"== {name}: median (p99) us per get_fitness(list[str]) call: byte area / request line", flush=True)
    for n in (1, 2, 3, 6, 7, 20):
        row = []
        ref = None
        for tiny in (0, 1):
            eng.set_option("serve_tiny", tiny)
            row.append(med_us(lambda: model.get_fitness(pool[:n])))
            got = model.get_fitness(pool[:n])
            if ref is None: ref = got
            else: assert np.array_equal(ref, got), (name, n)
        mark = "" if n * L <= 48 else "   (more than 48 bytes: byte area either way)"
        print(f"   N={n:<3d} {row[0][0]:6.2f} ({row[0][1]:5.1f}) / {row[1][0]:6.2f} ({row[1][1]:5.1f}){mark}", flush=True)
print("fallbacks:", eng.get_option("server_fallbacks"))
