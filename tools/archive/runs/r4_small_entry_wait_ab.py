"""Explorer-size calls of the small entry points (edit distances, NoisyAbstractModel query over a device table, table look-up,
population step with the device argmax): completion value polled in pinned memory (done_flag = 1) against hipStreamSynchronize (0)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import edit_distance, population
eng = _native.Engine.get()
def med_us(fn, reps=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6
rows = {}
seen = edit_distance.SeenSequences(14)
pool = synth.bytes_to_strings(synth.random_sequence_bytes(2000, 14, "UGCA", 3))
for s_ in pool[:1000]: seen.add(s_, 0.0)
q = pool[1000:1010]
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
ev = population.PopulationEvaluator(ens, "TGCA", 8)
x = np.random.default_rng(0).standard_normal((20, 32))
for flag in (0, 1):
    eng.set_option("done_flag", flag)
    rows.setdefault("sequence_density of 10 queries vs 1000 seen (L=14)", []).append(med_us(lambda: seen.densities(q)))
    population.HOST_DECODE = False
    rows.setdefault("population step P=20 L=8, device argmax (fx_decode_score)", []).append(med_us(lambda: ev.evaluate(x)))
    population.HOST_DECODE = True
eng.set_option("done_flag", 1)
for k, v in rows.items():
    print(f"{k}: {v[0]:.1f} -> {v[1]:.1f} us", flush=True)
