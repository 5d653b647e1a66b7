import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"
def med_us(fn, reps=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6
fams = [("3xCNN L=8 launch per call (serve_small=0)", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", {"serve_small": 0}),
        ("MLP H=300 L=14 (no resident form)", lambda: bm.MLP(14, 300, "UGCA", seed=0), 14, "UGCA", {}),
        ("8xCNN L=237 N=40 (not layer-parallel)", lambda: flexs_amd.Ensemble([bm.CNN(237, 32, 100, AAS, seed=m) for m in range(8)]), 237, AAS, {})]
for name, make, L, alpha, opts in fams:
    model = make()
    pool = synth.bytes_to_strings(synth.random_sequence_bytes(64, L, alpha, 3))
    for k, v in opts.items(): eng.set_option(k, v)
    row = []
    for n in (1, 40):
        r = []
        for flag in (0, 1):
            eng.set_option("done_flag", flag)
            r.append(med_us(lambda: model.get_fitness(pool[:n])))
            got = model.get_fitness(pool[:n])
        row.append(r)
    for k in opts: eng.set_option(k, 1)
    print(f"{name}: N=1 {row[0][0]:.1f} -> {row[0][1]:.1f} us, N=40 {row[1][0]:.1f} -> {row[1][1]:.1f} us (hipStreamSynchronize -> completion value / flag)", flush=True)
eng.set_option("done_flag", 1)
