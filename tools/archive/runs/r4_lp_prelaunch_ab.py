"""Pre-launched instance of the layer-parallel protein form (lp_prelaunch = 1: after an explorer-size call the NEXT instance is
enqueued at once, fills its weights and waits for its request word) against a launch per call (0): latency, same bits, and the
situations in which the instance must step aside."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import population
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"
def med_us(fn, reps=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, float(np.percentile(ts, 99)) * 1e6
for L, M in ((237, 3), (237, 1), (90, 3), (237, 8)):
    members = [bm.CNN(L, 32, 100, AAS, seed=m) for m in range(M)]
    ens = flexs_amd.Ensemble(members) if M > 1 else members[0]
    ev = population.PopulationEvaluator(ens, AAS, L)
    rng = np.random.default_rng(0)
    print(f"== {M} x CNN(32,100) L={L} A=20: median (p99) us, launch per call / pre-launched", flush=True)
    for n in (1, 16, 40):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, 12))
        row, ref = [], None
        for pre in (0, 1):
            eng.set_option("lp_prelaunch", pre)
            s0 = eng.get_option("lp_armed_served")
            row.append(med_us(lambda: ens.get_fitness(seqs)))
            got = ens.get_fitness(seqs)
            if ref is None: ref = got
            else: assert np.array_equal(ref, got), (L, M, n)
            served = eng.get_option("lp_armed_served") - s0
        print(f"   N={n:<3d} {row[0][0]:6.1f} ({row[0][1]:6.1f}) / {row[1][0]:6.1f} ({row[1][1]:6.1f})   answered by a pre-launched instance: {served} of 331", flush=True)
    for P in (15, 40):
        x = rng.standard_normal((P, L * 20))
        row = []
        for pre in (0, 1):
            eng.set_option("lp_prelaunch", pre)
            row.append(med_us(lambda: ev.evaluate(x), 150))
        print(f"   P={P} step {row[0][0]:6.1f} ({row[0][1]:6.1f}) / {row[1][0]:6.1f} ({row[1][1]:6.1f})", flush=True)
eng.set_option("lp_prelaunch", 1)
# situations
members = [bm.CNN(90, 32, 100, AAS, seed=m) for m in range(3)]
ens = flexs_amd.Ensemble(members)
small = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
s1 = synth.bytes_to_strings(synth.random_sequence_bytes(40, 90, AAS, 1))
s8 = synth.bytes_to_strings(synth.random_sequence_bytes(20, 8, "TGCA", 1))
eng.set_option("lp_prelaunch", 0)
want = {n: ens.get_fitness(s1[:n]) for n in (1, 7, 16, 17, 40)}
want8 = small.get_fitness(s8)
eng.set_option("lp_prelaunch", 1)
bad = 0
for it in range(3000):
    n = (1, 7, 16, 17, 40)[it % 5] if it % 3 == 0 else 7                      # changing sizes: the instance of another size leaves
    bad += not np.array_equal(ens.get_fitness(s1[:n]), want[n])
    if it % 50 == 49: bad += not np.array_equal(small.get_fitness(s8), want8) # another ensemble (resident form) in between
    if it % 400 == 399: time.sleep(0.003)                                      # an idle gap: the instance has left by itself
    if it % 700 == 699:
        try:
            ens.get_fitness(s1[:6] + [s1[6][:-1] + "!"]); bad += 1
        except ValueError:
            pass
    if it % 1000 == 999: ens.train(s1, np.arange(40.0)); eng.set_option("lp_prelaunch", 0); want = {k: ens.get_fitness(s1[:k]) for k in want}; eng.set_option("lp_prelaunch", 1)
print(f"3000 mixed calls: {bad} wrong; answered by pre-launched instances so far: {eng.get_option('lp_armed_served')}", flush=True)
