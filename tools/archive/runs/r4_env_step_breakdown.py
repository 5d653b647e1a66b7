"""Breakdown of DyNA-PPO's terminal environment step inside its loop (slice / evaluate / add_many / densities / rewards).
NOTE: do not run this (or anything that keeps resident workgroups / pre-launched instances alive) under `rocprofv3 --kernel-trace`:
in round 4 that combination never returned and cost 15 GPU-minutes; set serve_small = 0 and lp_prelaunch = 0 first when profiling."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils.edit_distance import SeenSequences
from flexs_amd.utils.population import PopulationEvaluator
AAS = "ILVAGMFYWEDQNHCRKSTP"
L, B = 90, 10
ens = flexs_amd.Ensemble([bm.GlobalEpistasisModel(L, 100, AAS, seed=m) for m in range(8)])
ev = PopulationEvaluator(ens, AAS, L)
rng = np.random.default_rng(0)
def states():
    st = np.zeros((B, L, 21)); codes = rng.integers(0, 20, (B, L))
    st[np.arange(B)[:, None], np.arange(L)[None, :], codes] = 1
    return st
seen = SeenSequences(L)
acc = np.zeros(5); n = 0
for it in range(300):
    st = states()
    t0 = time.perf_counter()
    x = np.asarray(st, np.float64)[:, :, :-1]
    t1 = time.perf_counter()
    seqs, fit = ev.evaluate(x)
    t2 = time.perf_counter()
    seen.add_many(seqs, fit)
    t3 = time.perf_counter()
    dens = seen.densities(seqs)
    t4 = time.perf_counter()
    rew = np.array([f - 0.1 * d for f, d in zip(fit, dens)])
    t5 = time.perf_counter()
    if it >= 100:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]; n += 1
print("slice %.1f, evaluate %.1f, add_many %.1f, densities %.1f, rewards %.1f us; total %.1f" % (*(acc / n * 1e6), acc.sum() / n * 1e6))
# the same loop with a launch per scoring call (no resident workgroups beside the distance kernel)
eng = _native.Engine.get()
eng.set_option("serve_small", 0)
acc = np.zeros(5); n = 0
for it in range(200):
    st = states()
    t0 = time.perf_counter(); x = np.asarray(st, np.float64)[:, :, :-1]
    t1 = time.perf_counter(); seqs, fit = ev.evaluate(x)
    t2 = time.perf_counter(); seen.add_many(seqs, fit)
    t3 = time.perf_counter(); dens = seen.densities(seqs)
    t4 = time.perf_counter(); rew = np.array([f - 0.1 * d for f, d in zip(fit, dens)])
    t5 = time.perf_counter()
    if it >= 50: acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]; n += 1
eng.set_option("serve_small", 1)
print("serve_small = 0: slice %.1f, evaluate %.1f, add_many %.1f, densities %.1f, rewards %.1f us; total %.1f" % (*(acc / n * 1e6), acc.sum() / n * 1e6))
# densities right after an add_many, no scoring call in between
acc2 = 0.0
for it in range(100):
    fresh = synth.bytes_to_strings(synth.random_sequence_bytes(10, L, AAS, 1000 + it))
    seen.add_many(fresh, [0.5] * 10)
    t0 = time.perf_counter(); seen.densities(fresh); acc2 += time.perf_counter() - t0
print("densities right behind add_many (no scoring call): %.1f us" % (acc2 / 100 * 1e6))
