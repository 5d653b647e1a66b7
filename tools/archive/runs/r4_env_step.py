"""DyNA-PPO's terminal environment step for a batch of 10 states (environments/dyna_ppo.py:144-163): decode + score (8 x GE L=90,
resident form) + record the batch + density-penalised rewards; per-sequence appends against one upload of the new keys."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils.edit_distance import SeenSequences
from flexs_amd.utils.population import PopulationEvaluator, terminal_rewards
AAS = "ILVAGMFYWEDQNHCRKSTP"
L, B = 90, 10
ens = flexs_amd.Ensemble([bm.GlobalEpistasisModel(L, 100, AAS, seed=m) for m in range(8)])
ev = PopulationEvaluator(ens, AAS, L)
rng = np.random.default_rng(0)
def states():
    st = np.zeros((B, L, 21)); codes = rng.integers(0, 20, (B, L))
    st[np.arange(B)[:, None], np.arange(L)[None, :], codes] = 1
    return st
for mode in ("one append per sequence", "one upload per batch"):
    seen = SeenSequences(L)
    if mode.startswith("one append"):
        class Plain:                                        # (hides add_many: the per-sequence loop of the reference)
            def __init__(self, s): self.s = s
            def add(self, *a): return self.s.add(*a)
            def densities(self, *a): return self.s.densities(*a)
        target = Plain(seen)
    else:
        target = seen
    for _ in range(100): terminal_rewards(ev, target, states(), 0.1)       # ~1000 recorded sequences
    ts = []
    for _ in range(200):
        st = states()
        t0 = time.perf_counter(); terminal_rewards(ev, target, st, 0.1); ts.append(time.perf_counter() - t0)
    print(f"{mode}: {np.median(ts) * 1e6:.1f} us per environment step of {B} sequences ({len(seen)} recorded)", flush=True)
# where the step goes (3000 recorded sequences)
st = states()
def t_us(fn, reps=100):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
seqs, fit = ev.evaluate(st[:, :, :-1])
print("  evaluate (decode + 8 x GE, resident): %.1f us" % t_us(lambda: ev.evaluate(st[:, :, :-1])))
q = _native.ragged_to_bytes(seqs, L)
print("  ragged_to_bytes of the 10 queries: %.1f us" % t_us(lambda: _native.ragged_to_bytes(seqs, L)))
print("  distance matrix 10 x %d (fx_cache_distances): %.1f us" % (len(seen), t_us(lambda: seen._cache.distances(q, seen._mode))))
print("  densities (distances + neighbour sums): %.1f us" % t_us(lambda: seen.densities(seqs)))
fresh = synth.bytes_to_strings(synth.random_sequence_bytes(10, L, AAS, 99))
print("  ragged_to_bytes + cache append of 10 new keys: %.1f us" % t_us(lambda: seen._cache.append(_native.ragged_to_bytes(fresh, L)), 30))
# the distance kernels alone: a block's cache rows through LDS on small launches (dist_stage = 1) against rows read from global memory (0)
eng = _native.Engine.get()
for Lq, C_, Q_ in ((90, 3000, 10), (90, 3000, 1), (14, 1000, 10), (14, 1000, 2000), (14, 100, 2000), (14, 20000, 2000), (150, 5000, 64), (237, 2000, 10)):
    alpha = AAS if Lq > 20 else "UGCA"
    cache = _native.NativeCache(eng, Lq)
    cache.append(synth.random_sequence_bytes(C_, Lq, alpha, 7))
    qq = synth.random_sequence_bytes(Q_, Lq, alpha, 8)
    row, keep = [], None
    for stage in (0, 1):
        eng.set_option("dist_stage", stage)
        ref = cache.distances(qq[:min(Q_, 32)]); md = cache.min_dist(qq)
        if keep is None: keep = (ref, md)
        else: assert np.array_equal(keep[0], ref) and np.array_equal(keep[1][0], md[0]) and np.array_equal(keep[1][1], md[1]), (Lq, C_, Q_)
        cache.time_min_dist(qq, reps=5)
        row.append((cache.time_min_dist(qq, reps=50) / 50 * 1e3, t_us(lambda: cache.distances(qq[:min(Q_, 10)]), 50)))
    print(f"  L={Lq} C={C_}: neighbour search of {Q_} queries (kernel, launches from C) {row[0][0]:.1f} -> {row[1][0]:.1f} us; distance matrix of {min(Q_, 10)} queries (host call) {row[0][1]:.1f} -> {row[1][1]:.1f} us", flush=True)
eng.set_option("dist_stage", 1)
