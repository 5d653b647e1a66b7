"""Launched-first host calls (fx_score_begin_staged): same bits as the packed-first call, the counters say which path ran, the error
paths raise what the reference raises and leave the engine usable; wall time of get_fitness(list[str]) with the option on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm

def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6

eng = _native.Engine.get()
cases = [("1xCNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA"),
         ("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=s) for s in range(3)]), 8, "TGCA"),
         ("1xCNN L=14", lambda: bm.CNN(14, 32, 100, "UGCA", seed=0), 14, "UGCA"),
         ("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA")]
for tag, make, L, alpha in cases:
    model = make()
    for n in (100_000, 100_003, 40_000, 250_000):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
        eng.set_option("launch_first", 0)
        ref = np.asarray(model.get_fitness(seqs)).copy()
        t_off = med(lambda: model.get_fitness(seqs))
        eng.set_option("launch_first", 1)
        c0, r0 = eng.get_option("launch_first_calls"), eng.get_option("launch_first_redone")
        got = np.asarray(model.get_fitness(seqs)).copy()
        took = eng.get_option("launch_first_calls") - c0
        same = bool((got.view(np.uint32) == ref.view(np.uint32)).all())
        t_on = med(lambda: model.get_fitness(seqs))
        print(f"{tag} n={n}: launched first {bool(took)}, redone {eng.get_option('launch_first_redone') - r0}, same bits {same}; "
              f"{t_off:.0f} us packed first, {t_on:.0f} us {'launched first' if took else '(same path)'}", flush=True)
        assert same
# error paths on a shape that launches first
model = bm.CNN(8, 32, 100, "TGCA", seed=0)
n = 100_000
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, 8, "TGCA", 5))
ref = np.asarray(model.get_fitness(seqs)).copy()
for what, mutate, exc in (("ragged", lambda s: s.__setitem__(n // 2, "ACGTACGTA"), ValueError),
                          ("bad letter", lambda s: s.__setitem__(n - 7, "ACGTACGX"), ValueError),
                          ("not a str", lambda s: s.__setitem__(17, 5), (TypeError, ValueError))):
    bad = list(seqs); mutate(bad)
    try:
        model.get_fitness(bad)
        print(what, "-> no exception"); raise SystemExit(1)
    except exc as ex:
        print(what, "->", type(ex).__name__, ex)
    again = np.asarray(model.get_fitness(seqs))
    assert (again.view(np.uint32) == ref.view(np.uint32)).all()
print("launch_first_calls", eng.get_option("launch_first_calls"), "redone", eng.get_option("launch_first_redone"))
print("OK")
