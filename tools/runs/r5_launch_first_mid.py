"""Launched-first calls below 32768 strings (FLEXS_AMD_CHUNKED_MIN_ROWS lowered): where does it start to pay?"""
import os, sys, time
os.environ["FLEXS_AMD_CHUNKED_MIN_ROWS"] = "4097"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
def med(f, n=21):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for tag, make, L, alpha in (("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=s) for s in range(3)]), 8, "TGCA"),
                            ("1xCNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA"),
                            ("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA"),
                            ("3xMLP L=14", lambda: flexs_amd.Ensemble([bm.MLP(14, 100, "UGCA", seed=s) for s in range(3)]), 14, "UGCA")):
    model = make()
    for n in (6000, 10000, 16384, 24000, 32000):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
        eng.set_option("launch_first", 0)
        ref = np.asarray(model.get_fitness(seqs)).copy()
        t_off = med(lambda: model.get_fitness(seqs))
        eng.set_option("launch_first", 1)
        c0 = eng.get_option("launch_first_calls")
        got = np.asarray(model.get_fitness(seqs)).copy()
        took = eng.get_option("launch_first_calls") - c0
        assert (got.view(np.uint32) == ref.view(np.uint32)).all()
        t_on = med(lambda: model.get_fitness(seqs))
        print(f"{tag} n={n}: launched first {bool(took)}; {t_off:.0f} us packed first, {t_on:.0f} us with the option on", flush=True)
