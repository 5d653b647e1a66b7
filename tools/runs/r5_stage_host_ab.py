"""Zero-copy host calls of the 4-letter CNN: a tile's bytes copied into LDS with one wide load (cnn_stage_host = 1) against a byte load
over PCIe per position (0); FX_AB_PAIR=1,2: against the same with the next tile's bytes asked for a tile ahead (2).  Same bits; wall time of fx_score on bytes in the pinned staging area and of get_fitness(list[str])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for tag, M, L, alpha, n in (("1xCNN L=8", 1, 8, "TGCA", 100_000), ("3xCNN L=8", 3, 8, "TGCA", 100_000), ("1xCNN L=14", 1, 14, "UGCA", 100_000),
                            ("3xCNN L=50", 3, 50, "UGCA", 40_000), ("3xCNN L=8", 3, 8, "TGCA", 6_000), ("1xCNN L=8", 1, 8, "TGCA", 20_003)):
    members = [bm.CNN(L, 32, 100, alpha, seed=s) for s in range(M)]
    model = members[0] if M == 1 else flexs_amd.Ensemble(members)
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    res = {}
    A, B = [int(x) for x in os.environ.get("FX_AB_PAIR", "0,1").split(",")]
    for opt in (A, B):
        eng.set_option("cnn_stage_host", opt)
        got = np.asarray(model.get_fitness(seqs)).copy()
        b = _native.sequences_to_bytes(seqs, L=L, staging=eng)
        nat = [m.native() for m in members]
        t_score = med(lambda: eng.score(nat, b, members[0]._lut, want_matrix=(M == 1), want_mean=(M > 1)))
        t_call = med(lambda: model.get_fitness(seqs))
        res[opt] = (got, t_score, t_call)
    same = bool((res[A][0].view(np.uint32) == res[B][0].view(np.uint32)).all())
    print(f"{tag} n={n}: same bits {same}; cnn_stage_host {A} -> {B}: fx_score on staged bytes {res[A][1]:.0f} -> {res[B][1]:.0f} us; get_fitness(list[str]) {res[A][2]:.0f} -> {res[B][2]:.0f} us", flush=True)
    assert same
eng.set_option("cnn_stage_host", 1)
