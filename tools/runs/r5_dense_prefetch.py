"""MLP / GE launches that read their rows from host memory: the next tile's bytes asked for a tile ahead (dense_prefetch = 2: every such launch; straight into
a second LDS scratch) against at the start of each tile (0).  Same bits; get_fitness(list[str]) and fx_score on staged bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
AAS = s_utils.AAS
def med(f, n=21):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
cases = [("C3 MLP L=14", lambda s: bm.MLP(14, 100, "UGCA", seed=s), 1, 14, "UGCA", 100_000),
         ("C4 8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS, 100_000),
         ("3xMLP L=14", lambda s: bm.MLP(14, 100, "UGCA", seed=s), 3, 14, "UGCA", 100_003),
         ("3xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 3, 90, AAS, 70_001),
         ("MLP L=50", lambda s: bm.MLP(50, 100, "UGCA", seed=s), 1, 50, "UGCA", 100_000),
         ("8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS, 250_000)]
for tag, make, M, L, alpha, n in cases:
    mods = [make(s) for s in range(M)]
    model = mods[0] if M == 1 else flexs_amd.Ensemble(mods)
    nat = [m.native() for m in mods]; lut = mods[0]._lut
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
    other = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n + 1))
    res = {}
    for pf in (0, 2, 0, 2):
        eng.set_option("dense_prefetch", pf)
        model.get_fitness(other)                                  # (another batch in between: nothing may be left behind)
        got = np.asarray(model.get_fitness(seqs)).copy()
        b = _native.sequences_to_bytes(seqs, L=L, staging=eng)
        got_b = eng.score(nat, b, lut, want_matrix=(M == 1), want_mean=(M > 1))
        got_b = (got_b[0][:, 0] if M == 1 else got_b[1]).copy()
        t_b = med(lambda: eng.score(nat, b, lut, want_matrix=(M == 1), want_mean=(M > 1)))
        t_s = med(lambda: model.get_fitness(seqs))
        res.setdefault(pf, []).append((got, got_b, t_b, t_s))
    ref = res[0][0][0].view(np.uint32)
    same = all(np.array_equal(r[0].view(np.uint32), ref) and np.array_equal(r[1].view(np.uint32), ref) for v in res.values() for r in v)
    print(f"{tag} n={n}: same bits {same}; fx_score on staged bytes {res[0][0][2]:.0f} / {res[0][1][2]:.0f} -> {res[2][0][2]:.0f} / {res[2][1][2]:.0f} us; "
          f"get_fitness(list[str]) {res[0][0][3]:.0f} / {res[0][1][3]:.0f} -> {res[2][0][3]:.0f} / {res[2][1][3]:.0f} us", flush=True)
    assert same
eng.set_option("dense_prefetch", 1)
print("OK")
