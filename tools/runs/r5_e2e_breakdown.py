"""Where the wall time of get_fitness(list[str]) goes for C3 (MLP L=14, 1e5 strings): packing alone, the staged call alone, the whole call by
number of pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
print("cores:", os.cpu_count())
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for tag, make, L, alpha, n in (("C3 MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 100_000),
                               ("C2 1xCNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 100_000)):
    model = make(); eng = model._engine()
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    model.get_fitness(seqs)
    print(f"{tag}: whole call {med(lambda: model.get_fitness(seqs)):.0f} us")
    print(f"   pack into the pinned staging area: {med(lambda: _native.sequences_to_bytes(seqs, L=L, staging=eng)):.0f} us")
    b = _native.sequences_to_bytes(seqs, L=L, staging=eng)
    nm = model.native()
    if nm is not None:
        lut = model._lut
        print(f"   fx_score on staged bytes (kernel + wait + result copy): {med(lambda: eng.score([nm], b, lut, want_matrix=False, want_mean=True)):.0f} us")
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); eng.score([nm], b, lut, want_matrix=True); t1 = time.perf_counter()
            ts.append([eng.get_option(f"call_prof_{i}") / 1e3 for i in range(4)] + [(t1 - t0) * 1e6])
        print("   fx_score (matrix) us since entry: prepared %.1f, launched %.1f, results there %.1f, copied out %.1f; Python wall %.1f" % tuple(np.median(np.array(ts), axis=0)))
        import torch
        d_in = torch.from_numpy(np.ascontiguousarray(b)).cuda(); d_out = torch.empty((n, 1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        def dev_call():
            eng.score_dev([nm], d_in.data_ptr(), n, L, lut, d_out.data_ptr(), None); eng.sync()
        print(f"   the same launch on bytes already in HBM, results left in HBM, one isolated call + sync: {med(dev_call):.0f} us")
        sp = _native._strpack
        dst = np.empty((n, L), np.uint8)
        for thr in (1, 2, 4, 8, 12, 16):
            sp.set_threads(thr)
            print(f"   packing alone with {thr} thread(s): {med(lambda: sp.pack(seqs, L, dst, 0, n), 25):.0f} us")
        sp.set_threads(0)
        for chunks in (1, 2, 3, 4, 8):
            print(f"   score_strings in {chunks} piece(s): {med(lambda: eng.score_strings([nm], seqs, L, lut, want_matrix=False, want_mean=True, chunks=chunks)):.0f} us")
