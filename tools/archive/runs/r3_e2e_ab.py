"""End-to-end get_fitness(list[str]) for C2/C3/C4: one stream vs two streams (chunk_overlap), piece sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get(0)

def run(tag, model, L, alpha, n):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    import ctypes as C
    mods = model.models if hasattr(model, "models") else [model]
    arr = (C.c_void_p * len(mods))(*[m.native().handle for m in mods])
    zc, pc = C.c_int(0), C.c_int(0)
    eng._lib.fx_plan_host_call(eng.handle, arr, len(mods), n, L, C.byref(zc), C.byref(pc))
    print(f"{tag}: plan zero_copy={zc.value} pieces={pc.value}", flush=True)
    ref = None
    for name, mode, cb in (("auto plan", -1, 0), ("always copy, one piece", 0, 1 << 40), ("always copy, 2 pieces", 0, n * L // 2),
                           ("always zero-copy, one piece", 1, 1 << 40), ("always zero-copy, 2 pieces", 1, n * L // 2),
                           ("always zero-copy, 4 pieces", 1, n * L // 4)):
        eng.set_option("zero_copy_mode", mode); _native.CHUNK_BYTES = cb
        out = model.get_fitness(seqs)
        if ref is None:
            ref = out
        assert np.array_equal(out, ref)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        print(f"{tag} n={n} {name}: {np.median(ts) * 1e3:.3f} ms  (min {min(ts) * 1e3:.3f})", flush=True)
    eng.set_option("zero_copy_mode", -1); _native.CHUNK_BYTES = 0

run("C2 3xCNN L=8", flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 100_000)
run("C3 MLP L=14", bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 100_000)
run("C4 8xGE L=90", flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=m) for m in range(8)]), 90, s_utils.AAS, 100_000)
run("C5 3xCNN L=237", flexs_amd.Ensemble([bm.CNN(237, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 237, s_utils.AAS, 62_500)
run("C4 8xGE L=90 1e6", flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=m) for m in range(8)]), 90, s_utils.AAS, 1_000_000)
run("3xCNN L=14", flexs_amd.Ensemble([bm.CNN(14, 32, 100, "UGCA", seed=m) for m in range(3)]), 14, "UGCA", 100_000)
run("MLP L=90 A=20", bm.MLP(90, 100, s_utils.AAS, seed=0), 90, s_utils.AAS, 100_000)
run("C2 3xCNN L=8 1e6", flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1_000_000)
