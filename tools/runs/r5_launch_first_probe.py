"""Launched-first kernel alone: the staging area already holds the rows; publish every stage at once / after a pause / stage by stage
without packing anything -- what does the waiting itself cost?"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
sp = _native._strpack
eng = _native.Engine.get()
lib = eng._lib
def spin(us):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e6 < us: pass
for tag, model, L, alpha, n in (("1xCNN L=8", bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 100_000),):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    ref = np.asarray(model.get_fitness(seqs)).copy()
    nm = model.native(); lut = model._lut
    arr = (_native._vp * 1)(nm.handle)
    lanes = sp.lanes_for(n * L)
    eng.set_option("launch_first", 0); model.get_fitness(seqs); eng.set_option("launch_first", 1)   # (the staging area holds these rows)
    def call(mode):
        p, w, base, stages, brows = _native._vp(), _native._vp(), C.c_uint(0), C.c_int(0), C.c_int(0)
        t0 = time.perf_counter()
        rc = lib.fx_score_begin_staged(eng.handle, arr, 1, n, L, _native._lut_ptr(lut), 1, 0, lanes, C.byref(p), C.byref(w), C.byref(base), C.byref(stages), C.byref(brows), None, 0)
        assert rc == 0, rc
        words = (C.c_uint * 16).from_address(w.value)
        Q = stages.value
        if mode == "at once":
            for l in range(lanes): words[l] = base.value + Q
        elif mode == "after 40 us":
            spin(40)
            for l in range(lanes): words[l] = base.value + Q
        elif mode == "stage by stage, 7 us apart":
            for j in range(Q):
                spin(7)
                for l in range(lanes): words[l] = base.value + j + 1
        elif mode == "never (finish publishes)":
            pass
        t1 = time.perf_counter()
        out = np.empty((n, 1), np.float32)
        rc = lib.fx_score_finish(eng.handle, _native._ptr(out), None)
        t2 = time.perf_counter()
        assert rc == 0
        assert (out[:, 0].view(np.uint32) == ref.view(np.uint32)).all()
        return (t1 - t0) * 1e6, (t2 - t1) * 1e6
    for mode in ("at once", "stage by stage, 7 us apart", "never (finish publishes)"):
        r = np.median(np.array([call(mode) for _ in range(15)]), axis=0)
        print(f"{tag}, stages published {mode}: begin + publishing {r[0]:.0f} us, finish {r[1]:.0f} us, total {r[0] + r[1]:.0f} us")
    b = _native.sequences_to_bytes(seqs, L=L, staging=eng)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); eng.score([nm], b, lut, want_matrix=True); ts.append((time.perf_counter() - t0) * 1e6)
    print(f"{tag}, plain fx_score on the staged bytes: {np.median(ts):.0f} us")
