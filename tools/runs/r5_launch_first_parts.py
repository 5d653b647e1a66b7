"""Where a launched-first call spends its time: begin (enqueue), staged packing, finish (wait + copy); against the plain packing."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
sp = _native._strpack
eng = _native.Engine.get()
lib = eng._lib
for tag, model, L, alpha, n in (("1xCNN L=8", bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA", 100_000),
                                ("1xCNN L=14", bm.CNN(14, 32, 100, "UGCA", seed=0), 14, "UGCA", 100_000)):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    model.get_fitness(seqs)
    nm = model.native(); lut = model._lut
    arr = (_native._vp * 1)(nm.handle)
    pitch = (16 * L + 127) // 128 * 128
    dst = np.empty((n, L), np.uint8); dst_t = np.empty(((n + 15) // 16) * pitch, np.uint8); words = np.zeros(16, np.uint32)
    lanes = sp.lanes_for(n * L)
    def t(f, k=15):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        return np.median(ts) * 1e6
    print(tag, "lanes", lanes)
    print("  plain pack into numpy: %.0f us" % t(lambda: sp.pack(seqs, L, dst, 0, n)))
    for Q in (1, 2, 6, 24):
        print("  staged pack, %d stages, words in host memory: %.0f us" % (Q, t(lambda: sp.pack_staged(seqs, L, dst_t.ctypes.data, Q, pitch, lanes, words.ctypes.data, 4096))))
    rows = []
    for _ in range(15):
        p, w, base, stages, brows = _native._vp(), _native._vp(), C.c_uint(0), C.c_int(0), C.c_int(0)
        t0 = time.perf_counter()
        rc = lib.fx_score_begin_staged(eng.handle, arr, 1, n, L, _native._lut_ptr(lut), 1, 0, lanes, C.byref(p), C.byref(w), C.byref(base), C.byref(stages), C.byref(brows), None, 0)
        assert rc == 0, rc
        t1 = time.perf_counter()
        st = sp.pack_staged(seqs, L, p.value, stages.value, brows.value, lanes, w.value, base.value)
        t2 = time.perf_counter()
        out = np.empty((n, 1), np.float32)
        rc = lib.fx_score_finish(eng.handle, _native._ptr(out), None)
        t3 = time.perf_counter()
        assert rc == 0 and st == 0
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    r = np.median(np.array(rows), axis=0)
    print("  launched first (%d stages): begin %.0f us, staged pack %.0f us, finish %.0f us" % (stages.value, r[0], r[1], r[2]))
    # the same staged packing into the pinned area with the words in device memory, no kernel running
    p2 = _native.sequences_to_bytes(seqs[:16], L=L, staging=eng)
    print("  staged pack into the pinned area, words behind the BAR, GPU idle: %.0f us" % t(lambda: sp.pack_staged(seqs, L, p.value, stages.value, brows.value, lanes, w.value, base.value - 4096)))
    print("  plain pack into the pinned area: %.0f us" % t(lambda: _native.sequences_to_bytes(seqs, L=L, staging=eng)))
