"""Round-2 A/B: explorer-size calls on long 4-letter sequences (RNA 50 / 100), position-segmented form over one or several workgroups."""
import sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import perf_survey as ps
import flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm

for L in (30, 50, 100, 200):
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, "UGCA", seed=m) for m in range(3)])
    for N in (1, 20, 100, 400):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, "UGCA", 3))
        ref = None
        for multi in (1, 0):
            ps.eng.set_option("cnn_seg_multi", multi)
            for _ in range(5): got = ens.get_fitness(seqs)
            ref = got if ref is None else ref
            assert np.array_equal(ref, got), (L, N)
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
            print({"what": f"Ensemble(3xCNN L={L} A=4).get_fitness N={N} cnn_seg_multi={multi}", "median_us": round(float(np.median(ts)) * 1e6, 1)}, flush=True)
        ps.eng.set_option("cnn_seg_multi", 1)
    for M, N in ((3, 1), (3, 20), (1, 20)):
        for multi in (1, 0):
            ps.time_score("cnn", L, "UGCA", 100, M, N, 32, 5, reps=300, label=f"kernel only: cnn L={L} A=4 M={M} N={N} cnn_seg_multi={multi}", opts={"cnn_seg_multi": multi})
        ps.eng.set_option("cnn_seg_multi", 1)
