"""Round-2 A/B: 4-wave workgroups for the segmented protein form (engine option cnn_pair_seg4), same box."""
import sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import perf_survey as ps
import flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm

for L in (237, 90, 30):
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, ps.AAS, seed=m) for m in range(3)])
    for N in (1, 16, 40, 100):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, ps.AAS, 3))
        ref = None
        for s4 in (1, 0, 1, 0):
            ps.eng.set_option("cnn_pair_seg4", s4)
            for _ in range(5): got = ens.get_fitness(seqs)
            ref = got if ref is None else ref
            assert np.array_equal(ref, got)
            ts = []
            for _ in range(100):
                t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
            print({"what": f"Ensemble(3xCNN L={L} A=20).get_fitness N={N} cnn_pair_seg4={s4}", "median_us": round(float(np.median(ts)) * 1e6, 1)}, flush=True)
        ps.eng.set_option("cnn_pair_seg4", 1)
    for M, N in ((3, 1), (3, 16), (1, 16)):
        for s4 in (1, 0):
            ps.time_score("cnn", L, ps.AAS, 100, M, N, 32, 5, reps=300, label=f"kernel only: cnn L={L} A=20 M={M} N={N} cnn_pair_seg4={s4}", opts={"cnn_pair_seg4": s4})
        ps.eng.set_option("cnn_pair_seg4", 1)
