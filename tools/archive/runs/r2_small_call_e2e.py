import sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm
for L, alpha in ((8, "TGCA"), (14, "UGCA"), (50, "UGCA"), (100, "UGCA")):
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)])
    for N in (1, 20, 100):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, alpha, 3))
        for _ in range(20): ens.get_fitness(seqs)
        ts = []
        for _ in range(500):
            t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        print({"what": f"Ensemble(3xCNN L={L}).get_fitness(list[str]) N={N}", "median_us": round(float(np.median(ts)) * 1e6, 1)}, flush=True)
