"""Round-2 probe: segments per tile of the protein CNN's small-batch form (engine option cnn_pair_seg = workgroups per tile).
Kernel time from the C launch loop and the whole call through the plugin API."""
import sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import perf_survey as ps
import flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm

for L, alpha in ((237, ps.AAS), (90, ps.AAS)):
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)])
    for N in (1, 16, 100):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, alpha, 3))
        for sb in (-1, 2, 4, 8, 12, 16, 24, 32):
            ps.eng.set_option("cnn_pair_seg", sb)
            try:
                for _ in range(5): ens.get_fitness(seqs)
                ts = []
                for _ in range(60):
                    t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                print({"what": f"Ensemble(3xCNN L={L} A=20).get_fitness N={N} cnn_pair_seg={sb}", "median_us": round(float(np.median(ts)) * 1e6, 1)}, flush=True)
            except Exception as ex:
                print({"what": f"L={L} N={N} sb={sb} failed: {ex}"}, flush=True)
            finally:
                ps.eng.set_option("cnn_pair_seg", -1)
