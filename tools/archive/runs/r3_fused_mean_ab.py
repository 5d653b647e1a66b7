"""Small-call latency with and without the in-kernel ensemble mean (fuse_mean)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
for L, alpha in ((8, "TGCA"), (14, "UGCA")):
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)])
    for n in (1, 20, 48, 100):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 2))
        for rep in range(2):
            for fuse in (1, 0):
                eng.set_option("fuse_mean", fuse)
                for _ in range(200): ens.get_fitness(seqs)
                ts = []
                for _ in range(3000):
                    t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                print(f"3xCNN L={L} N={n} fuse_mean={fuse} [{rep}]: {np.median(ts) * 1e6:.1f} us", flush=True)
eng.set_option("fuse_mean", 0)
import random
from flexs_amd.utils import rollouts
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
for fuse in (1, 0, 1, 0):
    eng.set_option("fuse_mean", fuse)
    ts = []
    for i in range(5):
        random.seed(1)
        t0 = time.perf_counter()
        rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet="TGCA")
        ts.append(time.perf_counter() - t0)
    print(f"Adalead round (2000 queries) fuse_mean={fuse}: {min(ts) * 1e3:.2f} ms", flush=True)
eng.set_option("fuse_mean", 0)
