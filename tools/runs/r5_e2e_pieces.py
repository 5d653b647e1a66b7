"""get_fitness(list[str]) wall time of C3 (MLP L=14, 1e5) and C4 (8 x GE L=90, 1e5) against the number of pieces the strings are packed and
submitted in (FLEXS_AMD_CHUNK_BYTES is read at import: one process per setting; this script is the child)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
rows = (("C3 MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA", 100_000),
        ("C4 8xGE L=90", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=m) for m in range(8)]), 90, s_utils.AAS, 100_000),
        ("C2 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 100_000))
for tag, make, L, alpha, n in rows:
    model = make()
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
    model.get_fitness(seqs)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
    print(f"CHUNK_BYTES={os.environ.get('FLEXS_AMD_CHUNK_BYTES', 'plan')} {tag}: median {np.median(ts) * 1e6:.0f} us, min {min(ts) * 1e6:.0f} us", flush=True)
