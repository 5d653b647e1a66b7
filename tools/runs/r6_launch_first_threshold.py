import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import flexs_amd
from flexs_amd import synth
from tools.bench_common import AAS, build_members
for kind, L, alpha, H, M in (("ge", 8, "TGCA", 100, 1), ("mlp", 14, "UGCA", 100, 1), ("cnn", 8, "TGCA", 100, 1), ("cnn", 8, "TGCA", 100, 3), ("mlp", 14, "UGCA", 100, 3), ("ge", 90, AAS, 100, 1), ("cnn", 14, "UGCA", 100, 1)):
    mods = build_members(kind, L, alpha, M, 0, Hx=H)
    m = mods[0] if M == 1 else flexs_amd.Ensemble(mods)
    row = []
    for n in (16384, 20000, 30000, 50000):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
        for _ in range(3): m.get_fitness(seqs)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); m.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        row.append(f"N={n}: {np.median(ts)*1e6:7.1f}")
    print(f"{os.environ.get('FLEXS_AMD_CHUNKED_MIN_ROWS','16384'):>7s} {kind} L={L} M={M}: " + "  ".join(row), flush=True)
