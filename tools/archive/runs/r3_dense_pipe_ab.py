"""Interleaved A/B of the software-pipelined MLP / GE form (dense_pipe) against round 2's, launches issued from C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.perf_survey as ps
AAS = "ILVAGMFYWEDQNHCRKSTP"
cases = [("mlp", 14, "UGCA", 100, 1, 100_000), ("mlp", 14, "UGCA", 100, 1, 1_000_000), ("mlp", 14, "UGCA", 100, 1, 20_000),
         ("mlp", 14, "UGCA", 100, 3, 100_000), ("mlp", 8, "TGCA", 100, 1, 100_000),
         ("ge", 90, AAS, 100, 1, 100_000), ("ge", 90, AAS, 100, 8, 100_000), ("ge", 90, AAS, 100, 8, 1_000_000), ("ge", 14, "UGCA", 100, 1, 100_000)]
for rep in range(2):
    for kind, L, alpha, H, M, N in cases:
        for pipe in (2, 1, 0):
            ps.time_score(kind, L, alpha, H, M, N, reps=200 if N <= 100_000 else 30, opts={"dense_pipe": pipe},
                          label=f"{kind} L={L} M={M} N={N} dense_pipe={pipe} [{rep}]")
        ps.eng.set_option("dense_pipe", 0)
