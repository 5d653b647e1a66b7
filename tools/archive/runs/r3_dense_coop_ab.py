"""Interleaved A/B of the shared last tiles of the persistent MLP / GE kernel (dense_coop), launches issued from C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.perf_survey as ps
AAS = "ILVAGMFYWEDQNHCRKSTP"
cases = [("mlp", 14, "UGCA", 100, 1, 100_000), ("mlp", 14, "UGCA", 100, 1, 98_304), ("mlp", 14, "UGCA", 100, 1, 104_096), ("mlp", 14, "UGCA", 100, 1, 108_192),
         ("mlp", 14, "UGCA", 100, 1, 1_000_000), ("mlp", 14, "UGCA", 100, 1, 20_000), ("mlp", 14, "UGCA", 100, 3, 100_000), ("mlp", 8, "TGCA", 100, 1, 100_000),
         ("ge", 90, AAS, 100, 1, 100_000), ("ge", 90, AAS, 100, 1, 98_304), ("ge", 90, AAS, 100, 8, 100_000), ("ge", 90, AAS, 100, 8, 1_000_000), ("ge", 14, "UGCA", 100, 1, 100_000)]
for rep in range(2):
    for kind, L, alpha, H, M, N in cases:
        for coop in ((2, 0) if kind == "ge" else (1, 0)):
            ps.time_score(kind, L, alpha, H, M, N, reps=200 if N <= 110_000 else 30, opts={"dense_coop": coop},
                          label=f"{kind} L={L} M={M} N={N} dense_coop={coop} [{rep}]")
        ps.eng.set_option("dense_coop", 1)
