"""Round-2 probe 1: where does the time go at small / medium N?  (GPU box only)
K1 at C1/C2 sizes with the 16-wave unrolled kernel forced (cnn_big_units = 1) vs the default 8-wave kernel,
an N scan to separate fixed cost / tile quantisation / steady state, and the GE / MLP configs at several N."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

AAS = ps.AAS
for M, N in ((1, 10_000), (3, 10_000), (1, 4_000), (1, 16_384), (1, 32_768), (3, 30_000)):
    for bu in (12, 1):
        ps.time_score("cnn", 8, "TGCA", 100, M, N, 32, 5, reps=200, label=f"cnn L=8 M={M} N={N} big_units={bu}", opts={"cnn_big_units": bu})
for N in (1_000, 100_000, 200_000, 400_000, 1_000_000):
    ps.time_score("ge", 90, AAS, 100, 8, N, reps=50, label=f"ge L=90 M=8 N={N}")
for N in (100_000, 200_000, 400_000):
    ps.time_score("ge", 90, AAS, 100, 1, N, reps=100, label=f"ge L=90 M=1 N={N}")
for N in (1_000, 100_000, 200_000, 400_000, 1_000_000):
    ps.time_score("mlp", 14, "UGCA", 100, 1, N, reps=100, label=f"mlp L=14 M=1 N={N}")
