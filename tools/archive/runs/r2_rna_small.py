"""Round-2 probe: kernel time of explorer-size calls on the 4-letter CNN ensemble (RNA lengths), launches issued from C."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

for L in (8, 14, 16):
    for M, N in ((3, 1), (3, 20), (3, 100), (1, 20), (1, 1000), (3, 1000), (1, 4096), (1, 8192), (1, 10_000), (3, 10_000)):
        for quad in (1, 0):
            ps.time_score("cnn", L, "UGCA", 100, M, N, 32, 5, reps=500, label=f"kernel only: cnn L={L} A=4 M={M} N={N} cnn_quad={quad}", opts={"cnn_quad": quad})
