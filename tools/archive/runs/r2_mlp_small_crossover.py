import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
for L, alpha, H in ((14, "UGCA", 100), (14, "UGCA", 200), (90, ps.AAS, 100)):
    for M, N in ((1, 6000), (1, 8192), (1, 12288), (1, 16384), (1, 32768), (3, 4000), (3, 8192)):
        for small in (2, 0):
            ps.time_score("mlp", L, alpha, H, M, N, reps=200, label=f"mlp L={L} H={H} M={M} N={N} dense_small={small}", opts={"dense_small": small})
