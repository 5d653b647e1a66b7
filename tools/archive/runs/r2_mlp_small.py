"""Round-2 A/B: explorer-size MLP launches, small-launch form (dense_small) vs the persistent kernel; launches issued from C."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

for L, alpha, H in ((8, "TGCA", 100), (14, "UGCA", 100), (50, "UGCA", 100), (100, "UGCA", 100), (14, "UGCA", 200), (90, ps.AAS, 100), (237, ps.AAS, 100), (90, ps.AAS, 200)):
    for M, N in ((3, 20), (3, 400), (1, 4000)):
        for small in (1, 0):
            ps.time_score("mlp", L, alpha, H, M, N, reps=300, label=f"mlp L={L} A={len(alpha)} H={H} M={M} N={N} dense_small={small}", opts={"dense_small": small})

for L, alpha, H in ((14, "UGCA", 100), (90, ps.AAS, 100), (237, ps.AAS, 100), (90, ps.AAS, 200)):
    for M, N in ((3, 20), (3, 400), (1, 4000)):
        for small in (1, 0):
            ps.time_score("ge", L, alpha, H, M, N, reps=300, label=f"ge L={L} A={len(alpha)} H={H} M={M} N={N} dense_small={small}", opts={"dense_small": small})
