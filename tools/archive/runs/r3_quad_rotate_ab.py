"""A/B of the rotated wave roles in the quad CNN form (quad_rotate), launches issued from C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.perf_survey as ps
cases = [("cnn", 8, "TGCA", 100, 1, 10_000), ("cnn", 8, "TGCA", 100, 3, 10_000), ("cnn", 8, "TGCA", 100, 3, 4_000), ("cnn", 8, "TGCA", 100, 1, 4_000),
         ("cnn", 8, "TGCA", 100, 3, 2_000), ("cnn", 14, "UGCA", 100, 3, 4_000)]
for rep in range(3):
    for kind, L, alpha, H, M, N in cases:
        for rot in (1, 0):
            ps.time_score(kind, L, alpha, H, M, N, F=32, K=5, reps=300, opts={"quad_rotate": rot}, label=f"{kind} L={L} M={M} N={N} quad_rotate={rot} [{rep}]")
        ps.eng.set_option("quad_rotate", 1)
