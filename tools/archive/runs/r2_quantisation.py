"""Round-2 probe: how much of the bench launch (3 x CNN, L = 8) is SIMD-level tile quantisation?  18 750 (member, tile) units on
1024 SIMDs are 18.31 per SIMD, i.e. 19 on the busiest; sizes whose units divide evenly show the cost per tile round."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

for rnd in range(2):
    for N in (87_376, 92_832, 98_304, 98_320, 99_000, 100_000, 103_744, 109_216):
        ps.time_score("cnn", 8, "TGCA", 100, 3, N, 32, 5, reps=300, label=f"cnn L=8 M=3 N={N} units/SIMD={3 * ((N + 15) // 16) / 1024:.3f} [{rnd}]")
