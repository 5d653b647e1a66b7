"""Round-2 A/B on ONE box: direct global -> LDS weight copies in the quad kernel (engine option dma_fill), alternated
three times per size so that box-to-box and clock drift cancel.  Launches are issued from C (fx_debug_time_score)."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

for rnd in range(3):
    for M, N in ((1, 10_000), (1, 4_000), (1, 1_000), (3, 1_000), (1, 12_000)):
        for dma in (1, 0):
            ps.time_score("cnn", 8, "TGCA", 100, M, N, 32, 5, reps=2000, label=f"cnn L=8 M={M} N={N} dma_fill={dma} [{rnd}]", opts={"dma_fill": dma})
for rnd in range(2):
    for q in (1, 0):
        ps.time_score("cnn", 8, "TGCA", 100, 1, 10_000, 32, 5, reps=2000, label=f"cnn L=8 M=1 N=10000 cnn_quad={q} [{rnd}]", opts={"cnn_quad": q})
    ps.time_score("cnn", 237, ps.AAS, 100, 3, 62_500, 32, 5, reps=5, label=f"cnn L=237 M=3 N=62500 [{rnd}]")
    ps.time_score("mlp", 14, "UGCA", 100, 1, 100_000, reps=500, label=f"mlp L=14 N=1e5 [{rnd}]")
