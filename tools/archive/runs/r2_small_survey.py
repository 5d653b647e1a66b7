"""Round-2 survey: kernel time of an explorer-size launch (N = 20, 3 members) for every model family / shape class."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps

AAS = ps.AAS
cases = [("cnn", 8, "TGCA", 100, 32, 5), ("cnn", 14, "UGCA", 100, 32, 5), ("cnn", 50, "UGCA", 100, 32, 5), ("cnn", 100, "UGCA", 100, 32, 5),
         ("cnn", 20, "UGCA", 100, 32, 5), ("cnn", 14, "UGCA", 100, 32, 3), ("cnn", 14, "UGCA", 100, 32, 7), ("cnn", 14, "UGCA", 100, 16, 5),
         ("cnn", 14, "UGCA", 64, 32, 5), ("cnn", 14, "UGCA", 200, 32, 5), ("cnn", 50, "UGCA", 100, 64, 5), ("cnn", 50, "UGCA", 100, 32, 4),
         ("cnn", 30, AAS, 100, 32, 5), ("cnn", 90, AAS, 100, 32, 5), ("cnn", 237, AAS, 100, 32, 5), ("cnn", 90, AAS, 100, 32, 3), ("cnn", 90, AAS, 200, 32, 5),
         ("cnn", 20, "01", 100, 32, 5),
         ("mlp", 8, "TGCA", 100, 0, 0), ("mlp", 14, "UGCA", 100, 0, 0), ("mlp", 50, "UGCA", 100, 0, 0), ("mlp", 100, "UGCA", 100, 0, 0), ("mlp", 14, "UGCA", 200, 0, 0),
         ("mlp", 90, AAS, 100, 0, 0), ("mlp", 237, AAS, 100, 0, 0),
         ("ge", 14, "UGCA", 100, 0, 0), ("ge", 90, AAS, 100, 0, 0), ("ge", 237, AAS, 100, 0, 0), ("ge", 90, AAS, 200, 0, 0)]
for kind, L, alpha, H, F, K in cases:
    for M, N in ((3, 20), (3, 400)):
        ps.time_score(kind, L, alpha, H, M, N, F, K, reps=300, label=f"{kind} L={L} A={len(alpha)} H={H} F={F} K={K} M={M} N={N}")
