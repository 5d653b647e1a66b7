import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_survey as ps
for L in (17, 18, 20, 24, 27):
    for M, N in ((3, 20), (3, 400), (1, 2000)):
        for multi in (1, 0):
            ps.time_score("cnn", L, "UGCA", 100, M, N, 32, 5, reps=300, label=f"cnn L={L} M={M} N={N} cnn_seg_multi={multi}", opts={"cnn_seg_multi": multi})
