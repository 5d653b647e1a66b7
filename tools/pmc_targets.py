#!/usr/bin/env python3
"""Launch a few representative kernels (for rocprofv3 --pmc runs): GE M=8, MLP, pair-form CNN, dynamic CNN L=50."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "none"]
import tools.perf_survey as ps  # noqa: E402

AAS = ps.AAS
ps.time_score("ge", 90, AAS, 100, 8, 1_000_000, reps=3, label="ge M=8 N=1e6")
ps.time_score("mlp", 14, "UGCA", 100, 1, 1_000_000, reps=3, label="mlp N=1e6")
ps.time_score("cnn", 237, AAS, 100, 3, 16_384, 32, 5, reps=2, label="pair L=237 M=3 N=16384")
ps.time_score("cnn", 50, "UGCA", 100, 3, 100_000, 32, 5, reps=2, label="cnn L=50 M=3 N=1e5")
ps.time_score("cnn", 8, "TGCA", 100, 3, 1_000_000, 32, 5, reps=3, label="cnn L=8 M=3 N=1e6 (bench kernel)")
ps.time_score("mlp", 14, "UGCA", 200, 1, 1_000_000, reps=3, label="mlp H=200 N=1e6 (slab)")
