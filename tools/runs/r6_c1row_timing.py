#!/usr/bin/env python3
"""Round 6: kernel times of the CNN kernels for ONE library build (before / after the conv1 gather rows were padded to 36 floats, FX_C1_ROW):
run once per build (FLEXS_AMD_LIB), compare the lines.  -> profiles/r6_c1row_ab.log"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import AAS, build_members, roofline_block, time_launches  # noqa: E402

eng = _native.Engine.get(0)
tag = os.path.basename(os.environ.get("FLEXS_AMD_LIB", "libflexs_amd.so"))
for name, L, alpha, M, n in (("3xCNN L=8 N=1e5 (headline)", 8, "TGCA", 3, 100_000), ("1xCNN L=14 N=1e5", 14, "UGCA", 1, 100_000), ("3xCNN L=8 N=1e6", 8, "TGCA", 3, 1_000_000),
                             ("1xCNN AAV L=90 N=1e5", 90, AAS, 1, 100_000), ("3xCNN GFP L=237 N=62500 (C5)", 237, AAS, 3, 62_500), ("1xCNN L=8 N=1e4 (C1)", 8, "TGCA", 1, 10_000),
                             ("1xCNN RNA L=100 N=1e5", 100, "UGCA", 1, 100_000)):
    mods = build_members("cnn", L, alpha, M, 0)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = torch.zeros((M, stride), dtype=torch.float32, device="cuda")
    us = float(np.median([time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, planes, stride, min_ms=60.0)[0] * 1e3 for _ in range(3)]))
    torch.cuda.synchronize()
    h = hashlib.sha1(planes[:, :n].cpu().numpy().tobytes()).hexdigest()[:12]
    fr = roofline_block("cnn", L, len(alpha), 100, 32, 5, M, n, us * 1e-3, "k")["frac"]
    print(f"{tag:26s} {name:32s} {us:10.2f} us ({fr:.3f})  bits {h}", flush=True)
