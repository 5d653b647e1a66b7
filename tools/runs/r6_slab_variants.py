#!/usr/bin/env python3
"""Round 6: kernel time + output hash of the slab form (H > 128) for ONE library build (FLEXS_AMD_LIB selects it; the shell loop in
tools/gpu_r6_s11.sh runs every variant): slabs through registers (rounds 1-5) against direct global -> LDS copies streamed across layers,
KG = 2 / 3 / 4 input tiles per slab.  -> profiles/r6_slab_dma_ab.log"""
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
CASES = [("mlp H=200 L=14 N=1e5", "mlp", 14, "UGCA", 200, 1, 100_000), ("mlp H=200 L=14 N=1e6", "mlp", 14, "UGCA", 200, 1, 1_000_000),
         ("mlp H=256 L=14 N=1e5", "mlp", 14, "UGCA", 256, 1, 100_000), ("3 x mlp H=200 L=14 N=1e5", "mlp", 14, "UGCA", 200, 3, 100_000),
         ("mlp H=200 L=40 N=1e5", "mlp", 40, "UGCA", 200, 1, 100_000), ("mlp H=160 L=14 N=1e5", "mlp", 14, "UGCA", 160, 1, 100_000),
         ("ge H=200 L=90 N=1e5", "ge", 90, AAS, 200, 1, 100_000), ("8 x ge H=200 L=90 N=1e5", "ge", 90, AAS, 200, 8, 100_000),
         ("mlp H=200 L=14 N=777", "mlp", 14, "UGCA", 200, 2, 777)]
for name, kind, L, alpha, H, M, n in CASES:
    mods = build_members(kind, L, alpha, M, 0, Hx=H)
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, 0)).cuda()
    stride = (n + 63) // 64 * 64
    plane = torch.zeros((M, stride), dtype=torch.float32, device="cuda")
    res = []
    for rep in range(3):
        ms, _ = time_launches(eng, mods, d_in.data_ptr(), n, L, mods[0]._lut, plane, stride, min_ms=40.0)
        res.append(ms * 1e3)
    torch.cuda.synchronize()
    h = hashlib.sha1(plane[:, :n].cpu().numpy().tobytes()).hexdigest()[:12]
    med = float(np.median(res))
    fr = roofline_block(kind, L, len(alpha), H, 0, 0, M, n, med * 1e-3, "k")["frac"]
    print(f"{tag:28s} {name:28s} {med:9.2f} us ({fr:.3f})  bits {h}", flush=True)
