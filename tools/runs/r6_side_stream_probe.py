#!/usr/bin/env python3
"""Round 6 probe (timing only, NOT a product path): would the ensemble-mean kernel overlap the next step's scoring kernel if it ran on a
second stream?  Loop A: K1, K3 on one stream (the product).  Loop B: K1 on the compute stream, an event, a stand-in for K3 (a device copy
of one plane: a launch of about K3's size) on a second stream that waits for the event.  Loop C: K1 alone.
-> profiles/r6_side_stream_probe.log"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from tools.bench_common import build_members  # noqa: E402

eng = _native.Engine.get(0)
L, M, n, alpha = 8, 3, 100_000, "TGCA"
mods = build_members("cnn", L, alpha, M, 0)
natives = [m.native() for m in mods]
st = eng.torch_stream()
side = torch.cuda.Stream()
with torch.cuda.stream(st):
    d_seq = torch.from_numpy(synth.random_sequence_bytes(n, L, alpha, seed=0)).cuda()
    stride = (n + 63) // 64 * 64
    planes = [torch.zeros((M, stride), dtype=torch.float32, device="cuda") for _ in range(2)]
    mean = [torch.zeros((stride,), dtype=torch.float32, device="cuda") for _ in range(2)]
st.synchronize()
lut = mods[0]._lut


def loop(kind, steps):
    ev = [torch.cuda.Event() for _ in range(2)]
    t0 = time.perf_counter()
    for i in range(steps):
        s = i & 1
        with torch.cuda.stream(st):
            if kind == "B" and i >= 2:
                st.wait_event(ev[s])                              # the slot's previous stand-in has read the planes
            eng.score_planes_dev(natives, d_seq.data_ptr(), n, L, lut, planes[s].data_ptr(), stride)
            if kind == "A":
                eng.ensemble_mean_planes_dev(planes[s].data_ptr(), n, M, stride, mean[s].data_ptr())
        if kind == "B":
            side.wait_stream(st)
            with torch.cuda.stream(side):
                mean[s].copy_(planes[s][0])
                ev[s].record(side)
    st.synchronize(); side.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for k in ("A", "B", "C"):
    loop(k, 50)
for rep in range(3):
    print("  ".join(f"{k}: {loop(k, 400):7.2f} us/step" for k in ("A", "B", "C")), flush=True)
