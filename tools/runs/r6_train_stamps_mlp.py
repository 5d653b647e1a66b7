"""Raw phase stamps (engine option train_trace: 100 MHz wall clock, workgroup (0, 0), wave 0, last step) of a GFP-length fit, including
the sub-phase stamps 40-59 (train_core.h FXT_STAMP): microseconds since the step's start, in time order.
usage: r5_train_stamps.py [L] [members] [train_swizzle or -1] [alphabet] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
L = int(sys.argv[1]) if len(sys.argv) > 1 else 237
M = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = _native.Engine.get()
if len(sys.argv) > 3 and int(sys.argv[3]) >= 0:
    eng.set_option("train_swizzle", int(sys.argv[3]))
ALPHA = sys.argv[4] if len(sys.argv) > 4 else s_utils.AAS
NROWS = int(sys.argv[5]) if len(sys.argv) > 5 else 500
NAMES = {0: "start", 1: "codes+labels", 2: "conv1", 3: "conv2", 4: "conv3", 5: "pool", 20: "dense0 fwd", 21: "dense1 fwd", 22: "dense2 fwd", 7: "loss",
         32: "dense2 bwd", 31: "dense1 bwd", 30: "dense0 bwd", 9: "pool bwd", 10: "conv3 bwd", 11: "conv2 bwd", 63: "end", 62: "entry",
         40: "conv3 in-grad done (w0)", 41: "conv2 taps committed", 42: "conv2 in-grad done (w0)", 43: "conv2 wgrad done (w0)"}
model = bm.MLP(L, int(os.environ.get("FX_H", 200)), ALPHA, seed=0)
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(NROWS, L, ALPHA, 3)); y = np.random.default_rng(0).random(NROWS)
model.train(seqs, y, seed=5)
out = np.zeros(64, np.uint64)
eng.set_option("train_trace", 1)
try:
    model.train(seqs, y, seed=5); torch.cuda.synchronize()
    eng.check(eng._lib.fx_debug_train_trace(eng.handle, out.ctypes.data))
finally:
    eng.set_option("train_trace", 0)
t0 = int(out[0])
prev = t0
for t, k in sorted((int(out[k]), k) for k in range(64) if out[k]):
    print(f"{(t - t0) / 100.0:9.2f} us  (+{(t - prev) / 100.0:7.2f})  stamp {k:2d}  {NAMES.get(k, '')}")
    prev = t
