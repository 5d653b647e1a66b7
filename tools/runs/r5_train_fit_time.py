"""Fit time of the GFP-length ensemble (3 x CNN(32, 100, 5), L = 237, n = 500) + wave 0's phase timeline, for whichever library
FLEXS_AMD_LIB names (round 5's kernel-form experiments are separate builds of train.hip: -DFXT_VAR=n)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
L = int(sys.argv[1]) if len(sys.argv) > 1 else 237
eng = _native.Engine.get()
model = flexs_amd.Ensemble([bm.CNN(L, 32, 100, s_utils.AAS, seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(500, L, s_utils.AAS, 3)); y = np.random.default_rng(0).random(500)
model.train(seqs, y, seed=5); torch.cuda.synchronize()
w = np.concatenate([np.concatenate([np.asarray(a, np.float32).ravel() for a in m.model.get_weights()]) for m in model.models])
ts = []
for _ in range(7):
    t0 = time.perf_counter(); model.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
out = np.zeros(64, np.uint64)
eng.set_option("train_trace", 1)
model.train(seqs, y, seed=5); torch.cuda.synchronize()
eng.check(eng._lib.fx_debug_train_trace(eng.handle, out.ctypes.data))
eng.set_option("train_trace", 0)
names = {1: "codes", 2: "conv1", 3: "conv2", 4: "conv3", 5: "pool", 20: "d0", 21: "d1", 22: "d2", 7: "loss", 32: "d2b", 31: "d1b", 30: "d0b", 9: "poolb", 10: "conv3b", 11: "conv2b", 63: "end"}
prev, parts = int(out[0]), []
for t, k in sorted((int(out[k]), k) for k in names if out[k]):
    parts.append(f"{names[k]} {(t - prev) / 100.0:.1f}"); prev = t
import zlib
print(f"{os.environ.get('FLEXS_AMD_LIB', 'default')}: fit {min(ts) * 1e3:.2f} ms (median {sorted(ts)[3] * 1e3:.2f}); weights crc {zlib.crc32(w.tobytes()):08x}; step {(prev - int(out[0])) / 100.0:.1f} us: " + ", ".join(parts), flush=True)
