"""Phase timeline of the training kernel's forward+backward launch (engine option train_trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get(0)
eng.set_option("train_trace", 1)
for tag, mk, L, alpha in (("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA"),
                          ("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA"),
                          ("CNN L=90 A=20", lambda: bm.CNN(90, 32, 100, s_utils.AAS, seed=0), 90, s_utils.AAS)):
    model = mk()
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, L, alpha, 3)); y = np.random.default_rng(0).random(1000)
    model.train(seqs, y); model.train(seqs, y)
    out = np.zeros(64, np.uint64)
    eng.check(eng._lib.fx_debug_train_trace(eng.handle, out.ctypes.data))
    t0 = int(out[0])
    if out[62]: print(f"   kernel entry -> weights staged: {(t0 - int(out[62])) / 100:.2f} us")
    names = {0: "start (after weight staging)", 1: "codes+labels", 2: "conv1", 3: "conv2", 4: "conv3", 5: "pool", 20: "dense0 fwd", 21: "dense1 fwd", 22: "dense2 fwd",
             23: "dense3 fwd", 7: "loss", 33: "dense3 bwd", 32: "dense2 bwd", 31: "dense1 bwd", 30: "dense0 bwd", 9: "pool bwd", 10: "conv3 bwd", 11: "conv2 bwd", 63: "end (conv1 wgrad)"}
    ev = sorted((int(out[k]), k) for k in names if out[k])
    print(tag)
    prev = t0
    for t, k in ev:
        print(f"   {names[k]:30s} +{(t - prev) / 100:.2f} us   (at {(t - t0) / 100:.2f})")
        prev = t
