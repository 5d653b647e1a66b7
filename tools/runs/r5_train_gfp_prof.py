"""rocprofv3 driver (round 5): `Ensemble.train` of 3 x CNN(32, 100, kernel 5) on 500 GFP-length sequences (L = 237, 20 letters) --
the fit BASELINE configs[4]'s explorer runs every round (flexs/explorer.py:157-160 -> keras_model.py:49-67).  No resident
workgroups, no pre-launched instance (a kernel that polls never ends under --kernel-trace).
usage: r5_train_gfp_prof.py [train_swizzle] [fits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
eng.set_option("serve_small", 0)
eng.set_option("lp_prelaunch", 0)
if len(sys.argv) > 1:
    eng.set_option("train_swizzle", int(sys.argv[1]))
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = 237
ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, s_utils.AAS, seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(500, L, s_utils.AAS, 3)); y = np.random.default_rng(0).random(500)
for _ in range(fits):
    ens.train(seqs, y)
print("train_swizzle", eng.get_option("train_swizzle"), "fits", fits)
