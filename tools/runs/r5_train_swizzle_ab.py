"""PREPARED at the end of round 4, to be run FIRST in round 5 (no GPU seconds were left to run it): rotated rows for the long protein
CNNs' training step (engine option train_swizzle = 1, csrc/train_core.h "Rotated rows"), and = 2: on top of it the gradient array over the last conv output
and the conv kernels of conv2 / conv3 staged through the LDS that frees, six taps at a time (fxt_gemm_staged; on the CPU its device
branch runs under the SIMT emulator of tests/native/simt_train.cpp, bit-identical and race-free there -- never yet on a device).  Checks that a fit gives the SAME BITS with the
option on and off, then times `train` both ways.  Expected where it applies (padded workspace past the 150 KiB LDS budget but unpadded within it:
CNN(32 filters, kernel 5) at one row per slice, L = 226 ... 239 -- GFP's 237 / 238 residues): the conv phases lose their 16-way LDS bank conflicts.  Shapes whose padded workspace fits are not touched by the
option (same time expected: a control).  If it wins: flip the default in fx_common.h, add the bit-identity leg below to
tests/test_train_native.py, record the numbers in csrc/OPTIONS.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
bad = 0
AS_JSON = "--json" in sys.argv          # bench.py's child: two shapes, one JSON line {shape: {ms: {...}, same_bits: {...}}}
report = {}

def run(tag, make, L, alpha, n):
    global bad
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 3))
    y = np.random.default_rng(0).random(n)
    weights, times = [], []
    for swz in (0, 1, 2):
        eng.set_option("train_swizzle", swz)
        model = make()
        model.train(seqs, y, seed=5); torch.cuda.synchronize()      # (seeded: the same shuffles and dropout masks in every leg)
        members = model.models if hasattr(model, "models") else [model]
        weights.append([np.concatenate([np.asarray(w, np.float32).ravel() for w in m.model.get_weights()]) for m in members])
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); model.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        times.append(min(ts) * 1e3)
    eng.set_option("train_swizzle", 0)
    same = [all(np.array_equal(a, b) for a, b in zip(weights[0], weights[k])) for k in (1, 2)]
    finite = all(np.isfinite(a).all() for k in (1, 2) for a in weights[k])
    bad += (not all(same)) or (not finite)
    report[f"{tag} n={n}"] = {"ms_per_fit": {"plain": round(times[0], 3), "rotated_rows": round(times[1], 3), "staged_conv_kernels": round(times[2], 3)},
                              "same_bits_as_plain": {"rotated_rows": bool(same[0]), "staged_conv_kernels": bool(same[1])}, "finite": bool(finite)}
    if AS_JSON:
        return
    print(f"{tag} n={n}: unrotated {times[0]:.2f} ms, rotated rows {times[1]:.2f} ms, + staged conv kernels {times[2]:.2f} ms; weights after "
          f"the first fit: rotated {'IDENTICAL' if same[0] else 'DIFFER'}, staged {'IDENTICAL' if same[1] else 'DIFFER'}"
          f"{'' if finite else ' (not finite)'}", flush=True)

run("Ensemble 3xCNN L=237 A=20", lambda: flexs_amd.Ensemble([bm.CNN(237, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 237, s_utils.AAS, 500)
if AS_JSON:
    import json
    run("CNN L=238 A=20 (GFP + 1)", lambda: bm.CNN(238, 32, 100, s_utils.AAS, seed=0), 238, s_utils.AAS, 300)
    print(json.dumps(report), flush=True)
    sys.exit(0)
run("CNN L=237 A=20", lambda: bm.CNN(237, 32, 100, s_utils.AAS, seed=0), 237, s_utils.AAS, 500)
run("CNN L=238 A=20 (GFP + 1)", lambda: bm.CNN(238, 32, 100, s_utils.AAS, seed=0), 238, s_utils.AAS, 300)
run("CNN L=230 A=20", lambda: bm.CNN(230, 32, 100, s_utils.AAS, seed=0), 230, s_utils.AAS, 500)
run("control: CNN L=200 A=20 (padded rows fit)", lambda: bm.CNN(200, 32, 100, s_utils.AAS, seed=0), 200, s_utils.AAS, 500)
run("control: Ensemble 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1000)
sys.exit(1 if bad else 0)
