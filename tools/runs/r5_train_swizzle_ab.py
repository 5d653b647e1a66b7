"""PREPARED at the end of round 4, to be run FIRST in round 5 (no GPU seconds were left to run it): rotated rows for the long protein
CNNs' training step (engine option train_swizzle = 1, csrc/train_core.h "Rotated rows"), and = 2: on top of it the gradient array over the last conv output
and the conv kernels of conv2 / conv3 staged through the LDS that frees, six taps at a time (fxt_gemm_staged; on the CPU its device
branch runs under the SIMT emulator of tests/native/simt_train.cpp, bit-identical and race-free there -- never yet on a device).  Checks that a fit gives the SAME BITS with the
option on and off, then times `train` both ways.  Expected where it applies (padded workspace past the 150 KiB LDS budget but unpadded within it:
CNN(32 filters, kernel 5) at one row per slice, L = 226 ... 239 -- GFP's 237 / 238 residues): the conv phases lose their 16-way LDS bank conflicts.  Shapes whose padded workspace fits are not touched by the
option (same time expected: a control).  If it wins: flip the default in fx_common.h, add the bit-identity leg below to
tests/test_train_native.py, record the numbers in csrc/OPTIONS.md."""
import json, os, sys, time
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

PHASES = {1: "codes+labels", 2: "conv1", 3: "conv2", 4: "conv3", 5: "pool", 20: "dense0 fwd", 21: "dense1 fwd", 22: "dense2 fwd", 7: "loss",
          32: "dense2 bwd", 31: "dense1 bwd", 30: "dense0 bwd", 9: "pool bwd", 10: "conv3 bwd", 11: "conv2 bwd", 63: "conv1 wgrad (end)"}

def phase_us(model, seqs, y):
    """One workgroup's timeline of the LAST mini-batch step of a fit (engine option train_trace: the 100 MHz wall clock after each
    phase's barrier): microseconds per phase."""
    out = np.zeros(64, np.uint64)
    eng.set_option("train_trace", 1)
    try:
        model.train(seqs, y, seed=5); torch.cuda.synchronize()
        eng.check(eng._lib.fx_debug_train_trace(eng.handle, out.ctypes.data))
    finally:
        eng.set_option("train_trace", 0)
    prev, res = int(out[0]), {}
    for t, k in sorted((int(out[k]), k) for k in PHASES if out[k]):
        res[PHASES[k]] = round((t - prev) / 100.0, 2)
        prev = t
    res["whole step"] = round((prev - int(out[0])) / 100.0, 2)
    return res

LEGS = ("plain", "rotated_rows", "staged_conv_kernels", "f32_form")

def run(tag, make, L, alpha, n, phases=False):
    """The three legs one after the other; a leg that fails is named in the report and the others still count.  In --json mode the
    report so far is printed after every leg (the parent keeps the last line it got, also when it had to stop this process)."""
    global bad
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 3))
    y = np.random.default_rng(0).random(n)
    row = report[f"{tag} n={n}"] = {"ms_per_fit": {}, "same_bits_as_plain": {}}
    weights = {}
    for swz, leg in enumerate(LEGS):
        try:
            eng.set_option("train_swizzle", swz)
            model = make()
            model.train(seqs, y, seed=5); torch.cuda.synchronize()      # (seeded: the same shuffles and dropout masks in every leg)
            members = model.models if hasattr(model, "models") else [model]
            weights[leg] = [np.concatenate([np.asarray(w, np.float32).ravel() for w in m.model.get_weights()]) for m in members]
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); model.train(seqs, y); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row["ms_per_fit"][leg] = round(min(ts) * 1e3, 3)
            if swz:
                ok = "plain" in weights and all(np.array_equal(a, b) for a, b in zip(weights["plain"], weights[leg]))
                fin = all(np.isfinite(a).all() for a in weights[leg])
                row["same_bits_as_plain"][leg] = bool(ok and fin)
                bad += not (ok and fin)
            if phases:
                try:
                    row.setdefault("phase_us_of_one_workgroup", {})[leg] = phase_us(model, seqs, y)
                except Exception as ex:                      # (the timeline is an extra: never at the cost of the times)
                    row.setdefault("phase_us_of_one_workgroup", {})[leg] = {"error": f"{type(ex).__name__}: {ex}"}
        except Exception as ex:  # noqa: BLE001
            row.setdefault("errors", {})[leg] = f"{type(ex).__name__}: {ex}"[:300]
            bad += 1
        finally:
            try:
                eng.set_option("train_swizzle", 0)
            except Exception:  # noqa: BLE001
                pass
        if AS_JSON:
            print(json.dumps(report), flush=True)
    if AS_JSON:
        return
    for leg, tl in row.get("phase_us_of_one_workgroup", {}).items():
        print(f"   {leg}: " + ", ".join(f"{k} {v}" for k, v in tl.items()), flush=True)
    print(f"{tag} n={n}: ms per fit {row['ms_per_fit']}; same bits as the plain step: {row['same_bits_as_plain']}"
          f"{'; errors ' + str(row['errors']) if 'errors' in row else ''}", flush=True)

run("Ensemble 3xCNN L=237 A=20", lambda: flexs_amd.Ensemble([bm.CNN(237, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 237, s_utils.AAS, 500, phases=True)
if AS_JSON:
    run("CNN L=238 A=20 (GFP + 1)", lambda: bm.CNN(238, 32, 100, s_utils.AAS, seed=0), 238, s_utils.AAS, 300)
    print(json.dumps(report), flush=True)
    sys.exit(0)
run("CNN L=237 A=20", lambda: bm.CNN(237, 32, 100, s_utils.AAS, seed=0), 237, s_utils.AAS, 500)
run("CNN L=238 A=20 (GFP + 1)", lambda: bm.CNN(238, 32, 100, s_utils.AAS, seed=0), 238, s_utils.AAS, 300)
run("CNN L=230 A=20", lambda: bm.CNN(230, 32, 100, s_utils.AAS, seed=0), 230, s_utils.AAS, 500)
run("Ensemble 3xCNN L=90 A=20 (AAV length)", lambda: flexs_amd.Ensemble([bm.CNN(90, 32, 100, s_utils.AAS, seed=m) for m in range(3)]), 90, s_utils.AAS, 500, phases=True)
run("CNN L=250 A=20 (past the five-array layout)", lambda: bm.CNN(250, 32, 100, s_utils.AAS, seed=0), 250, s_utils.AAS, 300)
run("control: CNN L=200 A=20 (padded rows fit)", lambda: bm.CNN(200, 32, 100, s_utils.AAS, seed=0), 200, s_utils.AAS, 500)
run("control: Ensemble 3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA", 1000)
sys.exit(1 if bad else 0)
