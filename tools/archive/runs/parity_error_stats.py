"""Observed error of the device scores against the float64 oracle (tests hold them to |err| <= 1e-5 |ref| + 1e-6)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flexs_amd import _native, synth
from flexs_amd.utils.sequence_utils import AAS
from oracle import c_oracle, ref_np

eng = _native.Engine.get()
rows = []
for kind, L, alpha, H, F, K, n in (("cnn", 8, "TGCA", 100, 32, 5, 20000), ("cnn", 14, "UGCA", 100, 32, 5, 20000),
                                   ("cnn", 100, "UGCA", 100, 32, 5, 3000), ("cnn", 237, AAS, 100, 32, 5, 400),
                                   ("mlp", 14, "UGCA", 100, 0, 0, 20000), ("mlp", 90, AAS, 200, 0, 0, 5000),
                                   ("ge", 90, AAS, 100, 0, 0, 20000)):
    A = len(alpha)
    shapes = {"cnn": lambda: ref_np.cnn_shapes(L, A, F, H, K), "mlp": lambda: ref_np.mlp_shapes(L, A, H),
              "ge": lambda: ref_np.ge_shapes(L, A, H)}[kind]()
    w = ref_np.synth_weights(shapes, 4242)
    nm = _native.NativeModel(eng, {"cnn": 0, "mlp": 1, "ge": 2}[kind], L, A, F, H, K)
    nm.set_weights(w)
    b = synth.random_sequence_bytes(n, L, alpha, 9)
    lut = _native.make_lut(alpha)
    got, _ = eng.score([nm], b, lut)
    ref = c_oracle.forward(kind, lut[b], A, w)
    err = np.abs(got[:, 0].astype(np.float64) - ref)
    rel = err / np.maximum(np.abs(ref), 1e-30)
    rows.append({"model": f"{kind} L={L} A={A} H={H}", "n": n, "max_abs_err": float(err.max()), "max_rel_err": float(rel.max()),
                 "median_rel_err": float(np.median(rel)), "max_err_over_tolerance": float((err / (1e-5 * np.abs(ref) + 1e-6)).max())})
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/parity_error_stats.json", "w"), indent=1)
