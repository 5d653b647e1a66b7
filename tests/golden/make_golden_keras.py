#!/usr/bin/env python3
"""Generate tests/golden/keras_forward.npz -- the fixture that would PIN the Keras forward pass.

    python tests/golden/make_golden_keras.py          # needs /root/reference AND tensorflow

TensorFlow is not installable in the build container or on the GPU box, so this script has never been run there
and the fixture does not exist: oracle/ and DESIGN.md say "Keras forward: parity unpinned".  On a machine that has
TensorFlow next to a checkout of the reference it runs the reference's own `CNN` / `MLP` / `GlobalEpistasisModel`
(`flexs/baselines/models/{cnn,mlp,global_epistasis_model}.py`) with Keras' random initial weights (biases replaced
by non-zero values so that every bias path counts) on random sequences and stores inputs, weights and
`get_fitness` outputs.  `tests/test_oracle.py::test_keras_forward_fixture` picks the file up when present and holds
the three oracles to it at 1e-5 relative.
"""
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference  # noqa: E402


def main():
    try:
        import tensorflow  # noqa: F401
    except ImportError:
        print("tensorflow is not installed here: keras_forward.npz cannot be generated (parity stays unpinned)")
        return 1
    _, s_utils, *_ = import_reference()
    importlib.import_module("flexs.baselines.models.keras_model")
    kinds = {"cnn": importlib.import_module("flexs.baselines.models.cnn").CNN,
             "mlp": importlib.import_module("flexs.baselines.models.mlp").MLP,
             "ge": importlib.import_module("flexs.baselines.models.global_epistasis_model").GlobalEpistasisModel}
    rng = np.random.default_rng(7)
    cases, arrays = [], {}
    # what the restatement had to guess from documentation is what the case list covers: 'same' padding split for even
    # kernels (left (k-1)//2, rest right), conv3 kernel = len(alphabet) - 1 for both alphabets (3 and 19 taps), the
    # smallest legal sequence (L == kernel_size), hidden sizes that are not multiples of 16, Flatten order (MLP / GE),
    # float32 outputs; plus the reference's own smoke shapes (tests/test_models.py:55-77)
    for i, (kind, L, alpha, kw) in enumerate((
            ("cnn", 8, s_utils.DNAA, dict(num_filters=32, hidden_size=100)),
            ("cnn", 14, s_utils.RNAA, dict(num_filters=32, hidden_size=100)),
            ("cnn", 30, s_utils.AAS, dict(num_filters=32, hidden_size=100)),                   # conv3 with 19 taps
            ("cnn", 9, s_utils.DNAA, dict(num_filters=8, hidden_size=20, kernel_size=4)),      # even kernel: padding split
            ("cnn", 12, s_utils.AAS, dict(num_filters=16, hidden_size=50, kernel_size=2)),     # even kernel + even-free conv3 (19)
            ("cnn", 11, s_utils.DNAA, dict(num_filters=24, hidden_size=37, kernel_size=6)),    # even kernel, odd sizes
            ("cnn", 5, s_utils.DNAA, dict(num_filters=32, hidden_size=100)),                   # L == kernel_size: one conv1 position
            ("cnn", 3, s_utils.DNAA, dict(num_filters=1, hidden_size=1, kernel_size=2)),       # the reference's smoke shape
            ("cnn", 10, s_utils.BA, dict(num_filters=32, hidden_size=100)),                    # binary alphabet: conv3 has one tap
            ("mlp", 14, s_utils.RNAA, dict(hidden_size=100)),
            ("mlp", 7, s_utils.AAS, dict(hidden_size=33)),
            ("mlp", 3, s_utils.DNAA, dict(hidden_size=1)),
            ("ge", 20, s_utils.AAS, dict(hidden_size=100)),
            ("ge", 90, s_utils.AAS, dict(hidden_size=100)),
            ("ge", 3, s_utils.DNAA, dict(hidden_size=1)))):
        model = kinds[kind](L, alphabet=alpha, **kw)
        weights = model.model.get_weights()
        weights = [w if w.ndim > 1 else rng.uniform(-0.1, 0.1, w.shape).astype(np.float32) for w in weights]
        model.model.set_weights(weights)
        seqs = ["".join(alpha[j] for j in rng.integers(0, len(alpha), L)) for _ in range(64)]
        out = np.asarray(model.get_fitness(seqs))
        cases.append({"kind": kind, "L": L, "alphabet": alpha, "kwargs": kw, "n_weights": len(weights), "sequences": seqs,
                      "dtype": str(out.dtype)})
        for j, w in enumerate(weights):
            arrays[f"c{i}_w{j}"] = np.asarray(w, np.float32)
        arrays[f"c{i}_out"] = out
    # the construction-time error for seq_len < kernel_size ('valid' Conv1D, cnn.py:25-32): record its type
    try:
        kinds["cnn"](3, alphabet=s_utils.DNAA, num_filters=4, hidden_size=4)
        err = None
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    cases.append({"kind": "cnn_too_short", "L": 3, "exception": err})
    arrays["meta"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "keras_forward.npz"), **arrays)
    print("wrote keras_forward.npz with", len(cases) - 1, "forward cases + the L < kernel_size exception type")
    return 0


if __name__ == "__main__":
    sys.exit(main())
