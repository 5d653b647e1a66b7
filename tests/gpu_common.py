"""Shared by the GPU parity test files (tests/test_gpu_*.py): the tolerance north_star states, the score assertion that also
records every family's worst error (gpurun_out/parity_error_stats.json -> profiles/), the engine fixture, seeded inputs."""

import json
import os

import numpy as np
import pytest

import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
from oracle import c_oracle, ref_np


# Round 5 (verdict item 7): the absolute term is what the kernels were MEASURED to need -- the worst absolute error over every
# family of this suite is 2.5e-7 (profiles/r1_run69_parity_error_stats.json, profiles/r5_parity_error_stats.json) -- not the 1e-6
# of rounds 1-4 (6 x slack); the relative term is north_star's.
RTOL, ATOL = 1e-5, 2.5e-7
ERROR_STATS = {}          # what -> worst figures seen by assert_scores in this session (written out when a module's `eng` fixture ends: cumulative over the files of the session)




def close(got, want):
    return np.abs(got - want) <= ATOL + RTOL * np.abs(want)


def assert_scores(got, want, what=""):
    assert got.dtype == np.float32
    assert not np.isnan(got).any(), f"{what}: {np.isnan(got).sum()} output elements were never written"
    g64, w64 = got.astype(np.float64), np.asarray(want, np.float64)
    if g64.size:
        err = np.abs(g64 - w64)
        row = ERROR_STATS.setdefault(what or "(unnamed)", {"n": 0, "max_abs_err": 0.0, "max_err_over_tolerance": 0.0, "max_abs_ref": 0.0})
        row["n"] += int(err.size)
        row["max_abs_err"] = max(row["max_abs_err"], float(err.max()))
        row["max_err_over_tolerance"] = max(row["max_err_over_tolerance"], float((err / (ATOL + RTOL * np.abs(w64))).max()))
        row["max_abs_ref"] = max(row["max_abs_ref"], float(np.abs(w64).max()))
    bad = ~close(got.astype(np.float64), want)
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} outside tolerance; max abs err "
                           f"{np.abs(got - want).max():.3e}, worst at {np.argmax(np.abs(got - want))}")


@pytest.fixture(scope="module")
def eng():
    e = _native.Engine.get(0)
    e.set_option("poison_outputs", 1)       # an output element that no kernel wrote shows up as NaN
    yield e
    try:                                    # per-config worst errors of this run (gpurun merges gpurun_out/ back; copied to profiles/ by hand)
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        rows = [dict(what=k, atol=ATOL, rtol=RTOL, **v) for k, v in sorted(ERROR_STATS.items())]
        with open(os.path.join(out_dir, "parity_error_stats.json"), "w") as fh:
            json.dump(rows, fh, indent=1)
    except OSError:
        pass
    e.set_option("poison_outputs", 0)
    e.set_option("force_generic", 0)
    e.set_option("cnn_variant", 0)
    e.set_option("cnn_conv1_mfma", 0)
    e.set_option("mlp_l1_mfma", 0)
    e.set_option("cnn_pair", 1)


def ab_option(eng, key, value):
    """Selects a kernel form that was measured and lost (csrc/OPTIONS.md): compiled into the A/B build only
    (`make -C flexs_amd/csrc ab`, FLEXS_AMD_LIB=.../libflexs_amd_ab.so).  False = the production library refused it: the
    caller skips that leg."""
    try:
        eng.set_option(key, value)
        return True
    except _native.FxError as ex:
        if ex.code == _native.FX_EUNSUPPORTED:
            return False
        raise


def rand_seqs(n, L, alphabet, seed):
    b = synth.random_sequence_bytes(n, L, alphabet, seed)
    return b, synth.bytes_to_strings(b)


def make_native(eng, kind, L, A, H, F=0, K=0, seed=1000):
    shapes = {"cnn": ref_np.cnn_shapes(L, A, F, H, K) if kind == "cnn" else None,
              "mlp": ref_np.mlp_shapes(L, A, H), "ge": ref_np.ge_shapes(L, A, H)}[kind]
    w = ref_np.synth_weights(shapes, seed)
    nm = _native.NativeModel(eng, {"cnn": 0, "mlp": 1, "ge": 2}[kind], L, A, F, H, K)
    nm.set_weights(w)
    return nm, w
