"""ctypes wrapper around oracle/libfx_oracle.so (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfx_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        u8p, f32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
        _lib.fxo_levenshtein.argtypes = [u8p, C.c_int, u8p, C.c_int]
        _lib.fxo_levenshtein.restype = C.c_int
        _lib.fxo_hamming.argtypes = [u8p, u8p, C.c_int]
        _lib.fxo_hamming.restype = C.c_int
        _lib.fxo_min_dist.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.c_int, C.c_int,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        _lib.fxo_min_dist.restype = None
        _lib.fxo_cnn_forward.argtypes = [u8p, C.c_int64] + [C.c_int] * 5 + [f32p, f64p]
        _lib.fxo_cnn_forward.restype = C.c_int
        _lib.fxo_mlp_forward.argtypes = [u8p, C.c_int64] + [C.c_int] * 3 + [f32p, f64p]
        _lib.fxo_mlp_forward.restype = C.c_int
        _lib.fxo_ge_forward.argtypes = [u8p, C.c_int64] + [C.c_int] * 3 + [f32p, f64p]
        _lib.fxo_ge_forward.restype = C.c_int
        _lib.fxo_ensemble_mean_f32.argtypes = [f32p, C.c_int64, C.c_int, f32p]
        _lib.fxo_ensemble_mean_f32.restype = None
        _lib.fxo_argmax_decode.argtypes = [f64p, C.c_int64, C.c_int, C.c_int, u8p]
        _lib.fxo_argmax_decode.restype = None
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _bytes(s):
    return np.frombuffer(s.encode("latin-1") if isinstance(s, str) else bytes(s), dtype=np.uint8)


def levenshtein(a, b) -> int:
    a, b = _bytes(a), _bytes(b)
    return lib().fxo_levenshtein(_p(a, C.c_uint8), len(a), _p(b, C.c_uint8), len(b))


def hamming(a, b) -> int:
    a, b = _bytes(a), _bytes(b)
    assert len(a) == len(b)
    return lib().fxo_hamming(_p(a, C.c_uint8), _p(b, C.c_uint8), len(a))


def min_dist(q_u8: np.ndarray, cache_u8: np.ndarray, mode: int = 0):
    """q (Q, L) uint8, cache (C, L) uint8 -> (dist int32 (Q,), argmin int64 (Q,))."""
    q = np.ascontiguousarray(q_u8, np.uint8)
    c = np.ascontiguousarray(cache_u8, np.uint8)
    Q, L = q.shape
    dist = np.empty(Q, np.int32)
    arg = np.empty(Q, np.int64)
    lib().fxo_min_dist(_p(q, C.c_uint8), Q, _p(c, C.c_uint8), c.shape[0], L, mode,
                       _p(dist, C.c_int32), _p(arg, C.c_int64))
    return dist, arg


def blob_of(weights) -> np.ndarray:
    return np.concatenate([np.asarray(w, np.float32).ravel() for w in weights])


def forward(kind: str, codes: np.ndarray, A: int, weights, F=None, H=None, K=None) -> np.ndarray:
    """Scalar-loop float64 forward for codes (N, L) uint8."""
    codes = np.ascontiguousarray(codes, np.uint8)
    N, L = codes.shape
    blob = blob_of(weights)
    out = np.empty(N, np.float64)
    if kind == "cnn":
        K = weights[0].shape[0]
        F = weights[0].shape[2]
        H = weights[6].shape[1]
        rc = lib().fxo_cnn_forward(_p(codes, C.c_uint8), N, L, A, F, H, K, _p(blob, C.c_float), _p(out, C.c_double))
    elif kind == "mlp":
        H = weights[0].shape[1]
        rc = lib().fxo_mlp_forward(_p(codes, C.c_uint8), N, L, A, H, _p(blob, C.c_float), _p(out, C.c_double))
    elif kind == "ge":
        H = weights[2].shape[1]
        rc = lib().fxo_ge_forward(_p(codes, C.c_uint8), N, L, A, H, _p(blob, C.c_float), _p(out, C.c_double))
    else:
        raise ValueError(kind)
    if rc != 0:
        raise ValueError("valid conv with L < kernel_size")
    return out


def ensemble_mean_f32(scores_NM: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(scores_NM, np.float32)
    out = np.empty(s.shape[0], np.float32)
    lib().fxo_ensemble_mean_f32(_p(s, C.c_float), s.shape[0], s.shape[1], _p(out, C.c_float))
    return out


def argmax_decode(one_hot: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(one_hot, np.float64)
    P, L, A = x.shape
    idx = np.empty((P, L), np.uint8)
    lib().fxo_argmax_decode(_p(x, C.c_double), P, L, A, _p(idx, C.c_uint8))
    return idx
