"""ctypes binding of libflexs_amd.so (C ABI in include/flexs_amd.h).

There is deliberately NO CPU fallback: if the shared library is missing, or no
gfx950 device is visible when an engine is needed, this module raises.  PyTorch
is imported first only so that its bundled HIP runtime (same soname,
libamdhip64.so.7) is the single runtime in the process -- torch tensors and
streams can then be handed to the `_dev` entry points by raw pointer.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLEXS_AMD_LIB: another build of the same library (profiling builds, e.g. `make trace` = -DFX_TRACE_PHASES)
LIB_PATH = os.environ.get("FLEXS_AMD_LIB") or os.path.join(_HERE, "libflexs_amd.so")

FX_OK, FX_EINVAL, FX_ESHAPE, FX_EBADCHAR, FX_ENODEV = 0, -1, -2, -3, -4
FX_EHIP, FX_ENOMEM, FX_EUNSUPPORTED, FX_ESTATE = -5, -6, -7, -8
FX_CNN, FX_MLP, FX_GE = 0, 1, 2
FX_LEVENSHTEIN, FX_HAMMING = 0, 1

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); the CPU test-suite checks that every one is exported
from flexs_amd._abi import SIGNATURES  # noqa: E402  (the ctypes signature table, one entry per function of include/flexs_amd.h)

_lib = None


def lib():
    """Load libflexs_amd.so (fails loudly if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C flexs_amd/csrc`). "
                "flexs_amd has no CPU fallback."
            )
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first -> one HIP runtime per process)

        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def status_name(code: int) -> str:
    return lib().fx_status_name(code).decode()


class FxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{status_name(code)}: {msg}")
        self.code = code


def _raise(code: int, eng_handle=None):
    msg = lib().fx_last_error(eng_handle).decode()
    # exception types follow the reference: unknown character -> ValueError (str.index, sequence_utils.py:46); shape problems ->
    # ValueError (Keras raises ValueError on incompatible input shapes)
    if code in (FX_EBADCHAR, FX_ESHAPE):
        raise ValueError(msg or status_name(code))
    raise FxError(code, msg)


def raise_small_status(st: int, eng_handle=None):
    """Status words of the C fast paths in csrc/strpack.c (score_small, population_step): 0 done, -1 / 1 / 1001-1003 "not for this
    path" (handled by the callers), 2000 + |FX_E*| a library error.  One place turns the last form back into the (negative) FX code
    `_raise` maps to the reference's exception types; anything else is reported as it is instead of being passed on as if it were a
    code (round-4 advisor finding)."""
    if st > 2000:
        _raise(-(st - 2000), eng_handle)
    if st < 0:
        _raise(st, eng_handle)
    raise FxError(FX_EINVAL, f"unexpected status {st} from the C fast path")


def _ptr(a: Optional[np.ndarray]):
    """Address of an array's buffer as a plain int (c_void_p parameters take ints; `ndarray.ctypes.data_as` costs
    microseconds, which is a tenth of a small call)."""
    return None if a is None else a.__array_interface__["data"][0]


_LUT_BUFFERS: Dict[bytes, "C.Array"] = {}


def _lut_ptr(lut: np.ndarray):
    """Persistent ctypes copy of a 256-byte LUT (keyed by content), so the per-call pointer is a dict look-up."""
    key = lut.tobytes()
    buf = _LUT_BUFFERS.get(key)
    if buf is None:
        buf = _LUT_BUFFERS[key] = C.cast((C.c_uint8 * 256).from_buffer_copy(key), _u8p)
    return buf


def make_lut(alphabet: str) -> np.ndarray:
    """byte -> alphabet index (first occurrence, like str.index); 0xFF = absent."""
    lut = np.full(256, 0xFF, np.uint8)
    for i in range(len(alphabet) - 1, -1, -1):
        o = ord(alphabet[i])
        if o > 255:
            raise ValueError("alphabet characters must be single bytes")
        lut[o] = i
    return lut


try:                                   # CPython helper built next to libflexs_amd.so by csrc/Makefile
    from flexs_amd import _strpack
except ImportError:                    # host-side convenience only: the pure-Python path below does the same
    _strpack = None
if _strpack is not None and os.environ.get("FLEXS_AMD_PACK_THREADS"):
    _strpack.set_threads(int(os.environ["FLEXS_AMD_PACK_THREADS"]))     # 0 = auto (min(8, cores / 2)), 1 = single-threaded


def ragged_to_bytes(sequences, L: int) -> np.ndarray:
    """Strings of any lengths <= L -> (N, L) uint8 rows, NUL-padded on the right: the row format of the
    edit-distance entry points (fx_min_dist / fx_cache_*), which `editdistance.eval` semantics require to
    take unequal lengths (noisy_abstract_model.py:51)."""
    # (one join + one encode for the whole batch: per string an encode, a frombuffer, a NUL scan and a slice assignment were
    #  ~2 us each -- 20 of the 67 us of a ten-query sequence_density call; the checks and their order are the same)
    parts = []
    for s in sequences:
        s = str(s)
        k = len(s)
        if k > L:
            raise ValueError(f"sequence of length {k} does not fit a row of {L} bytes")
        if not s.isascii():
            try:
                s.encode("latin-1")
            except UnicodeEncodeError:
                raise ValueError("substring not found") from None
        if "\0" in s:
            raise ValueError("NUL characters cannot be part of a sequence")
        parts.append(s if k == L else s + "\0" * (L - k))
    if not parts:
        return np.zeros((0, L), np.uint8)
    return np.frombuffer("".join(parts).encode("latin-1"), np.uint8).reshape(len(parts), L).copy()


def sequences_to_bytes(sequences, L: Optional[int] = None, staging: Optional["Engine"] = None) -> np.ndarray:
    """list/tuple/ndarray of str -> contiguous (N, L) uint8 (latin-1 code points).

    Raises ValueError for ragged batches (Keras raises on a shape mismatch) and
    for characters that do not fit one byte (cannot be in any FLEXS alphabet)."""
    if isinstance(sequences, np.ndarray) and sequences.dtype.kind in "SU":
        # fixed-width NumPy strings: the item size may be wider than the strings (dtype 'U10' holding 8-mers, a slice of
        # a wider array): the common length is what counts, as for the reference's per-string loop
        a = np.ascontiguousarray(sequences)
        N = a.shape[0]
        if a.dtype.kind == "S":
            cp = a.view(np.uint8).reshape(N, a.dtype.itemsize)
        else:
            cp = a.view(np.uint32).reshape(N, a.dtype.itemsize // 4)
        if N:
            if cp.shape[1] and cp.min() != 0:                              # the usual case: the item size IS the length
                pass
            else:
                used = cp != 0
                w = int(used.any(axis=0).nonzero()[0].max(initial=-1)) + 1  # longest string
                cp = cp[:, :w]
                if not used[:, :w].all():
                    raise ValueError("ragged sequence batch")
            if cp.dtype != np.uint8 and cp.max(initial=0) > 255:
                raise ValueError("substring not found")
        out = np.ascontiguousarray(cp.astype(np.uint8, copy=False))
    else:
        seqs = sequences if isinstance(sequences, (list, tuple)) else list(sequences)
        N = len(seqs)
        if N == 0:
            return np.zeros((0, L or 0), np.uint8)
        w = len(seqs[0])
        if _strpack is not None:
            # with `staging`, straight into the engine's pinned input area (big batches: saves one pass over the bytes)
            out = staging.staging_rows(N, w) if (staging is not None and N * w >= (256 << 10)) else np.empty((N, w), np.uint8)
            status = _strpack.pack(seqs, w, out)          # one pass, memcpy per string (csrc/strpack.c)
        else:                                             # same checks in pure Python (helper not built)
            joined = "".join(seqs)
            status = 1 if (len(joined) != N * w or set(map(len, seqs)) != {w}) else 0
            if status == 0:
                try:
                    out = np.frombuffer(joined.encode("latin-1"), dtype=np.uint8).reshape(N, w)
                except UnicodeEncodeError:
                    status = 2
        if status == 1:
            raise ValueError("ragged sequence batch: all sequences must have the same length")
        if status == 2:
            raise ValueError("substring not found")
        if status == 3:
            raise TypeError("sequences must be str")
    if L is not None and out.shape[0] and out.shape[1] != L:
        raise ValueError(f"sequence length {out.shape[1]} does not match the model's seq_len {L}")
    return out


CHUNKED_MIN_ROWS = int(os.environ.get("FLEXS_AMD_CHUNKED_MIN_ROWS", 16384))   # list[str] batches from this size on are launched first and packed behind where the engine can (score_strings), else packed and scored in overlapping pieces where that pays (16384: where ensembles start to gain, profiles/r5_launch_first_mid.log)
CHUNK_BYTES = int(os.environ.get("FLEXS_AMD_CHUNK_BYTES", 0))   # target bytes per piece (0 = auto, see score_strings)


# ---- explorer-size calls: one C call packs the strings and runs fx_score (csrc/strpack.c score_small) -------------------
SMALL_CALL_ROWS = 4096           # what the engine's resident form answers (FX_SERVE_CAP; 256 until round 4)
SMALL_CALL_BYTES = 65536         # (FX_SERVE_BYTES, and the stack buffer of csrc/strpack.c score_small)
# calls of at least STREAM_MIN_ROWS strings are STREAMED when a resident generation can take them: the request is posted first and
# the strings are packed straight into its mailbox, STREAM_STEP_ROWS at a time (fx_score_stream_*, include/flexs_amd.h); 0 = never
STREAM_MIN_ROWS = 384
STREAM_STEP_ROWS = 256
_HAS_SCORE_SMALL = _strpack is not None and hasattr(_strpack, "score_small")
_HAS_PACK_STAGED = _strpack is not None and hasattr(_strpack, "pack_staged")


def small_plan(engine: "Engine", models: Sequence["NativeModel"], L: int, lut: np.ndarray, want_mean: bool) -> Optional[bytes]:
    """The argument block of `score_small` for one model list (struct SmallPlan in csrc/strpack.c); build once, reuse
    while the fx_model handles stay the same objects.  None when the helper is not built or the list does not fit."""
    import struct

    M = len(models)
    if not _HAS_SCORE_SMALL or not 1 <= M <= 16 or L < 1 or L > SMALL_CALL_BYTES:
        return None
    fn = C.cast(engine._lib.fx_score, C.c_void_p).value
    handles = [int(m.handle.value if hasattr(m.handle, "value") else m.handle) for m in models] + [0] * (16 - M)
    eh = engine.handle
    stream = [C.cast(getattr(engine._lib, name), C.c_void_p).value or 0
              for name in ("fx_score_stream_begin", "fx_score_stream_rows", "fx_score_stream_end")]
    return struct.pack("PPqqq16P256sPPPqq", fn, int(eh.value if hasattr(eh, "value") else eh), M, L, 2 if want_mean else 1,
                       *handles, np.ascontiguousarray(lut, np.uint8).tobytes(), *stream, STREAM_MIN_ROWS, STREAM_STEP_ROWS)


def score_small(engine: "Engine", plan: bytes, seqs, M: int, want_mean: bool) -> Optional[np.ndarray]:
    """get_fitness of a short list / tuple of str in one C call.  Returns the (N,) mean or the (N, M) matrix, or None
    when the call is not for this path (too big, not strings, ragged ...): the caller then takes the general path, which
    raises what the reference raises."""
    n = len(seqs)
    out = np.empty(n if want_mean else (n, M), np.float32)
    st = _strpack.score_small(plan, seqs, out)
    if st == 0:
        return out
    if st == -1 or 1000 < st < 2000:
        return None
    _raise(-(st - 2000) if st > 2000 else st, engine.handle)


# 1 (default since round 6) = launched-first calls hand their results out IN PLACE: arrays over registered host buffers of a pool, no copy
# (~10-20 us of a 1e5-string call).  The buffers are ordinary anonymous memory registered with the device (fx_result_alloc: mmap +
# hipHostRegister), so a fork()ed child inherits a result array copy-on-write like any NumPy array (round 5 used hipHostMalloc memory,
# which a child does not inherit: that kept the form opt-in; tests/test_gpu_api.py forks and reads).  0 = always copy into np.empty.
RESULTS_IN_PLACE = int(os.environ.get("FLEXS_AMD_RESULTS_IN_PLACE", 1))


class _ResultPool:
    """Registered (pinned, GPU-mapped) anonymous host buffers (fx_result_alloc) for the calls whose kernels write the scores straight into the memory the
    caller gets back: the array wraps the buffer and the buffer returns to the pool when the array's last view dies.  Bounded -- a
    caller that keeps every result alive simply gets ordinary arrays (a copy) from the 17th outstanding buffer or the 257th MiB on;
    buffers are never handed back to HIP before the process ends (an array may outlive the engine object)."""
    MAX_OUT, MAX_BYTES = 16, 256 << 20

    def __init__(self, engine: "Engine"):
        self._engine = engine
        self._free: Dict[int, list] = {}
        self._out = 0
        self._bytes = 0

    def take(self, nbytes: int):
        cap = 1 << max(int(nbytes - 1).bit_length(), 16)
        free = self._free.get(cap)
        if free:
            addr = free.pop()
        else:
            if self._out >= self.MAX_OUT or self._bytes + cap > self.MAX_BYTES:
                return None
            p = _vp()
            if self._engine._lib.fx_result_alloc(self._engine.handle, cap, C.byref(p)) != FX_OK:
                return None
            addr = p.value
            self._bytes += cap
        self._out += 1
        return addr, cap

    def give_back(self, addr: int, cap: int):
        self._out -= 1
        self._free.setdefault(cap, []).append(addr)

    def wrap(self, addr: int, cap: int, parts):
        """float32 arrays over one leased buffer: parts = [(byte offset, shape), ...]; the lease ends with the last of their views."""
        import weakref
        raw = (C.c_char * cap).from_address(addr)
        weakref.finalize(raw, self.give_back, addr, cap)
        return [np.frombuffer(raw, np.float32, int(np.prod(shape)), offset=off).reshape(shape) for off, shape in parts]


def _raise_pack_status(status: int):
    """What the reference raises for a batch `_strpack` could not pack (string_to_one_hot's np.array / str.index failures)."""
    errors = {1: ValueError("ragged sequence batch: all sequences must have the same length"), 2: ValueError("substring not found"),
              3: TypeError("sequences must be str")}
    if status in errors:
        raise errors[status]


def wants_chunked(sequences, L: int) -> bool:
    return (_strpack is not None and isinstance(sequences, (list, tuple)) and len(sequences) >= CHUNKED_MIN_ROWS
            and isinstance(sequences[0], str) and len(sequences[0]) == L)


class Engine:
    """One GPU's scoring engine (stream, scratch, deferred-error word).

    NOT thread-safe, as include/flexs_amd.h says of the handle: calls into one engine come from one thread at a time.  `Engine.get`
    hands every thread of the process the SAME engine per device, and the GIL is released inside the library calls, so two Python
    threads that score or train through it at the same moment must serialise themselves (one lock around their model calls) -- or,
    the deployment model, live in one process per GPU (`distributed.py`)."""

    _instances: Dict[int, "Engine"] = {}

    def __init__(self, device: int = 0):
        self._lib = lib()
        h = _vp()
        rc = self._lib.fx_engine_create(device, C.byref(h))
        if rc != FX_OK:
            msg = self._lib.fx_last_error(None).decode()
            raise RuntimeError(
                f"flexs_amd: cannot create a scoring engine on HIP device {device} "
                f"({status_name(rc)}: {msg}). An MI355X (gfx950) GPU is required; there is no CPU fallback."
            )
        self.handle = h
        self.device = device

    @classmethod
    def get(cls, device: Optional[int] = None) -> "Engine":
        if device is None:
            device = int(os.environ.get("FLEXS_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = lib().fx_device_count()
            if n > 0:
                device %= n
        if device not in cls._instances:
            cls._instances[device] = Engine(device)
        return cls._instances[device]

    def check(self, rc: int):
        if rc != FX_OK:
            _raise(rc, self.handle)

    def _results(self) -> "_ResultPool":
        pool = self.__dict__.get("_result_pool")
        if pool is None:
            pool = self.__dict__["_result_pool"] = _ResultPool(self)
        return pool

    # ---- options / sync / timing
    def set_option(self, key: str, value: int):
        self.check(self._lib.fx_engine_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        self.check(self._lib.fx_engine_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def set_stream(self, hip_stream: Optional[int]):
        self._torch_stream = None                         # (torch_stream() re-lends on its next use)
        self.check(self._lib.fx_engine_set_stream(self.handle, _vp(hip_stream) if hip_stream else None))

    def sync(self):
        self.check(self._lib.fx_engine_sync(self.handle))

    def error_word_dev(self, d_dst: int):
        """Stream-ordered: the float at device address `d_dst` = this engine's deferred error bits (see flexs_amd.h)."""
        self.check(self._lib.fx_engine_error_word_dev(self.handle, _vp(d_dst)))

    def torch_stream(self):
        """A torch.cuda.Stream lent to the engine (created on first use, then kept): torch-side work (uploads,
        RCCL collectives through stream waits) and the engine's kernels are ordered on it without host syncs."""
        st = getattr(self, "_torch_stream", None)
        if st is None:
            import torch

            with torch.cuda.device(self.device):
                st = torch.cuda.Stream()
            self.set_stream(st.cuda_stream)
            self._torch_stream = st
        return st

    COUNTER_NAMES = ("host_calls", "device_calls", "sequences", "forwards", "bytes_h2d", "bytes_d2h", "zero_copy_calls",
                     "pair_evals", "train_steps")

    def counters(self, reset: bool = False) -> dict:
        """Engine-side counters (fx_engine_counters): what went through this engine since creation / the last reset."""
        out = (C.c_int64 * 9)()
        self.check(self._lib.fx_engine_counters(self.handle, out, int(reset)))
        return dict(zip(self.COUNTER_NAMES, [int(v) for v in out]))

    def timer_start(self):
        self.check(self._lib.fx_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.check(self._lib.fx_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def time_score_planes(self, models, d_ascii: int, N: int, L: int, lut: np.ndarray, d_planes: int, stride: int,
                          reps: int) -> float:
        """Milliseconds for `reps` back-to-back scoring launches issued from C (one hipEvent pair around them)."""
        M = len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        ms = C.c_float()
        self.check(self._lib.fx_debug_time_score(self.handle, arr, M, _vp(d_ascii), N, L, _lut_ptr(lut), _vp(d_planes),
                                                 stride, reps, C.byref(ms)))
        return ms.value

    def trace_read(self) -> np.ndarray:
        """(1024 workgroups, 16 waves, 16 slots) uint64 in-kernel timeline of the last traced launch (option "trace")."""
        out = np.zeros((1024, 16, 16), np.uint64)
        self.check(self._lib.fx_debug_trace_read(self.handle, _ptr(out), out.size))
        return out

    def staging_rows(self, n: int, width: int) -> np.ndarray:
        """(n, width) uint8 view of the engine's pinned input staging area: marshal strings straight into it and
        hand it to `score` -- the call then skips its pageable-to-pinned copy.  Valid until the next call."""
        p = _vp()
        self.check(self._lib.fx_staging_input(self.handle, n * width, C.byref(p)))
        buf = (C.c_uint8 * max(n * width, 1)).from_address(p.value)
        return np.frombuffer(buf, np.uint8, n * width).reshape(n, width)

    # ---- scoring
    def score(self, models: Sequence["NativeModel"], seq_bytes: np.ndarray, lut: np.ndarray,
              want_matrix: bool = True, want_mean: bool = False):
        N, L = seq_bytes.shape
        M = len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        out_nm = np.empty((N, M), np.float32) if want_matrix else None
        out_mean = np.empty((N,), np.float32) if want_mean else None
        seq_bytes = np.ascontiguousarray(seq_bytes)
        self.check(self._lib.fx_score(self.handle, arr, M, _ptr(seq_bytes), N, L,
                                      _lut_ptr(lut), _ptr(out_nm), _ptr(out_mean)))
        return out_nm, out_mean

    def score_strings(self, models: Sequence["NativeModel"], seqs, L: int, lut: np.ndarray,
                      want_matrix: bool = True, want_mean: bool = False, chunks: int = 0):
        """`score` for a big list / tuple of str: the strings are packed into the pinned staging area in `chunks`
        pieces and each piece is submitted as soon as it is packed, so packing piece k+1 (host) overlaps the
        transfer and scoring of piece k (GPU).  Same results and exceptions as sequences_to_bytes + score."""
        N, M = len(seqs), len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        if chunks <= 0 and CHUNK_BYTES <= 0 and _HAS_PACK_STAGED:
            # launched first, packed behind: where the engine's kernels can wait for their rows (fx_score_begin_staged) the launch
            # latency and most of the kernel's run lie beside the packing instead of behind it
            lanes = min(_strpack.lanes_for(N * L), 16)
            p, w, base, stages, pitch = _vp(), _vp(), C.c_uint(0), C.c_int(0), C.c_int(0)
            nm_res = 4 * N * M if want_matrix else 0
            lease = self._results().take(nm_res + (4 * N if want_mean else 0)) if RESULTS_IN_PLACE else None
            rc = self._lib.fx_score_begin_staged(self.handle, arr, M, N, L, _lut_ptr(lut), int(want_matrix), int(want_mean), lanes,
                                                 C.byref(p), C.byref(w), C.byref(base), C.byref(stages), C.byref(pitch),
                                                 _vp(lease[0]) if lease else None, lease[1] if lease else 0)
            if rc == FX_OK:
                status, handed_out = 3, False
                try:
                    status = _strpack.pack_staged(seqs, L, p.value, stages.value, pitch.value, lanes, w.value, base.value)
                finally:
                    if status:
                        self._lib.fx_score_abandon(self.handle)
                    try:
                        if lease:
                            out_nm = out_mean = None
                            rc = self._lib.fx_score_finish(self.handle, None, None)
                            if status == 0 and rc == FX_OK:
                                parts = ([(0, (N, M))] if want_matrix else []) + ([(nm_res, (N,))] if want_mean else [])
                                views = self._results().wrap(lease[0], lease[1], parts)
                                handed_out = True
                                out_nm = views[0] if want_matrix else None
                                out_mean = views[-1] if want_mean else None
                        else:
                            out_nm = np.empty((N, M), np.float32) if want_matrix else None
                            out_mean = np.empty((N,), np.float32) if want_mean else None
                            rc = self._lib.fx_score_finish(self.handle, _ptr(out_nm), _ptr(out_mean))
                    finally:
                        if lease and not handed_out:
                            self._results().give_back(*lease)
                if status == 5:         # (legacy str objects need the GIL: the plain path)
                    return self.score(models, sequences_to_bytes(seqs, L=L, staging=self), lut, want_matrix=want_matrix, want_mean=want_mean)
                _raise_pack_status(status)
                self.check(rc)
                return out_nm, out_mean
            if lease:
                self._results().give_back(*lease)
            if rc != FX_EUNSUPPORTED:
                self.check(rc)
        if chunks <= 0:
            if CHUNK_BYTES > 0:
                chunks = min(max(N * L // CHUNK_BYTES, 1), 16)
            else:
                # the engine's plan: pieces of ~2 MB when its kernels read the staging area directly (zero-copy),
                # 4 MB pieces for big uploads, ONE piece otherwise (profiles/r3_e2e_ab.log)
                zc, pieces = C.c_int(0), C.c_int(1)
                self.check(self._lib.fx_plan_host_call(self.handle, arr, M, N, L, C.byref(zc), C.byref(pieces)))
                chunks = pieces.value
        if chunks <= 1:
            seq_bytes = sequences_to_bytes(seqs, L=L, staging=self)        # packed (several threads) straight into the pinned area
            return self.score(models, seq_bytes, lut, want_matrix=want_matrix, want_mean=want_mean)
        p = _vp()
        self.check(self._lib.fx_score_begin(self.handle, arr, M, N, L, _lut_ptr(lut), int(want_matrix),
                                            int(want_mean), C.byref(p)))
        staging = np.frombuffer((C.c_uint8 * (N * L)).from_address(p.value), np.uint8, N * L).reshape(N, L)
        step = -(-N // chunks)
        step += -step % 256
        status = 0
        try:
            for r0 in range(0, N, step):
                rows = min(step, N - r0)
                status = _strpack.pack(seqs, L, staging[r0:r0 + rows], r0, rows)
                if status:
                    break
                self.check(self._lib.fx_score_submit(self.handle, r0, rows))
        finally:
            out_nm = np.empty((N, M), np.float32) if want_matrix else None
            out_mean = np.empty((N,), np.float32) if want_mean else None
            rc = self._lib.fx_score_finish(self.handle, _ptr(out_nm), _ptr(out_mean))
        _raise_pack_status(status)
        self.check(rc)
        return out_nm, out_mean

    def score_dev(self, models: Sequence["NativeModel"], d_ascii: int, N: int, L: int, lut: np.ndarray,
                  d_out_nm: Optional[int], d_out_mean: Optional[int]):
        M = len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        self.check(self._lib.fx_score_dev(self.handle, arr, M, _vp(d_ascii), N, L, _lut_ptr(lut),
                                          _vp(d_out_nm) if d_out_nm else None,
                                          _vp(d_out_mean) if d_out_mean else None))

    def score_planes_dev(self, models: Sequence["NativeModel"], d_ascii: int, N: int, L: int, lut: np.ndarray,
                         d_planes: int, stride: int):
        M = len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        self.check(self._lib.fx_score_planes_dev(self.handle, arr, M, _vp(d_ascii), N, L, _lut_ptr(lut),
                                                 _vp(d_planes), stride))

    def ensemble_mean_planes_dev(self, d_planes: int, N: int, M: int, stride: int, d_out32: int):
        self.check(self._lib.fx_ensemble_mean_planes_dev(self.handle, _vp(d_planes), N, M, stride, _vp(d_out32)))

    def score_mean_planes_dev(self, models, d_ascii: int, N: int, L: int, lut: np.ndarray, d_planes: int, stride: int, d_out32: int):
        """score_planes_dev + ensemble_mean_planes_dev as one call."""
        arr = (_vp * len(models))(*[m.handle for m in models])
        self.check(self._lib.fx_score_mean_planes_dev(self.handle, arr, len(models), _vp(d_ascii), N, L, _lut_ptr(lut), _vp(d_planes), stride, _vp(d_out32)))

    def encode_onehot(self, seq_bytes: np.ndarray, lut: np.ndarray, A: int) -> np.ndarray:
        N, L = seq_bytes.shape
        out = np.empty((N, L, A), np.float32)
        seq_bytes = np.ascontiguousarray(seq_bytes)
        self.check(self._lib.fx_encode_onehot(self.handle, _ptr(seq_bytes), N, L, _lut_ptr(lut), A, _ptr(out)))
        return out

    def encode_onehot_dev(self, d_ascii: int, N: int, L: int, lut: np.ndarray, A: int, d_out: int):
        self.check(self._lib.fx_encode_onehot_dev(self.handle, _vp(d_ascii), N, L, _lut_ptr(lut), A, _vp(d_out)))

    def ensemble_mean(self, scores_nm: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(scores_nm, np.float32)
        out = np.empty(s.shape[0], np.float32)
        self.check(self._lib.fx_ensemble_reduce(self.handle, _ptr(s), s.shape[0], s.shape[1], None, _ptr(out), None))
        return out

    def ensemble_weighted_sum(self, scores_nm: np.ndarray, weights: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(scores_nm, np.float32)
        w = np.ascontiguousarray(weights, np.float64)
        out = np.empty(s.shape[0], np.float64)
        self.check(self._lib.fx_ensemble_reduce(self.handle, _ptr(s), s.shape[0], s.shape[1], _ptr(w), None, _ptr(out)))
        return out

    def ensemble_reduce_dev(self, d_scores: int, N: int, M: int, d_out32: int):
        self.check(self._lib.fx_ensemble_reduce_dev(self.handle, _vp(d_scores), N, M, None, _vp(d_out32), None))

    def argmax_decode(self, one_hot: np.ndarray, alphabet: str) -> np.ndarray:
        x = np.ascontiguousarray(one_hot, np.float64)
        P, L, A = x.shape
        if A > len(alphabet):
            # np.argmax could pick an index with no character: mirror the IndexError
            pass
        al = np.frombuffer(alphabet.encode("latin-1"), np.uint8)
        if A > al.shape[0]:
            al = np.concatenate([al, np.zeros(A - al.shape[0], np.uint8)])
        out = np.empty((P, L), np.uint8)
        self.check(self._lib.fx_argmax_decode(self.handle, _ptr(x), P, L, A, _ptr(np.ascontiguousarray(al)), _ptr(out)))
        return out

    def decode_score(self, models: Sequence["NativeModel"], one_hot: np.ndarray, alphabet: str, lut: np.ndarray,
                     want_matrix: bool = True, want_mean: bool = False):
        """(P, L, A) floats -> ((P, L) decoded bytes, (P, M) scores, (P,) mean) in one device round trip."""
        x = np.ascontiguousarray(one_hot, np.float64)
        P, L, A = x.shape
        if A != len(alphabet):
            raise ValueError(f"one-hot width {A} does not match the alphabet ({len(alphabet)} letters)")
        M = len(models)
        arr = (_vp * M)(*[m.handle for m in models])
        al = np.frombuffer(alphabet.encode("latin-1"), np.uint8)
        chars = np.empty((P, L), np.uint8)
        out_nm = np.empty((P, M), np.float32) if want_matrix else None
        out_mean = np.empty((P,), np.float32) if want_mean else None
        self.check(self._lib.fx_decode_score(self.handle, arr, M, _ptr(x), P, L, A, _ptr(np.ascontiguousarray(al)),
                                             _lut_ptr(lut), _ptr(chars), _ptr(out_nm), _ptr(out_mean)))
        return chars, out_nm, out_mean

    def min_dist(self, queries: np.ndarray, cache: np.ndarray, mode: int = FX_LEVENSHTEIN):
        q = np.ascontiguousarray(queries, np.uint8)
        c = np.ascontiguousarray(cache, np.uint8)
        Q, L = q.shape
        dist = np.empty(Q, np.int32)
        arg = np.empty(Q, np.int64)
        self.check(self._lib.fx_min_dist(self.handle, mode, _ptr(q), Q, _ptr(c), c.shape[0], L, _ptr(dist), _ptr(arg)))
        return dist, arg

    def mfma_probe(self, a64, b64, c256) -> np.ndarray:
        a = np.ascontiguousarray(a64, np.float32)
        b = np.ascontiguousarray(b64, np.float32)
        c = np.ascontiguousarray(c256, np.float32).reshape(64, 4)
        d = np.empty((64, 4), np.float32)
        self.check(self._lib.fx_debug_mfma_probe(self.handle, _ptr(a), _ptr(b), _ptr(c), _ptr(d)))
        return d

    def nam_combine(self, signal, noise, d, alpha_tab) -> np.ndarray:
        signal = np.ascontiguousarray(signal, np.float64)
        noise = np.ascontiguousarray(noise, np.float64)
        d = np.ascontiguousarray(d, np.int32)
        tab = np.ascontiguousarray(alpha_tab, np.float64)
        out = np.empty(signal.shape[0], np.float64)
        self.check(self._lib.fx_nam_combine(self.handle, signal.shape[0], _ptr(signal), _ptr(noise), _ptr(d),
                                            _ptr(tab), tab.shape[0], _ptr(out)))
        return out


class NativeModel:
    """fx_model handle: shape + device-resident weights of one surrogate."""

    def __init__(self, engine: Engine, kind: int, L: int, A: int, F: int, H: int, K: int):
        self.engine = engine
        h = _vp()
        rc = engine._lib.fx_model_create(engine.handle, kind, L, A, F, H, K, C.byref(h))
        if rc != FX_OK:
            _raise(rc, engine.handle)
        self.handle = h
        self.kind, self.L, self.A, self.F, self.H, self.K = kind, L, A, F, H, K
        self.num_params = engine._lib.fx_model_num_params(h)

    def set_weights(self, arrays: List[np.ndarray]):
        blob = np.concatenate([np.asarray(a, np.float32).ravel() for a in arrays])
        self.engine.check(self.engine._lib.fx_model_set_weights(self.handle, blob.ctypes.data_as(_f32p), blob.shape[0]))

    def get_blob(self) -> np.ndarray:
        blob = np.empty(self.num_params, np.float32)
        self.engine.check(self.engine._lib.fx_model_get_weights(self.handle, blob.ctypes.data_as(_f32p), blob.shape[0]))
        return blob

    def __del__(self):
        try:
            if self.handle:
                self.engine._lib.fx_model_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class NativeCache:
    """Device-resident, append-only NoisyAbstractModel key store."""

    def __init__(self, engine: Engine, L: int):
        self.engine = engine
        self.L = L
        h = _vp()
        engine.check(engine._lib.fx_cache_create(engine.handle, L, C.byref(h)))
        self.handle = h

    def __len__(self):
        return int(self.engine._lib.fx_cache_size(self.handle))

    def append(self, keys_u8: np.ndarray):
        k = np.ascontiguousarray(keys_u8, np.uint8)
        if k.shape[0]:
            self.engine.check(self.engine._lib.fx_cache_append(self.handle, _ptr(k), k.shape[0]))

    def min_dist(self, queries: np.ndarray, mode: int = FX_LEVENSHTEIN):
        q = np.ascontiguousarray(queries, np.uint8)
        Q = q.shape[0]
        dist = np.empty(Q, np.int32)
        arg = np.empty(Q, np.int64)
        self.engine.check(self.engine._lib.fx_cache_min_dist(self.handle, mode, _ptr(q), Q, _ptr(dist), _ptr(arg)))
        return dist, arg

    def nam_query(self, table: "NativeTable", queries: np.ndarray, E: np.ndarray, alpha_tab: np.ndarray, mode: int = FX_LEVENSHTEIN,
                  append: Optional[np.ndarray] = None):
        """NoisyAbstractModel's batch on a table landscape in one device round trip (fx_cache_nam_query): returns
        (fitness float64, distance int32, nearest-neighbour index int64, flags int32).  `append`: rows that join the
        cache before the search (as `append`)."""
        q = np.ascontiguousarray(queries, np.uint8)
        Q = q.shape[0]
        app = None if append is None or append.shape[0] == 0 else np.ascontiguousarray(append, np.uint8)
        E = np.ascontiguousarray(E, np.float64)
        tab = np.ascontiguousarray(alpha_tab, np.float64)
        out = np.empty(Q, np.float64)
        dist = np.empty(Q, np.int32)
        arg = np.empty(Q, np.int64)
        flags = np.empty(Q, np.int32)
        self.engine.check(self.engine._lib.fx_cache_nam_query(self.handle, table.handle, table.bits, _lut_ptr(table.lut), mode,
                                                              _ptr(app), 0 if app is None else app.shape[0], _ptr(q), Q,
                                                              _ptr(E), _ptr(tab), tab.shape[0], _ptr(out), _ptr(dist), _ptr(arg), _ptr(flags)))
        return out, dist, arg, flags

    def time_min_dist(self, queries: np.ndarray, mode: int = FX_LEVENSHTEIN, reps: int = 10) -> float:
        """Total milliseconds of `reps` back-to-back neighbour-search launches (fx_debug_time_min_dist, issued from C)."""
        q = np.ascontiguousarray(queries, np.uint8)
        ms = C.c_float(0.0)
        self.engine.check(self.engine._lib.fx_debug_time_min_dist(self.handle, mode, _ptr(q), q.shape[0], reps, C.byref(ms)))
        return float(ms.value)

    def density(self, queries: np.ndarray, fitness: np.ndarray, radius: int, mode: int = FX_LEVENSHTEIN):
        """fx_cache_density: (float64 densities, int32 neighbour counts) of the queries against every stored key."""
        q = np.ascontiguousarray(queries, np.uint8)
        f = np.ascontiguousarray(fitness, np.float64)
        if f.shape[0] < len(self):
            raise ValueError("one fitness value per stored key")
        dens = np.empty(q.shape[0], np.float64)
        cnt = np.empty(q.shape[0], np.int32)
        self.engine.check(self.engine._lib.fx_cache_density(self.handle, mode, _ptr(q), q.shape[0], int(radius), _ptr(f), _ptr(dens), _ptr(cnt)))
        return dens, cnt

    def distances(self, queries: np.ndarray, mode: int = FX_LEVENSHTEIN) -> np.ndarray:
        """(Q, C) uint8 matrix of min(distance, 255) against every stored key."""
        q = np.ascontiguousarray(queries, np.uint8)
        out = np.empty((q.shape[0], len(self)), np.uint8)
        if out.size:
            self.engine.check(self.engine._lib.fx_cache_distances(self.handle, mode, _ptr(q), q.shape[0], _ptr(out)))
        return out

    def __del__(self):
        try:
            if self.handle:
                self.engine._lib.fx_cache_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001
            pass


class NativeTable:
    """Device-resident look-up landscape: fitness = table[packed k-mer] (`lookup`), or a
    per-position sum over an (L, ncol) table (`additive_sum`)."""

    def __init__(self, engine: Engine, table: np.ndarray, alphabet: str, bits: int = 0, lut: np.ndarray = None):
        self.engine = engine
        self.bits = bits
        self.lut = make_lut(alphabet) if lut is None else np.ascontiguousarray(lut, np.uint8)
        self.shape = np.shape(table)
        t = np.ascontiguousarray(table, np.float64).ravel()
        h = _vp()
        engine.check(engine._lib.fx_table_create(engine.handle, _ptr(t), t.shape[0], C.byref(h)))
        self.handle = h

    def lookup(self, seq_bytes: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(seq_bytes, np.uint8)
        out = np.empty(b.shape[0], np.float64)
        if b.shape[0]:
            self.engine.check(self.engine._lib.fx_table_lookup(self.handle, _ptr(b), b.shape[0], b.shape[1],
                                                               _lut_ptr(self.lut), self.bits, _ptr(out)))
        return out

    def additive_sum(self, seq_bytes: np.ndarray) -> np.ndarray:
        """out[n] = sum_i table[i, lut[seq[n, i]]], float64, accumulated in position order."""
        b = np.ascontiguousarray(seq_bytes, np.uint8)
        L, ncol = self.shape
        if b.ndim != 2 or (b.shape[0] and b.shape[1] != L):
            raise ValueError(f"additive table expects rows of {L} bytes")
        out = np.empty(b.shape[0], np.float64)
        if b.shape[0]:
            self.engine.check(self.engine._lib.fx_table_additive(self.handle, _ptr(b), b.shape[0], L,
                                                                 _lut_ptr(self.lut), ncol, _ptr(out)))
        return out

    def __del__(self):
        try:
            if self.handle:
                self.engine._lib.fx_table_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001
            pass


# ---- host-only test hooks (usable without a GPU) ---------------------------
def debug_pack_weights(kind, L, A, F, H, K, arrays) -> np.ndarray:
    blob = np.concatenate([np.asarray(a, np.float32).ravel() for a in arrays])
    n = lib().fx_debug_packed_size(kind, L, A, F, H, K)
    out = np.empty(n, np.float32)
    rc = lib().fx_debug_pack_weights(kind, L, A, F, H, K, blob.ctypes.data_as(_f32p), blob.shape[0],
                                     out.ctypes.data_as(_f32p), n)
    if rc != FX_OK:
        raise ValueError(status_name(rc))
    return out


def debug_pack_layout(kind, L, A, F, H, K) -> dict:
    v = (C.c_int64 * 16)()
    rc = lib().fx_debug_pack_layout(kind, L, A, F, H, K, v)
    if rc != FX_OK:
        raise ValueError(status_name(rc))
    names = ["FT", "HT", "SG1", "off_first", "off_c2", "off_c3", "off_cb", "conv_floats", "off_d1", "off_d2",
             "off_d3", "off_db", "RLH", "total_floats", "off_w1p", "HTR"]
    return dict(zip(names, list(v)))


def mfma_per_tile(kind, L, A, F, H, K) -> int:
    """MFMA instructions (2048 FLOP each) issued per 16-sequence tile per member (host-side restatement of the
    kernels' loop bounds, pack.cpp)."""
    n = lib().fx_debug_mfma_per_tile(kind, L, A, F, H, K)
    if n < 0:
        raise ValueError(f"no MFMA kernel for kind={kind} L={L} A={A} F={F} H={H} K={K}")
    return int(n)


def debug_myers_strips(a: bytes, b: bytes, words_per_strip: int = 12) -> int:
    a = np.frombuffer(a, np.uint8)
    b = np.frombuffer(b, np.uint8)
    return lib().fx_debug_myers_strips(_ptr(a) if len(a) else None, len(a), _ptr(b) if len(b) else None, len(b), words_per_strip)


def debug_bounded_distance(a: bytes, b: bytes, K: int, hamming: bool = False) -> int:
    a = np.frombuffer(bytes(a), np.uint8)
    b = np.frombuffer(bytes(b), np.uint8)
    return lib().fx_debug_bounded_distance(_ptr(a) if len(a) else None, len(a), _ptr(b) if len(b) else None, len(b), K, int(hamming))


def debug_myers(a: bytes, b: bytes) -> int:
    a = np.frombuffer(bytes(a), np.uint8)
    b = np.frombuffer(bytes(b), np.uint8)
    return lib().fx_debug_myers(_ptr(a) if len(a) else None, len(a), _ptr(b) if len(b) else None, len(b))


# ---------------------------------------------------------------------------------------------------------------
# training (csrc/train.hip)
class FxFitJob(C.Structure):
    """`fx_fit_job` of include/flexs_amd.h."""
    _fields_ = [("kind", C.c_int), ("L", C.c_int), ("A", C.c_int), ("F", C.c_int), ("H", C.c_int), ("K", C.c_int),
                ("weights", _vp), ("adam_m", _vp), ("adam_v", _vp), ("step", C.c_int64), ("order", _vp),
                ("epochs", C.c_int), ("batch", C.c_int), ("keep", _vp), ("seed", C.c_uint64), ("step_loss", _vp)]


def train_orders(seed: int, n: int, epochs: int) -> np.ndarray:
    """fx_train_orders: (epochs, n) int32, one uniformly random permutation of 0 .. n - 1 per epoch, all from `seed` (host only)."""
    out = np.empty((epochs, n), np.int32)
    rc = lib().fx_train_orders(C.c_uint64(seed & (2 ** 64 - 1)), n, epochs, _ptr(out))
    if rc:
        raise ValueError(f"fx_train_orders failed ({rc})")
    return out


def train_fit(engine: Engine, jobs: List[dict], seq_bytes: np.ndarray, lut: np.ndarray, labels: np.ndarray):
    """fx_train_fit.  jobs: dicts with kind (int), L, A, F, H, K, weights / adam_m / adam_v (flat float32 arrays, updated
    IN PLACE), step (int), order (int32 array [epochs * steps * batch]), epochs, batch, optional keep (uint8) and seed;
    returns [(new step count, step_loss array)] per job."""
    seq_bytes = np.ascontiguousarray(seq_bytes, np.uint8)
    labels = np.ascontiguousarray(labels, np.float32)
    n, L = seq_bytes.shape
    arr = (FxFitJob * len(jobs))()
    keepalive = []
    for s, j in zip(arr, jobs):
        steps = j["epochs"] * (-(-n // j["batch"])) if n else 0
        order = np.ascontiguousarray(j["order"], np.int32)
        if order.size != steps * j["batch"]:
            raise ValueError("train_fit: `order` must hold epochs * ceil(n / batch) * batch entries")
        loss = np.zeros(max(steps, 1), np.float32)
        keep = None if j.get("keep") is None else np.ascontiguousarray(j["keep"], np.uint8)
        for name in ("weights", "adam_m", "adam_v"):
            a = j[name]
            if a.dtype != np.float32 or not a.flags.c_contiguous:
                raise ValueError(f"train_fit: {name} must be a contiguous float32 array (it is updated in place)")
        s.kind, s.L, s.A, s.F, s.H, s.K = j["kind"], j["L"], j["A"], j["F"], j["H"], j["K"]
        s.weights, s.adam_m, s.adam_v = j["weights"].ctypes.data, j["adam_m"].ctypes.data, j["adam_v"].ctypes.data
        s.step, s.order, s.epochs, s.batch = int(j["step"]), order.ctypes.data, j["epochs"], j["batch"]
        s.keep, s.seed, s.step_loss = (keep.ctypes.data if keep is not None else None), int(j.get("seed", 0)), loss.ctypes.data
        keepalive.append((order, keep, loss))
    engine.check(engine._lib.fx_train_fit(engine.handle, C.byref(arr), len(jobs), _ptr(seq_bytes), n, L, _lut_ptr(lut), _ptr(labels)))
    return [(int(s.step), k[2]) for s, k in zip(arr, keepalive)]


def debug_train_step_host(kind: int, L: int, A: int, F: int, H: int, K: int, weights: np.ndarray, adam_m: np.ndarray,
                          adam_v: np.ndarray, step: int, seq_bytes: np.ndarray, lut: np.ndarray, labels: np.ndarray,
                          keep: Optional[np.ndarray] = None, R: int = 16):
    """fx_debug_train_step_host (no GPU): one mini-batch step through the host build of the training kernels' source;
    weights / moments updated in place; returns (new step count, loss before the update)."""
    seq_bytes = np.ascontiguousarray(seq_bytes, np.uint8)
    labels = np.ascontiguousarray(labels, np.float32)
    st, loss = C.c_int64(step), C.c_float(0.0)
    keep = None if keep is None else np.ascontiguousarray(keep, np.uint8)
    rc = lib().fx_debug_train_step_host(kind, L, A, F, H, K, weights.ctypes.data, adam_m.ctypes.data, adam_v.ctypes.data, C.addressof(st),
                                        _ptr(seq_bytes), seq_bytes.shape[0], _lut_ptr(lut), _ptr(labels), _ptr(keep), R, C.addressof(loss))
    if rc:
        raise FxError(rc, f"fx_debug_train_step_host failed: {status_name(rc)}")
    return int(st.value), float(loss.value)
