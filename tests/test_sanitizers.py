"""ASAN / UBSAN builds of the host-compilable native sources (SURVEY.md section 5 aux: sanitizer build).

* `tests/native/sanitize_host.cpp`: csrc/train_core.h (the training kernels' source, threads as loops), csrc/host_collect.cc (answer collection) and csrc/myers.h
  (register and strip forms of the bit-parallel Levenshtein) under  g++ -fsanitize=address,undefined  with exact-size heap
  buffers: an index past any array, a signed overflow or an invalid shift in the kernels' index arithmetic aborts the run.
* `tests/native/simt_train.cpp`: the DEVICE branches of csrc/train_core.h under a SIMT emulator (a host thread per GPU thread) with
  ThreadSanitizer and AddressSanitizer.
* `csrc/strpack.c` (the CPython packing helper and its thread pool) rebuilt with the sanitizers and driven from a Python
  child process that preloads the sanitizer runtimes: bytes, error statuses, sub-ranges, several threads."""
import os
import subprocess
import sys
import sysconfig

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_kernel_sources_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "sanitize_host"
    r = subprocess.run(["g++", "-std=c++17", *SAN, os.path.join(ROOT, "tests", "native", "sanitize_host.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "unsupported" in r.stderr:
        pytest.skip("sanitizers not available to this g++")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "sanitize_host: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _host_clang():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(cand):
            return cand
    return None


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_device_branches_of_the_training_source_under_a_simt_emulator(tmp_path, sanitizer):
    """tests/native/simt_train.cpp: the DEVICE branches of csrc/train_core.h (FXT_EMUL) run by one host thread per GPU thread -- the
    MFMA as a rendezvous of a wave's 64 threads, barriers as pthread barriers, LDS as a heap array of exactly the bytes the host code
    asks for.  Plain rows, rotated rows and the staged conv kernels (`train_swizzle` 1 / 2, written when no GPU time was left) must
    give the gradient partials and weights of the HOST build of the same source BIT FOR BIT; under ThreadSanitizer a missing barrier
    is a data race (checked by removing one), under AddressSanitizer an index past the workspace or the staging buffer a report."""
    cxx = _host_clang()
    if cxx is None:
        pytest.skip("no host clang++ (the emulator needs ext_vector_type)")
    exe = tmp_path / "simt_train"
    src = [os.path.join(ROOT, "tests", "native", f) for f in ("simt_ref.cpp", "simt_train.cpp")]
    r = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all", *src, "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and ("unsupported" in r.stderr or "cannot find" in r.stderr):
        pytest.skip("sanitizer runtime not available to this clang++")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), "quick"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    if r.returncode != 0 and ("Resource temporarily unavailable" in r.stderr or "std::system_error" in r.stderr or "failed to create thread" in r.stderr):
        pytest.skip("this environment does not let a process start a workgroup's worth of threads")
    assert r.returncode == 0 and "simt_train: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_string_packing_helper_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    so = tmp_path / ("_strpack" + sysconfig.get_config_var("EXT_SUFFIX"))
    r = subprocess.run(["gcc", *SAN, "-fPIC", "-shared", "-pthread", "-I", sysconfig.get_paths()["include"],
                        os.path.join(ROOT, "flexs_amd", "csrc", "strpack.c"), "-o", str(so)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    child = r'''
import sys, threading
sys.path.insert(0, sys.argv[1])
import _strpack as sp
L, N = 24, 30000
seqs = ["".join(chr(65 + (i * 7 + j * 3) % 20) for j in range(L)) for i in range(N)]
want = "".join(seqs).encode()
for threads in (1, 2, 5, 8):
    sp.set_threads(threads)
    out = bytearray(N * L)
    assert sp.pack(seqs, L, out) == 0 and bytes(out) == want
    part = bytearray(1001 * L)
    assert sp.pack(seqs, L, part, 777, 1001) == 0 and bytes(part) == want[777 * L:1778 * L]
sp.set_threads(4)
out = bytearray(N * L)
for where, bad, status in ((25000, "A" * (L - 1), 1), (9000, "A" * (L - 1) + "Δ", 2), (300, 7, 3)):
    broken = list(seqs); broken[where] = bad
    assert sp.pack(broken, L, out) == status
try:
    sp.pack(seqs, L, bytearray(10))
    raise SystemExit("short buffer accepted")
except ValueError:
    pass
errs = []
def run():
    o = bytearray(N * L)
    for _ in range(4):
        if sp.pack(seqs, L, o) != 0 or bytes(o) != want: errs.append(1)
ts = [threading.Thread(target=run) for _ in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs
# the argmax decode (scalar rows and the 256-bit form) on exact-size buffers, over the thread pool and its hot window
import array, random, struct
for A in (3, 8, 20, 23):
    rows = 5000
    x = array.array("d", [random.random() for _ in range(rows * A)])
    x[7 * A + 1] = float("nan")
    alpha = bytes(range(65, 65 + A))
    for threads in (1, 4):
        sp.set_threads(threads)
        out = bytearray(rows)
        assert sp.decode_argmax(x, rows, A, alpha, out) == 0
        for r in (0, 7, 4999):
            row = x[r * A:(r + 1) * A]
            want_i = next((i for i, v in enumerate(row) if v != v), max(range(A), key=lambda i: (row[i], -i)))
            assert out[r] == alpha[want_i], (A, r)
# one Adalead tree level: every draw through the callables, strings of exact size
random.seed(5)
nodes = [(i, "".join(random.choice("ACGT") for _ in range(9))) for i in range(12)]
idxs, kids = sp.adalead_children(nodes, 2, "ACGT", {nodes[0][1]}, {nodes[1][1]: 1.0}, random.random, random.getrandbits)
assert len(kids) == 12 and all(len(k) == 9 for k in kids) and len(idxs) == 12
# population_step and score_small through a plan whose function is a ctypes callback
import ctypes as C
FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint8), C.c_longlong, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))
def fake(e, models, M, ascii, Nq, Lq, lut, out_nm, out_mean):
    for i in range(Nq): out_mean[i] = float(sum(ascii[i * Lq + j] for j in range(Lq)))
    return 0
fn = FN(fake)
plan = struct.pack("PPqqq16P256sPPPqq", C.cast(fn, C.c_void_p).value, 1, 1, 6, 2, *([1] + [0] * 15), bytes(256), 0, 0, 0, 0, 0)
xs = array.array("d", [random.random() for _ in range(9 * 6 * 4)])
chars, scores = bytearray(9 * 6), array.array("f", [0.0] * 9)
st, names = sp.population_step(plan, xs, 9, 4, b"ACGT", chars, scores)
assert st == 0 and len(names) == 9 and all(scores[i] == float(sum(chars[i * 6:(i + 1) * 6])) for i in range(9))
out = array.array("f", [0.0] * 3)
assert sp.score_small(plan, ["ACGTAC", "TTTTTT", "GGGGGG"], out) == 0 and out[1] == 6.0 * ord("T")
print("strpack sanitized: ok")
'''
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-c", child, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "strpack sanitized: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
