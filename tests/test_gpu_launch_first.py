"""GPU tests of the launched-first host call (include/flexs_amd.h fx_score_begin_staged): a big list of str is scored by kernels that
were enqueued BEFORE the strings were packed and wait, tile by tile, for the packing threads.  Same bits as the packed-first call,
the reference's exceptions, a device that never hangs on a host that stops packing."""
import ctypes as C
import time

import numpy as np
import pytest

import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils

from gpu_common import eng, rand_seqs  # noqa: F401  (eng: the session fixture)

pytestmark = pytest.mark.gpu


def _model(kind, L, alpha, M):
    make = {"cnn": lambda s: bm.CNN(L, 32, 100, alpha, seed=s), "mlp": lambda s: bm.MLP(L, 100, alpha, seed=s),
            "ge": lambda s: bm.GlobalEpistasisModel(L, 100, alpha, seed=s)}[kind]
    members = [make(s) for s in range(M)]
    return members[0] if M == 1 else flexs_amd.Ensemble(members)


def _counts(eng):
    return eng.get_option("launch_first_calls"), eng.get_option("launch_first_redone")


@pytest.fixture
def launch_first(eng):
    if not eng.get_option("large_bar"):
        pytest.skip("the host cannot store into this device's memory (no large BAR): launched-first calls are not offered")
    yield eng
    eng.set_option("launch_first", 1)


# (kind, L, alphabet, members, strings, launched first?)  -- the last column is what the plan is expected to say on an MI355X:
# zero-copy shapes with at least two tiles per SIMD launch first; the protein CNN's pair form and copy-planned calls do not
CASES = [("cnn", 8, "TGCA", 1, 100_003, True), ("cnn", 8, "TGCA", 3, 70_001, True), ("cnn", 14, "UGCA", 1, 50_000, True),
         ("cnn", 50, "UGCA", 3, 33_000, True), ("mlp", 14, "UGCA", 1, 100_000, True), ("mlp", 14, "UGCA", 3, 40_001, True),
         ("ge", 14, "UGCA", 1, 100_000, None), ("mlp", 50, "UGCA", 1, 40_000, None), ("cnn", 90, s_utils.AAS, 1, 33_000, False),
         ("cnn", 8, "TGCA", 1, 32_768, True), ("cnn", 8, "TGCA", 3, 250_000, True), ("mlp", 14, "UGCA", 3, 16_384, True),
         ("cnn", 8, "TGCA", 1, 16_400, False)]


@pytest.mark.parametrize("kind,L,alpha,M,n,expect", CASES)
def test_launched_first_call_gives_the_packed_first_bits(launch_first, kind, L, alpha, M, n, expect):
    eng = launch_first
    model = _model(kind, L, alpha, M)
    _, seqs = rand_seqs(n, L, alpha, seed=n % 97)
    eng.set_option("launch_first", 0)
    c0 = _counts(eng)
    want = np.asarray(model.get_fitness(seqs)).copy()
    assert _counts(eng) == c0                                   # (the option is honoured)
    eng.set_option("launch_first", 1)
    got = np.asarray(model.get_fitness(seqs))
    c1 = _counts(eng)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert c1[1] == c0[1]                                       # nothing had to be redone
    if expect is not None:
        assert (c1[0] - c0[0] == 1) == expect, "the plan changed: update CASES (or the planner)"
    again = np.asarray(model.get_fitness(tuple(seqs)))           # tuples too; and the second call of a row finds warm buffers
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
    if M > 1:                                                    # the matrix path (BO's combine_with = identity)
        ident = flexs_amd.Ensemble(model.models, combine_with=lambda x: x)
        eng.set_option("launch_first", 0)
        want_nm = ident.get_fitness(seqs).copy()
        eng.set_option("launch_first", 1)
        assert np.array_equal(ident.get_fitness(seqs).view(np.uint32), want_nm.view(np.uint32))


def test_launched_first_call_raises_what_the_reference_raises(launch_first):
    eng = launch_first
    L, alpha, n = 8, "TGCA", 70_001
    ens = _model("cnn", L, alpha, 3)
    _, seqs = rand_seqs(n, L, alpha, seed=5)
    want = ens.get_fitness(seqs).copy()
    c0 = _counts(eng)
    assert c0[0] > 0
    # first tile of stage 0, the middle, the ragged last tile; a character outside the alphabet, outside latin-1, a wrong length, a non-str
    for pos, bad, exc in ((0, "TGCAZGCA", ValueError), (n // 2, "TGCAZGCA", ValueError), (n - 1, "TGCAZGCA", ValueError),
                          (n - 5, "TGCA", ValueError), (17, "TGCATGCAT", ValueError), (60_000, 7, TypeError), (3, "TGCATΔCA", ValueError)):
        broken = list(seqs)
        broken[pos] = bad
        with pytest.raises(exc):
            ens.get_fitness(broken)
        assert np.array_equal(ens.get_fitness(seqs), want)       # the engine is usable afterwards, and nothing of the failed call is left
    assert _counts(eng)[0] - c0[0] == 14
    assert ens.get_fitness([]).shape == (0,)


def test_explorer_size_calls_around_a_launched_first_call(launch_first):
    """The resident workgroups of explorer-size calls and a launched-first launch on the same engine: each leaves the other's answers alone."""
    eng = launch_first
    L, alpha = 8, "TGCA"
    ens = _model("cnn", L, alpha, 3)
    _, big = rand_seqs(60_000, L, alpha, seed=8)
    _, small = rand_seqs(40, L, alpha, seed=9)
    want_big, want_small = ens.get_fitness(big).copy(), ens.get_fitness(small).copy()
    for _ in range(3):
        for _ in range(5):
            assert np.array_equal(ens.get_fitness(small), want_small)
        assert np.array_equal(ens.get_fitness(big), want_big)
    assert np.array_equal(ens.get_fitness(small), want_small)


def test_a_host_that_stops_packing_does_not_hang_the_device(launch_first):
    """The kernels of a launched-first call give up on rows that do not come within 0.25 s (FX_ERR_STARVED) instead of waiting for
    ever; fx_score_finish then runs the launch once more over the rows the caller has packed since."""
    eng = launch_first
    if not _native._HAS_PACK_STAGED:
        pytest.skip("no _strpack.pack_staged in this build")
    L, alpha, n = 8, "TGCA", 50_000
    model = _model("cnn", L, alpha, 1)
    _, seqs = rand_seqs(n, L, alpha, seed=3)
    want = model.get_fitness(seqs).copy()
    lib, nm, lut = eng._lib, model.native(), model._lut
    arr = (_native._vp * 1)(nm.handle)
    p, w, base, stages, pitch = _native._vp(), _native._vp(), C.c_uint(0), C.c_int(0), C.c_int(0)
    redone = _counts(eng)[1]
    rc = lib.fx_score_begin_staged(eng.handle, arr, 1, n, L, _native._lut_ptr(lut), 1, 0, 4, C.byref(p), C.byref(w), C.byref(base),
                                   C.byref(stages), C.byref(pitch), None, 0)
    assert rc == _native.FX_OK and stages.value >= 2 and pitch.value == 128
    time.sleep(0.4)                                              # nobody packs: the waves time out
    assert _native._strpack.pack_staged(seqs, L, p.value, stages.value, pitch.value, 4, w.value, base.value) == 0
    out = np.empty((n, 1), np.float32)
    eng.check(lib.fx_score_finish(eng.handle, _native._ptr(out), None))
    assert np.array_equal(out[:, 0], want)
    assert _counts(eng)[1] == redone + 1
    # ... and a caller that gives up altogether: finish only waits
    rc = lib.fx_score_begin_staged(eng.handle, arr, 1, n, L, _native._lut_ptr(lut), 1, 0, 4, C.byref(p), C.byref(w), C.byref(base),
                                   C.byref(stages), C.byref(pitch), None, 0)
    assert rc == _native.FX_OK
    eng.check(lib.fx_score_abandon(eng.handle))
    eng.check(lib.fx_score_finish(eng.handle, _native._ptr(out), None))
    assert np.array_equal(model.get_fitness(seqs), want)
    # misuse: no call in flight
    assert lib.fx_score_abandon(eng.handle) == _native.FX_ESTATE


@pytest.mark.parametrize("L,alpha,M,n", [(8, "TGCA", 3, 70_001), (14, "UGCA", 1, 40_000), (8, "TGCA", 1, 6_000), (50, "UGCA", 3, 9_001), (23, "TGCA", 2, 33_000)])
def test_host_resident_bytes_through_lds_give_the_same_bits(launch_first, L, alpha, M, n):
    """Zero-copy host calls of the 4-letter CNN copy a tile's bytes into LDS with one wide load (engine option cnn_stage_host = 1)
    instead of a byte load over PCIe per position: the same walk, the same bits, launched first or not."""
    eng = launch_first
    model = _model("cnn", L, alpha, M)
    b, seqs = rand_seqs(n, L, alpha, seed=11)
    got = {}
    try:
        for stage in (0, 1, 2):
            eng.set_option("cnn_stage_host", stage)
            for first in (0, 1):
                eng.set_option("launch_first", first)
                got[stage, first] = np.asarray(model.get_fitness(seqs)).copy()
            members = model.models if M > 1 else [model]                      # (packed bytes in: the plain host call)
            nm, mean = eng.score([m.native() for m in members], b, members[0]._lut, want_matrix=(M == 1), want_mean=(M > 1))
            got[stage, "bytes"] = (nm[:, 0] if M == 1 else mean).copy()
    finally:
        eng.set_option("cnn_stage_host", 1)
    want = got[0, 0].view(np.uint32)
    for key, v in got.items():
        assert np.array_equal(v.view(np.uint32), want), key


def test_results_in_place_survive_a_fork(launch_first, monkeypatch):
    """Round 6: results in place are the default.  The buffers are anonymous memory registered with the device (fx_result_alloc: mmap +
    hipHostRegister), so a fork()ed child that reads a result array it inherited sees the parent's values (round 5's hipHostMalloc
    buffers were not inherited by a child: it would have faulted) -- and the parent's buffer still takes the next call's scores."""
    import os

    assert _native.RESULTS_IN_PLACE == 1 or os.environ.get("FLEXS_AMD_RESULTS_IN_PLACE") == "0"
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 1)
    L, alpha, n = 8, "TGCA", 50_000
    ens = _model("cnn", L, alpha, 3)
    _, seqs = rand_seqs(n, L, alpha, seed=21)
    got = ens.get_fitness(seqs)
    assert not got.flags.owndata                                  # over a leased buffer
    want = got.copy()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                                  # child: plain reads of inherited memory, no HIP call
        try:
            ok = bool(np.array_equal(got, want)) and float(got.sum()) == float(want.sum())
            os.write(w, b"1" if ok else b"0")
        finally:
            os._exit(0)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    answer = os.read(r, 1)
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child died: status {status}"
    assert answer == b"1"
    # the parent goes on: the same buffers take the next calls (the fork write-protected their pages for a moment)
    again = ens.get_fitness(seqs)
    assert np.array_equal(again, want) and np.array_equal(got, want)
    del got, again
    _, seqs2 = rand_seqs(n, L, alpha, seed=22)
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 0)
    want2 = ens.get_fitness(seqs2)
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 1)
    assert np.array_equal(ens.get_fitness(seqs2), want2)


def test_results_in_place_are_ordinary_arrays_over_leased_pinned_buffers(launch_first, monkeypatch):
    """FLEXS_AMD_RESULTS_IN_PLACE=1: a launched-first call's kernels write into a pinned buffer that the returned array wraps (no copy);
    the buffer goes back to the pool with the array's last view, a caller that hoards results gets ordinary arrays, a failed call leaks
    nothing."""
    import gc
    eng = launch_first
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 1)
    L, alpha, n = 8, "TGCA", 70_001
    ens = _model("cnn", L, alpha, 3)
    _, seqs = rand_seqs(n, L, alpha, seed=13)
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 0)
    want = ens.get_fitness(seqs).copy()
    want_nm = flexs_amd.Ensemble(ens.models, combine_with=lambda x: x).get_fitness(seqs).copy()
    monkeypatch.setattr(_native, "RESULTS_IN_PLACE", 1)
    pool = eng._results()
    out0 = pool._out
    got = ens.get_fitness(seqs)
    assert got.dtype == np.float32 and got.shape == (n,) and np.array_equal(got, want)
    assert pool._out == out0 + 1 and not got.flags.owndata
    view = got[10:20]
    got += 1.0                                                   # writable, like any result array
    assert np.array_equal(got, want + 1.0)
    del got
    gc.collect()
    assert pool._out == out0 + 1                                 # the view keeps the lease
    assert np.array_equal(view, want[10:20] + 1.0)
    del view
    gc.collect()
    assert pool._out == out0
    # matrix and mean of one call share a lease
    nm, mean = eng.score_strings([m.native() for m in ens.models], seqs, L, ens.models[0]._lut, want_matrix=True, want_mean=True)
    assert np.array_equal(nm, want_nm) and np.array_equal(mean, want) and pool._out == out0 + 1
    del nm
    gc.collect()
    assert pool._out == out0 + 1 and np.array_equal(mean, want)
    del mean
    gc.collect()
    assert pool._out == out0
    # a hoarder: beyond the pool's bound the results are ordinary arrays
    kept = [ens.get_fitness(seqs) for _ in range(_native._ResultPool.MAX_OUT + 3)]
    assert all(np.array_equal(k, want) for k in kept)
    assert pool._out == _native._ResultPool.MAX_OUT and sum(k.flags.owndata for k in kept) >= 3
    del kept
    gc.collect()
    assert pool._out == out0
    # a failed call gives its lease back
    broken = list(seqs)
    broken[n // 3] = "TGCAZGCA"
    with pytest.raises(ValueError):
        ens.get_fitness(broken)
    broken[n // 3] = "TGC"
    with pytest.raises(ValueError):
        ens.get_fitness(broken)
    assert pool._out == out0 and np.array_equal(ens.get_fitness(seqs), want)


@pytest.mark.parametrize("kind,L,alpha,M,n", [("ge", 90, s_utils.AAS, 8, 100_000), ("ge", 90, s_utils.AAS, 3, 40_003), ("ge", 237, s_utils.AAS, 8, 20_000),
                                              ("ge", 50, "UGCA", 5, 70_001)])
def test_relay_through_member_zero_gives_the_same_bits(launch_first, kind, L, alpha, M, n):
    """Ensembles whose members would each read the rows over PCIe again (the "copy" plan): in a launched-first call member 0's
    workgroups pass every tile on through device memory and the other members read it there (FxRelay, engine option launch_relay).
    Same bits as pack -> upload -> launch, call after call (the tiles travel between workgroups on different XCDs past the caches),
    mean and matrix paths, and the reference's exceptions."""
    eng = launch_first
    ens = _model(kind, L, alpha, M)
    ident = flexs_amd.Ensemble(ens.models, combine_with=lambda x: x)
    _, seqs = rand_seqs(n, L, alpha, seed=n % 89)
    try:
        eng.set_option("launch_relay", 0)
        want, want_nm = ens.get_fitness(seqs).copy(), ident.get_fitness(seqs).copy()
        eng.set_option("launch_relay", 1)
        c0 = eng.get_option("launch_relay_calls")
        redone = eng.get_option("launch_first_redone")
        for _ in range(6):
            assert np.array_equal(ens.get_fitness(seqs).view(np.uint32), want.view(np.uint32))
        assert np.array_equal(ident.get_fitness(seqs).view(np.uint32), want_nm.view(np.uint32))
        relayed = eng.get_option("launch_relay_calls") - c0
        assert relayed in (0, 7) and eng.get_option("launch_first_redone") == redone
        if (kind, L, M) == ("ge", 90, 8):
            assert relayed == 7, "BASELINE config C4's shape is expected to take the relay on an MI355X"
        # the plain host call on packed bytes (an ndarray of bytes, a C caller): no upload in front of the launch, the same relay
        b, _ = rand_seqs(n, L, alpha, seed=n % 89)
        nat, lut = [m.native() for m in ens.models], ens.models[0]._lut
        c1 = eng.get_option("launch_relay_calls")
        for want_matrix in (False, True):
            nm, mean = eng.score(nat, b, lut, want_matrix=want_matrix, want_mean=True)
            assert np.array_equal(mean.view(np.uint32), want.view(np.uint32))
            assert nm is None or np.array_equal(nm.view(np.uint32), want_nm.view(np.uint32))
        assert eng.get_option("launch_relay_calls") - c1 in (0, 2) and (relayed == 0) == (eng.get_option("launch_relay_calls") == c1)
        bad = np.array(b, copy=True)
        bad[n - 3, L - 1] = ord("!")
        with pytest.raises(ValueError):
            eng.score(nat, bad, lut, want_matrix=False, want_mean=True)
        assert np.array_equal(eng.score(nat, b, lut, want_matrix=False, want_mean=True)[1], want)
        broken = list(seqs)
        broken[n // 2] = seqs[0][:-1]
        with pytest.raises(ValueError):
            ens.get_fitness(broken)
        broken[n // 2] = seqs[0][:-1] + "!"
        with pytest.raises(ValueError):
            ens.get_fitness(broken)
        assert np.array_equal(ens.get_fitness(seqs), want)
    finally:
        eng.set_option("launch_relay", 1)


_TWO_PROC = r"""
import os, sys, time, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
ens = flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=s) for s in range(8)])
cnn = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=s) for s in range(3)])
a = synth.bytes_to_strings(synth.random_sequence_bytes(60_000, 90, s_utils.AAS, 7))
b = synth.bytes_to_strings(synth.random_sequence_bytes(70_001, 8, "TGCA", 8))
ens.get_fitness(a[:100]); cnn.get_fitness(b[:100])
open(sys.argv[2] + ".ready", "w").close()
while not os.path.exists(sys.argv[3] + ".ready"):
    time.sleep(0.001)
h = hashlib.sha256()
for _ in range(12):
    h.update(ens.get_fitness(a).tobytes()); h.update(cnn.get_fitness(b).tobytes())
eng = _native.Engine.get()
print(h.hexdigest(), eng.get_option("launch_first_calls"), eng.get_option("launch_relay_calls"), eng.get_option("launch_first_redone"))
"""


def test_two_processes_on_one_gpu_with_launched_first_calls(launch_first, tmp_path):
    """Two processes whose persistent launches wait on their own hosts (and, in a relay, on their own member-0 workgroups) share one
    GPU: whichever launch the device runs first, both get the single-process bits -- a launch that starves because the other process
    holds the CUs is run again (and a relay, if need be, without the relay)."""
    import hashlib
    import subprocess
    import sys
    eng = launch_first
    ens = _model("ge", 90, s_utils.AAS, 8)
    cnn = _model("cnn", 8, "TGCA", 3)
    a = synth.bytes_to_strings(synth.random_sequence_bytes(60_000, 90, s_utils.AAS, 7))
    b = synth.bytes_to_strings(synth.random_sequence_bytes(70_001, 8, "TGCA", 8))
    h = hashlib.sha256()
    wa, wb = ens.get_fitness(a), cnn.get_fitness(b)
    for _ in range(12):
        h.update(wa.tobytes()); h.update(wb.tobytes())
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    marks = [str(tmp_path / "p0"), str(tmp_path / "p1")]
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_PROC, root, marks[i], marks[1 - i]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for i in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        digest, first, relayed, redone = out.strip().split()[-4:]
        assert digest == h.hexdigest(), (out, err[-500:])
        assert int(first) >= 24 and int(relayed) >= 12


@pytest.mark.parametrize("kind,L,alpha,M", [("ge", 90, s_utils.AAS, 8), ("cnn", 8, "TGCA", 3), ("mlp", 14, "UGCA", 1)])
def test_alternating_batches_leave_nothing_behind(launch_first, kind, L, alpha, M):
    """Consecutive launched-first (and relay) calls on DIFFERENT batches of one shape, and of different sizes: bytes of the previous
    call left in a cache, in the staging area, in the relay area or behind a tile flag would show as the other batch's scores."""
    eng = launch_first
    model = _model(kind, L, alpha, M)
    batches = []
    try:
        eng.set_option("launch_first", 0)
        eng.set_option("launch_relay", 0)
        for k, n in enumerate((70_000, 70_000, 40_001, 90_000)):
            _, seqs = rand_seqs(n, L, alpha, seed=500 + k)
            batches.append((seqs, np.asarray(model.get_fitness(seqs)).copy()))
        eng.set_option("launch_first", 1)
        eng.set_option("launch_relay", 1)
        for _ in range(5):
            for seqs, want in batches:
                assert np.array_equal(np.asarray(model.get_fitness(seqs)).view(np.uint32), want.view(np.uint32))
    finally:
        eng.set_option("launch_first", 1)
        eng.set_option("launch_relay", 1)


@pytest.mark.parametrize("kind,L,alpha,M,n", [("ge", 90, s_utils.AAS, 8, 70_001), ("mlp", 14, "UGCA", 1, 60_000), ("mlp", 14, "UGCA", 3, 40_003),
                                              ("ge", 90, s_utils.AAS, 3, 40_000), ("ge", 237, s_utils.AAS, 8, 20_000)])
def test_next_tile_prefetch_gives_the_same_bits(launch_first, kind, L, alpha, M, n):
    """MLP / GE launches that read their rows from host memory may claim a tile ahead and ask for its bytes straight into a second
    LDS scratch (engine option dense_prefetch: 1 = in a relay of at least four members, 2 = always, 0 = never): the same bits from
    strings and from packed bytes, with another batch in between."""
    eng = launch_first
    model = _model(kind, L, alpha, M)
    members = model.models if M > 1 else [model]
    b, seqs = rand_seqs(n, L, alpha, seed=31)
    _, other = rand_seqs(n, L, alpha, seed=32)
    got = {}
    try:
        for pf in (0, 1, 2):
            eng.set_option("dense_prefetch", pf)
            model.get_fitness(other)
            got[pf, "str"] = np.asarray(model.get_fitness(seqs)).copy()
            nm, mean = eng.score([m.native() for m in members], b, members[0]._lut, want_matrix=(M == 1), want_mean=(M > 1))
            got[pf, "bytes"] = (nm[:, 0] if M == 1 else mean).copy()
    finally:
        eng.set_option("dense_prefetch", 1)
    want = got[0, "str"].view(np.uint32)
    for key, v in got.items():
        assert np.array_equal(v.view(np.uint32), want), key
