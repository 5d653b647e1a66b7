"""Round-3 soak: (1) the seeded random-shape scoring sweep over 300 more seeds; (2) the hand-written training step on 150 random
shapes against oracle/train_np.py (two steps each); (3) 400 get_fitness(list[str]) calls of random size through every host-call
plan (zero-copy / copies / pieces) against the one-piece device result; (4) edit distances of random long rows (strips) against the
C oracle; (5) interleaved training + scoring on the same engine for 200 rounds (the explorer loop's pattern)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_parity as T
import test_train_native as TN
import flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
from oracle import c_oracle, ref_np

eng = _native.Engine.get(0)
eng.set_option("poison_outputs", 1)
fails = 0
t_start = time.time()


def fail(tag, ex):
    global fails
    fails += 1
    print("FAIL", tag, "->", str(ex)[:300], flush=True)


# (1) scoring sweep
SEED0 = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for seed in range(SEED0, SEED0 + 300):
    kind, alpha, A, L, H, F, K, M, n = T._random_case(seed)
    try:
        natives, ws = zip(*[T.make_native(eng, kind, L, A, H, F, K, seed=500 + 7 * seed + m) for m in range(M)])
        lut = _native.make_lut(alpha)
        b, seqs = T.rand_seqs(n, L, alpha, seed=seed)
        got, mean = eng.score(list(natives), b, lut, want_matrix=True, want_mean=True)
        for m in range(M):
            T.assert_scores(got[:, m], ref_np.keras_fitness(seqs, alpha, kind, ws[m], exact=True), f"member {m}")
        assert np.array_equal(mean, np.mean(got, axis=1))
    except Exception as ex:                                   # noqa: BLE001
        fail(f"score seed {seed} {kind} L={L} H={H} F={F} K={K} M={M} n={n}", ex)
print(f"[{time.time() - t_start:.0f}s] scoring sweep done", flush=True)

# (2) training sweep
rng = np.random.default_rng(99 + SEED0)
for draw in range(150):
    kind = ("cnn", "mlp", "ge")[draw % 3]
    alphabet = ("UGCA", ref_np.AAS, "01", "TGCA")[int(rng.integers(0, 4))]
    A = len(alphabet)
    K = int(rng.integers(2, 8)) if kind == "cnn" else 0
    L = int(rng.integers(max(K, 1), 60))
    F = int(rng.integers(1, 48)) if kind == "cnn" else 0
    H = int(rng.integers(1, 140))
    rows = int(rng.integers(1, 200))
    lut = _native.make_lut(alphabet)

    def step_fn(w, m, v, t, b, y, keep, kind=kind, L=L, A=A, F=F, H=H, K=K, rows=rows, lut=lut):
        t2, loss = TN._fit_once(eng, kind, L, A, F, H, K, w, m, v, t, b, y, np.arange(rows, dtype=np.int32), 1, rows,
                                keep=None if keep is None else keep[None], lut=lut)
        return t2, float(loss[0])

    try:
        TN.check_against_oracle(step_fn, kind, L, alphabet, F, H, K, rows, steps=2)
    except Exception as ex:                                   # noqa: BLE001
        fail(f"train draw {draw} {kind} L={L} A={A} F={F} H={H} K={K} rows={rows}", ex)
print(f"[{time.time() - t_start:.0f}s] training sweep done", flush=True)

# (3) host-call plans
members = [bm.CNN(8, 32, 100, "TGCA", seed=s) for s in range(3)]
ens = flexs_amd.Ensemble(members)
ge = flexs_amd.Ensemble([bm.GlobalEpistasisModel(40, 100, s_utils.AAS, seed=s) for s in range(4)])
for it in range(400):
    model, L, alpha = (ens, 8, "TGCA") if it % 2 == 0 else (ge, 40, s_utils.AAS)
    n = int(rng.choice([1, 17, 1000, 32768, 40_001, 100_000, 250_000]))
    b = synth.random_sequence_bytes(n, L, alpha, 1000 + it)
    seqs = synth.bytes_to_strings(b)
    try:
        eng.set_option("zero_copy_mode", int(rng.integers(-1, 2)))
        _native.CHUNK_BYTES = int(rng.choice([0, 0, n * L // 3 + 1, 1 << 40]))
        got = model.get_fitness(seqs)
        eng.set_option("zero_copy_mode", 0); _native.CHUNK_BYTES = 1 << 40
        want = model.get_fitness(np.array(seqs, dtype="S"))
        assert np.array_equal(got, want)
    except Exception as ex:                                   # noqa: BLE001
        fail(f"plan it {it} n={n} L={L}", ex)
    finally:
        eng.set_option("zero_copy_mode", -1); _native.CHUNK_BYTES = 0
print(f"[{time.time() - t_start:.0f}s] host-call plans done", flush=True)

# (4) long edit distances
for it in range(40):
    L = int(rng.integers(769, 2200)); nsym = int(rng.choice([4, 20])); Cn, Q = int(rng.integers(1, 60)), int(rng.integers(1, 6))
    base = rng.integers(65, 65 + nsym, (1, L)).astype(np.uint8)
    cache = np.repeat(base, Cn, 0); mut = rng.random((Cn, L)) < 0.05; cache[mut] = rng.integers(65, 65 + nsym, mut.sum())
    rot = rng.random(Cn) < 0.4; cache[rot] = np.roll(cache[rot], int(rng.integers(1, 4)), axis=1)
    q = cache[rng.integers(0, Cn, Q)].copy(); qm = rng.random((Q, L)) < 0.03; q[qm] = rng.integers(65, 65 + nsym, qm.sum())
    try:
        for mode in (0, 1):
            d_want, a_want = c_oracle.min_dist(q, cache, mode)
            d_got, a_got = eng.min_dist(q, cache, mode)
            assert np.array_equal(d_got, d_want) and np.array_equal(a_got, a_want)
    except Exception as ex:                                   # noqa: BLE001
        fail(f"long distances it {it} L={L} C={Cn} Q={Q}", ex)
print(f"[{time.time() - t_start:.0f}s] long edit distances done", flush=True)

# (5) the explorer loop's pattern: train, then hundreds of small calls, on one engine
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(600, 8, "TGCA", 5))
y = np.array([s.count("G") / 8 for s in seqs], np.float32)
try:
    for rnd in range(200):
        ens.train(seqs[: 200 + 2 * rnd], y[: 200 + 2 * rnd], seed=rnd)
        ref = np.stack([ref_np.keras_fitness(seqs[:64], "TGCA", "cnn", m.model.get_weights(), exact=True) for m in members], axis=1).mean(axis=1)
        got = np.concatenate([ens.get_fitness(seqs[i:i + 8]) for i in range(0, 64, 8)])
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6 and np.isfinite(got).all()
    final_loss = float(np.mean((ens.get_fitness(seqs) - y) ** 2))
    assert final_loss < 0.01, final_loss
except Exception as ex:                                       # noqa: BLE001
    fail("explorer-loop pattern", ex)
print(f"[{time.time() - t_start:.0f}s] explorer-loop pattern done", flush=True)

# (6) the resident small-call form under an explorer's life: several model lists taking turns, bursts of small calls of random
#     size, new weights, training, big calls and pauses in between -- every answer against the launched form's
pool = {
    "3cnn8": [bm.CNN(8, 32, 100, "TGCA", seed=s) for s in range(3)],
    "mlp14": [bm.MLP(14, 100, "UGCA", seed=7)],
    "dyna": [bm.GlobalEpistasisModel(14, 100, "UGCA", seed=1), bm.MLP(14, 200, "UGCA", seed=2), bm.CNN(14, 32, 100, "UGCA", seed=3)],
    "8ge90": [bm.GlobalEpistasisModel(90, 100, s_utils.AAS, seed=s) for s in range(8)],
}
data = {k: synth.random_sequence_bytes(3000, v[0].model.L, v[0].alphabet, 11) for k, v in pool.items()}
want = {}


def refresh(key):
    eng.set_option("serve_small", 0)
    nat = [m.native() for m in pool[key]]
    want[key], _ = eng.score(nat, data[key], pool[key][0]._lut, want_matrix=True, want_mean=False)
    eng.set_option("serve_small", 1)


eng.set_option("poison_outputs", 0)
for k in pool:
    refresh(k)
served0 = eng.get_option("server_calls")
try:
    for rnd in range(1500):
        key = list(pool)[int(rng.integers(0, len(pool)))]
        members_k = pool[key]
        ens_k = flexs_amd.Ensemble(members_k, combine_with=lambda x: x)
        for _ in range(int(rng.integers(1, 30))):
            n = int(rng.integers(1, 257)); off = int(rng.integers(0, 3000 - n))
            got = ens_k.get_fitness(synth.bytes_to_strings(data[key][off:off + n]))
            if not np.array_equal(got, want[key][off:off + n]):
                raise AssertionError(f"round {rnd} {key} n={n} off={off}: {int((got != want[key][off:off + n]).sum())} scores differ")
        ev = rng.random()
        if ev < 0.15:                                         # new weights for one member
            m = members_k[int(rng.integers(0, len(members_k)))]
            m.model.set_weights([w * np.float32(1.0 + 0.01 * rng.standard_normal()) for w in m.model.get_weights()])
            refresh(key)
        elif ev < 0.22 and key == "3cnn8":                    # a training round
            flexs_amd.Ensemble(members_k).train(seqs[:300], y[:300], seed=rnd)
            refresh(key)
        elif ev < 0.30:                                       # a launch that fills the chip
            got = ens_k.get_fitness(synth.bytes_to_strings(data[key]))
            assert np.array_equal(got, want[key])
        elif ev < 0.40:
            time.sleep(float(rng.choice([0.0003, 0.0007, 0.003])))
    assert eng.get_option("server_fallbacks") <= 3, eng.get_option("server_last_fallback")   # (timing events; answers are checked above)
except Exception as ex:                                       # noqa: BLE001
    fail("resident small-call form", ex)
print(f"[{time.time() - t_start:.0f}s] resident form: {eng.get_option('server_calls') - served0} requests answered by resident workgroups, "
      f"{eng.get_option('server_starts')} generations, {eng.get_option('server_fallbacks')} fallbacks", flush=True)
print(f"[{time.time() - t_start:.0f}s] soak done, failures: {fails}", flush=True)
