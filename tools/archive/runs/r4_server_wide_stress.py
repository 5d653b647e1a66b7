"""Wide resident form under stress, WITHOUT the per-tile fence: random request sizes 1 ... capacity, every answer compared with
the launched form's; a stale or missing answer would show as a wrong value or a fallback."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
rng = np.random.default_rng(0)
AAS = "ILVAGMFYWEDQNHCRKSTP"
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
for kind, L, alpha, M in (("cnn", 8, "TGCA", 3), ("cnn", 8, "TGCA", 1), ("cnn", 14, "UGCA", 16), ("mlp", 14, "UGCA", 3),
                          ("ge", 14, "UGCA", 8), ("mlp", 90, AAS, 2), ("ge", 90, AAS, 8)):
    members = [bm.CNN(L, 32, 100, alpha, seed=m) if kind == "cnn" else bm.MLP(L, 100, alpha, seed=m) if kind == "mlp"
               else bm.GlobalEpistasisModel(L, 100, alpha, seed=m) for m in range(M)]
    natives = [m.native() for m in members]
    lut = members[0]._lut
    cap = min(4096, 65536 // L)
    pool = synth.random_sequence_bytes(8192, L, alpha, 5)
    eng.set_option("serve_small", 0)
    want_all, _ = eng.score(natives, pool, lut, want_matrix=True, want_mean=False)
    eng.set_option("serve_small", 1)
    c0, f0 = eng.get_option("server_calls"), eng.get_option("server_fallbacks")
    bad = 0
    t0 = time.time()
    for it in range(ITERS):
        # mostly explorer-size, a tail of mid-size requests up to the capacity
        n = int(rng.integers(1, 161)) if rng.random() < 0.6 else int(rng.integers(161, cap + 1))
        off = int(rng.integers(0, 8192 - n))
        got, _ = eng.score(natives, pool[off:off + n], lut, want_matrix=True, want_mean=False)
        if not np.array_equal(got, want_all[off:off + n]):
            bad += 1
            idx = np.argwhere(got != want_all[off:off + n])
            if bad <= 8:
                print(f"  it {it} n={n}: {len(idx)} wrong, rows {sorted(set(idx[:, 0].tolist()))[:12]} members {sorted(set(idx[:, 1].tolist()))}", flush=True)
        if it % 3000 == 2999 and rng.random() < 0.5:
            time.sleep(0.01)
    print(f"{M}x{kind.upper()} L={L} (capacity {cap}): {bad} wrong answers of {ITERS} in {time.time() - t0:.1f} s; served "
          f"{eng.get_option('server_calls') - c0}, fallbacks {eng.get_option('server_fallbacks') - f0}, starts {eng.get_option('server_starts')}", flush=True)
