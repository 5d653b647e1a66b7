"""Resident small-call form under stress: random request sizes, every answer compared with the launched form's."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
rng = np.random.default_rng(0)
AAS = "ILVAGMFYWEDQNHCRKSTP"
for kind, L, alpha, M in (("cnn", 7, "TGCA", 8), ("cnn", 8, "TGCA", 3), ("cnn", 14, "UGCA", 16), ("cnn", 14, "UGCA", 3), ("mlp", 14, "UGCA", 3),
                          ("ge", 14, "UGCA", 8), ("mlp", 90, AAS, 2), ("ge", 90, AAS, 1)):
    members = [bm.CNN(L, 32, 100, alpha, seed=m) if kind == "cnn" else bm.MLP(L, 100, alpha, seed=m) if kind == "mlp"
               else bm.GlobalEpistasisModel(L, 100, alpha, seed=m) for m in range(M)]
    natives = [m.native() for m in members]
    lut = members[0]._lut
    pool = synth.random_sequence_bytes(4096, L, alpha, 5)
    eng.set_option("serve_small", 0)
    want_all, _ = eng.score(natives, pool, lut, want_matrix=True, want_mean=False)
    eng.set_option("serve_small", 1)
    bad = 0
    prev = None
    t0 = time.time()
    for it in range(40000):
        n = int(rng.integers(1, 161))
        off = int(rng.integers(0, 4096 - n))
        got, _ = eng.score(natives, pool[off:off + n], lut, want_matrix=True, want_mean=False)
        want = want_all[off:off + n]
        if not np.array_equal(got, want):
            bad += 1
            idx = np.argwhere(got != want)
            stale = prev is not None and all(r < prev[1].shape[0] and got[r, c] == prev[1][r, c] for r, c in idx)
            if bad <= 8:
                print(f"  it {it} n={n}: {len(idx)} wrong, rows {sorted(set(idx[:, 0].tolist()))[:20]} members {sorted(set(idx[:, 1].tolist()))}"
                      f" equal to the previous answer in those slots: {stale}; previous n={prev[0] if prev else None}", flush=True)
        prev = (n, got)
        if it % 5000 == 4999 and rng.random() < 0.5: time.sleep(0.01)
    print(f"{M}x{kind.upper()} L={L}: {bad} wrong answers of 40000 in {time.time() - t0:.1f} s; server calls/starts/fallbacks",
          eng.get_option("server_calls"), eng.get_option("server_starts"), eng.get_option("server_fallbacks"), flush=True)
