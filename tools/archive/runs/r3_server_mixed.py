"""Mixed ensembles on the resident form: hunt for fallbacks."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
L, alpha = 14, "UGCA"
lists = {
    "dyna_ppo": [bm.GlobalEpistasisModel(L, 100, alpha, seed=1), bm.MLP(L, 200, alpha, seed=2), bm.CNN(L, 32, 100, alpha, seed=3)],
    "cnn_mlp_cnn_cnn": [bm.CNN(L, 32, 100, alpha, seed=4), bm.MLP(L, 100, alpha, seed=5), bm.CNN(L, 32, 100, alpha, seed=6), bm.CNN(L, 32, 100, alpha, seed=7)],
    "two_mlp_sizes": [bm.MLP(L, 100, alpha, seed=8), bm.MLP(L, 50, alpha, seed=9), bm.MLP(L, 50, alpha, seed=10)],
}
rng = np.random.default_rng(1)
pool = synth.random_sequence_bytes(4096, L, alpha, 5)
for rep in range(3):
    for name, members in lists.items():
        natives = [m.native() for m in members]
        lut = members[0]._lut
        eng.set_option("serve_small", 0)
        want_all, _ = eng.score(natives, pool, lut, want_matrix=True, want_mean=False)
        eng.set_option("serve_small", 1)
        f0, s0, c0 = eng.get_option("server_fallbacks"), eng.get_option("server_starts"), eng.get_option("server_calls")
        wrong = 0
        for it in range(20000):
            n = int(rng.integers(1, 257)); off = int(rng.integers(0, 4096 - n))
            t0 = time.perf_counter()
            got, _ = eng.score(natives, pool[off:off + n], lut, want_matrix=True, want_mean=False)
            if it < 3 and rep == 0: print(f"  {name} call {it}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
            wrong += not np.array_equal(got, want_all[off:off + n])
            f = eng.get_option("server_fallbacks")
            if f != f0:
                print(f"  {name} it {it} n={n}: fallback, info {eng.get_option('server_last_fallback')}", flush=True)
                f0 = f
        print(f"{name} [{rep}]: wrong {wrong}, starts {eng.get_option('server_starts') - s0}, served {eng.get_option('server_calls') - c0}", flush=True)
