"""Explorer-size calls: resident workgroups + mailbox (serve_small = 1) against a launch per call (serve_small = 0)."""
import sys, time; sys.path.insert(0, ".")
import random
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import rollouts
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"
for kind, L, alpha, M in (("cnn", 8, "TGCA", 3), ("cnn", 14, "UGCA", 3), ("cnn", 8, "TGCA", 1), ("mlp", 14, "UGCA", 1), ("mlp", 8, "TGCA", 3),
                          ("ge", 14, "UGCA", 3), ("mlp", 90, AAS, 1), ("dyna_ppo", 14, "UGCA", 3)):
    members = [bm.GlobalEpistasisModel(L, 100, alpha, seed=1), bm.MLP(L, 200, alpha, seed=2), bm.CNN(L, 32, 100, alpha, seed=3)] if kind == "dyna_ppo" else [bm.CNN(L, 32, 100, alpha, seed=m) if kind == "cnn" else bm.MLP(L, 100, alpha, seed=m) if kind == "mlp"
               else bm.GlobalEpistasisModel(L, 100, alpha, seed=m) for m in range(M)]
    ens = flexs_amd.Ensemble(members) if M > 1 else members[0]
    natives = [m.native() for m in members]
    for n in (1, 20, 32, 96):
        b = synth.random_sequence_bytes(n, L, alpha, 2)
        seqs = synth.bytes_to_strings(b)
        for rep in range(2):
            for serve in (1, 0):
                eng.set_option("serve_small", serve)
                for _ in range(300): ens.get_fitness(seqs)
                ts = []
                for _ in range(3000):
                    t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                for _ in range(300): eng.score(natives, b, members[0]._lut, want_matrix=False, want_mean=True)
                tr = []
                for _ in range(3000):
                    t0 = time.perf_counter(); eng.score(natives, b, members[0]._lut, want_matrix=False, want_mean=True); tr.append(time.perf_counter() - t0)
                print(f"{M}x{kind.upper()} L={L} N={n} serve_small={serve} [{rep}]: get_fitness(list[str]) {np.median(ts) * 1e6:.1f} us"
                      f"  (p90 {np.percentile(ts, 90) * 1e6:.1f}), raw fx_score {np.median(tr) * 1e6:.1f} us (p90 {np.percentile(tr, 90) * 1e6:.1f})", flush=True)
eng.set_option("serve_small", 1)
print("server calls / starts / fallbacks:", eng.get_option("server_calls"), eng.get_option("server_starts"), eng.get_option("server_fallbacks"))
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
for serve in (1, 0, 1, 0):
    eng.set_option("serve_small", serve)
    ts = []
    for i in range(5):
        random.seed(1)
        t0 = time.perf_counter()
        rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet="TGCA")
        ts.append(time.perf_counter() - t0)
    print(f"Adalead round (2000 queries) serve_small={serve}: {min(ts) * 1e3:.2f} ms", flush=True)
eng.set_option("serve_small", 1)
print("server calls / starts / fallbacks:", eng.get_option("server_calls"), eng.get_option("server_starts"), eng.get_option("server_fallbacks"))
