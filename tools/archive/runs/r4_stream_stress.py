"""Streamed requests under stress: random request sizes 1 ... capacity through Ensemble.get_fitness(list[str]) (requests of >= 384
strings are streamed: posted first, strings packed straight into the mailbox), every answer compared with the launched form's bits;
every 700th call carries a string that cannot be packed (found mid-stream) and must raise."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
rng = np.random.default_rng(1)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
fams = [("3xCNN L=8", lambda: [bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)], 8, "TGCA"),
        ("1xMLP L=14", lambda: [bm.MLP(14, 100, "UGCA", seed=0)], 14, "UGCA"),
        ("GE+MLP200+CNN L=14", lambda: [bm.GlobalEpistasisModel(14, 100, "UGCA", seed=1), bm.MLP(14, 200, "UGCA", seed=2), bm.CNN(14, 32, 100, "UGCA", seed=3)], 14, "UGCA"),
        ("8xGE L=90", lambda: [bm.GlobalEpistasisModel(90, 100, "ILVAGMFYWEDQNHCRKSTP", seed=m) for m in range(8)], 90, "ILVAGMFYWEDQNHCRKSTP")]
for name, make, L, alpha in fams:
    members = make()
    ens = flexs_amd.Ensemble(members) if len(members) > 1 else members[0]
    cap = min(4096, 65536 // L)
    pool = synth.bytes_to_strings(synth.random_sequence_bytes(8192, L, alpha, 5))
    eng.set_option("serve_small", 0)
    want_all = ens.get_fitness(pool)
    eng.set_option("serve_small", 1)
    c0, f0, s0 = eng.get_option("server_calls"), eng.get_option("server_fallbacks"), eng.get_option("server_streamed")
    bad = raised = 0
    t0 = time.time()
    for it in range(ITERS):
        n = int(rng.integers(1, 161)) if rng.random() < 0.4 else int(rng.integers(161, cap + 1))
        off = int(rng.integers(0, 8192 - n))
        batch = pool[off:off + n]
        if it % 700 == 699 and n >= 400:
            broken = list(batch); broken[int(rng.integers(300, n))] = 7
            try:
                ens.get_fitness(broken)
            except TypeError:
                raised += 1
            continue
        got = ens.get_fitness(batch)
        if not np.array_equal(got, want_all[off:off + n]):
            bad += 1
            if bad <= 5:
                print(f"  it {it} n={n}: {int((got != want_all[off:off + n]).sum())} wrong", flush=True)
    print(f"{name} (capacity {cap}): {bad} wrong answers of {ITERS} in {time.time() - t0:.1f} s; served {eng.get_option('server_calls') - c0}, "
          f"streamed {eng.get_option('server_streamed') - s0}, fallbacks {eng.get_option('server_fallbacks') - f0}, mid-stream TypeErrors raised {raised}", flush=True)
