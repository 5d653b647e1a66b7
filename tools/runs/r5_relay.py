"""Launched-first calls of dense ENSEMBLES whose plan says "copy" (every member reading the rows over PCIe again would not hide): member
0's workgroups relay the rows through device memory (launch_relay = 1) against pack -> upload -> launch (0).  Same bits, 25 repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
def med(f, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
AAS = s_utils.AAS
cases = [("8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS, 100_000),
         ("3xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 3, 90, AAS, 100_003),
         ("8xGE L=237", lambda s: bm.GlobalEpistasisModel(237, 100, AAS, seed=s), 8, 237, AAS, 40_000),
         ("8xMLP L=50", lambda s: bm.MLP(50, 100, "UGCA", seed=s), 8, 50, "UGCA", 100_000),
         ("8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS, 250_000)]
for tag, make, M, L, alpha, n in cases:
    ens = flexs_amd.Ensemble([make(s) for s in range(M)])
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, n))
    eng.set_option("launch_relay", 0)
    want = ens.get_fitness(seqs).copy()
    t_off = med(lambda: ens.get_fitness(seqs))
    eng.set_option("launch_relay", 1)
    c0, r0 = eng.get_option("launch_relay_calls"), eng.get_option("launch_first_redone")
    bad = 0
    for _ in range(25):
        got = ens.get_fitness(seqs)
        bad += int(not np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    took = eng.get_option("launch_relay_calls") - c0
    t_on = med(lambda: ens.get_fitness(seqs))
    # the plain host call on packed bytes (what an ndarray of bytes, or a C caller, gets): no upload in front of the launch
    b = np.asarray(synth.random_sequence_bytes(n, L, alpha, n)).reshape(n, L)
    nat = [m.native() for m in ens.models]
    eng.set_option("launch_relay", 0)
    want_b = eng.score(nat, b, ens.models[0]._lut, want_matrix=False, want_mean=True)[1].copy()
    tb_off = med(lambda: eng.score(nat, b, ens.models[0]._lut, want_matrix=False, want_mean=True))
    eng.set_option("launch_relay", 1)
    cb = eng.get_option("launch_relay_calls")
    for _ in range(10):
        got_b = eng.score(nat, b, ens.models[0]._lut, want_matrix=False, want_mean=True)[1]
        bad += int(not np.array_equal(got_b.view(np.uint32), want_b.view(np.uint32)))
    nm_b = eng.score(nat, b, ens.models[0]._lut, want_matrix=True, want_mean=True)
    eng.set_option("launch_relay", 0)
    nm_w = eng.score(nat, b, ens.models[0]._lut, want_matrix=True, want_mean=True)
    eng.set_option("launch_relay", 1)
    bad += int(not (np.array_equal(nm_b[0], nm_w[0]) and np.array_equal(nm_b[1], nm_w[1])))
    took_b = eng.get_option("launch_relay_calls") - cb
    tb_on = med(lambda: eng.score(nat, b, ens.models[0]._lut, want_matrix=False, want_mean=True))
    print(f"{tag} n={n}: fx_score on packed bytes: relayed {took_b} of 11 calls; {tb_off:.0f} us without, {tb_on:.0f} us with the relay", flush=True)
    print(f"{tag} n={n}: relayed {took} of 25 calls, redone {eng.get_option('launch_first_redone') - r0}, calls with other bits {bad}; "
          f"{t_off:.0f} us without, {t_on:.0f} us with the relay", flush=True)
    assert bad == 0
print("OK")
