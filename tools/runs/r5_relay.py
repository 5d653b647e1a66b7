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
    print(f"{tag} n={n}: relayed {took} of 25 calls, redone {eng.get_option('launch_first_redone') - r0}, calls with other bits {bad}; "
          f"{t_off:.0f} us without, {t_on:.0f} us with the relay", flush=True)
    assert bad == 0
print("OK")
