"""Relay launches: member 0's workgroups (the only ones that pull rows over PCIe) on ONE XCD (relay_spread = 0: the XCD-aware unit
ranges of every other launch) or on all eight (1: plain block order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
def med(f, n=21):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
AAS = s_utils.AAS
for tag, M, L, n in (("8xGE L=90", 8, 90, 100_000), ("3xGE L=90", 3, 90, 100_003), ("8xGE L=237", 8, 237, 40_000), ("8xGE L=90", 8, 90, 250_000)):
    ens = flexs_amd.Ensemble([bm.GlobalEpistasisModel(L, 100, AAS, seed=s) for s in range(M)])
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, n))
    res = {}
    for spread in (0, 1, 0, 1):
        eng.set_option("relay_spread", spread)
        got = ens.get_fitness(seqs).copy()
        res.setdefault(spread, []).append((got, med(lambda: ens.get_fitness(seqs))))
    same = all(np.array_equal(res[0][0][0].view(np.uint32), g.view(np.uint32)) for v in res.values() for g, _ in v)
    print(f"{tag} n={n}: same bits {same}; one XCD {res[0][0][1]:.0f} / {res[0][1][1]:.0f} us, all XCDs {res[1][0][1]:.0f} / {res[1][1][1]:.0f} us", flush=True)
    assert same
eng.set_option("relay_spread", 0)
