"""Relay launches on ALTERNATING batches of one shape: bytes of the previous call left in a cache of another XCD would show as the
other batch's scores.  Also alternating sizes (flags of a longer earlier call) and the launched-first calls without a relay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import flexs_amd
from flexs_amd import synth, _native
from flexs_amd.baselines import models as bm
from flexs_amd.utils import sequence_utils as s_utils
eng = _native.Engine.get()
AAS = s_utils.AAS
bad = 0
for tag, make, M, L, alpha in (("8xGE L=90", lambda s: bm.GlobalEpistasisModel(90, 100, AAS, seed=s), 8, 90, AAS),
                               ("3xCNN L=8", lambda s: bm.CNN(8, 32, 100, "TGCA", seed=s), 3, 8, "TGCA"),
                               ("MLP L=14", lambda s: bm.MLP(14, 100, "UGCA", seed=s), 1, 14, "UGCA")):
    mods = [make(s) for s in range(M)]
    model = mods[0] if M == 1 else flexs_amd.Ensemble(mods)
    batches = []
    for k, n in enumerate((100_000, 100_000, 60_001, 100_000, 33_333)):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1000 + k))
        eng.set_option("launch_first", 0); eng.set_option("launch_relay", 0)
        want = np.asarray(model.get_fitness(seqs)).copy()
        batches.append((seqs, want))
    eng.set_option("launch_first", 1); eng.set_option("launch_relay", 1)
    c0 = (eng.get_option("launch_first_calls"), eng.get_option("launch_relay_calls"), eng.get_option("launch_first_redone"))
    calls = 0
    for rep in range(12):
        for seqs, want in batches:
            got = np.asarray(model.get_fitness(seqs))
            calls += 1
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                print(tag, "rep", rep, "n", len(seqs), "differs in", int((got != want).sum()), "scores", flush=True)
    c1 = (eng.get_option("launch_first_calls"), eng.get_option("launch_relay_calls"), eng.get_option("launch_first_redone"))
    print(f"{tag}: {calls} alternating calls, launched first {c1[0] - c0[0]}, relayed {c1[1] - c0[1]}, redone {c1[2] - c0[2]}, wrong {bad}", flush=True)
assert bad == 0
print("OK")
