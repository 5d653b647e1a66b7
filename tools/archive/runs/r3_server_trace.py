"""One Adalead round (2000 model queries on the 3 x CNN L=8 ensemble) with the resident form, then with a launch per
call -- run under `rocprofv3 --kernel-trace --stats`: the first shows a handful of resident kernels, the second ~2 launches
per call."""
import sys, time, random; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import rollouts
eng = _native.Engine.get()
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3)); y = np.random.default_rng(0).random(1000)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng.set_option("serve_small", mode)
for i in range(3):
    random.seed(1)
    c0 = ens.cost
    t0 = time.perf_counter()
    rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet="TGCA")
    print(f"serve_small={mode}: Adalead round {1e3 * (time.perf_counter() - t0):.2f} ms, {int(ens.cost - c0)} model queries,"
          f" resident requests so far {eng.get_option('server_calls')}, generations {eng.get_option('server_starts')}", flush=True)
