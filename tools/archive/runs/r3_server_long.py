"""Six seconds of back-to-back explorer-size calls: the host replaces a resident generation after 4 s (the workgroups' own
lifetime limit is 10 s); every answer is checked."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
members = [bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]
ens = flexs_amd.Ensemble(members)
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(20, 8, "TGCA", 2))
eng.set_option("serve_small", 0); want = ens.get_fitness(seqs); eng.set_option("serve_small", 1)
t0 = time.time(); n = wrong = 0; worst = 0.0
while time.time() - t0 < 6.5:
    t1 = time.perf_counter(); got = ens.get_fitness(seqs); dt = time.perf_counter() - t1
    worst = max(worst, dt); n += 1; wrong += not np.array_equal(got, want)
print(f"{n} calls in 6.5 s ({6.5e6 / n:.1f} us per call), wrong {wrong}, slowest call {worst * 1e6:.0f} us, generations {eng.get_option('server_starts')}, "
      f"served {eng.get_option('server_calls')}, fallbacks {eng.get_option('server_fallbacks')}")
