"""Timeline of a streamed get_fitness(list[str]) call (3 x CNN L=8, wide generation): ns since the call entered the library."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
eng.set_option("serve_wide", 2)
model = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
pool = synth.bytes_to_strings(synth.random_sequence_bytes(4096, 8, "TGCA", 3))
for n in (500, 1000, 2001, 4096):
    batch = pool[:n]
    for _ in range(30): model.get_fitness(batch)
    rows, walls = [], []
    for _ in range(200):
        t0 = time.perf_counter(); model.get_fitness(batch); walls.append((time.perf_counter() - t0) * 1e6)
        rows.append([eng.get_option(f"server_prof_{k}") for k in range(8)])
    p = np.median(np.array(rows), axis=0)
    print(f"N={n}: wall {np.median(walls):.1f} us; admitted {p[0]:.0f}, posted {p[1]:.0f}, first answer seen {p[2]:.0f}, planes done {p[5]:.0f} {p[6]:.0f} {p[7]:.0f}, "
          f"collected {p[3]:.0f}, outputs written {p[4]:.0f} ns; streamed so far {eng.get_option('server_streamed')}", flush=True)
