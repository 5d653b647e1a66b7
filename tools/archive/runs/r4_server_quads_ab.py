"""Resident form, wide generation: three tiles per workgroup (serve_quads = 3) against one (serve_quads = 1), and where a
2001-sequence call goes: get_fitness(list[str]) / strpack alone / Engine.score on packed bytes (the C call)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
eng.set_option("serve_wide", 2)


def med_us(fn, reps=300):
    for _ in range(30):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


fams = [("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA"),
        ("1xCNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA"),
        ("GE+MLP+CNN L=8", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(8, 100, "TGCA", seed=1), bm.MLP(8, 100, "TGCA", seed=2), bm.CNN(8, 32, 100, "TGCA", seed=3)]), 8, "TGCA")]
for name, make, L, alpha in fams:
    model = make()
    pool = synth.bytes_to_strings(synth.random_sequence_bytes(4096, L, alpha, 3))
    ref = {}
    print(f"== {name}: median us per get_fitness(list[str]) call", flush=True)
    print("   N        quads=1   quads=3", flush=True)
    for n in (1, 20, 100, 256, 500, 1000, 2001, 4096):
        row = []
        for q in (1, 3):
            if q != 1 and not eng.get_option("ab_build"):
                row.append(float("nan"))                 # (three tiles per workgroup: A/B build only)
                continue
            eng.set_option("serve_quads", q)
            row.append(med_us(lambda: model.get_fitness(pool[:n]), 300 if n <= 1000 else 150))
            got = model.get_fitness(pool[:n])
            if q == 1: ref[n] = got
            else: assert np.array_equal(ref[n], got), (name, n)
        if eng.get_option("ab_build"): eng.set_option("serve_quads", 1)
        print(f"   {n:<6d} {row[0]:9.1f} {row[1]:9.1f}", flush=True)
    print(f"   server calls/starts/fallbacks: {eng.get_option('server_calls')} {eng.get_option('server_starts')} {eng.get_option('server_fallbacks')}", flush=True)

# breakdown of the 2001-sequence call of the 3 x CNN ensemble
model = fams[0][1]()
pool = synth.bytes_to_strings(synth.random_sequence_bytes(4096, 8, "TGCA", 3))
m0 = model.models[0]
natives = [m.native() for m in model.models]
for n in (257, 1000, 2001, 4096):
    batch = pool[:n]
    total = med_us(lambda: model.get_fitness(batch), 200)
    pack = med_us(lambda: _native.sequences_to_bytes(batch, L=8), 200)
    packed = _native.sequences_to_bytes(batch, L=8)
    call = med_us(lambda: m0._engine().score(natives, packed, m0._lut, want_matrix=False, want_mean=True), 200)
    print(f"3xCNN L=8 N={n}: get_fitness {total:.1f} us = pack {pack:.1f} + Engine.score(bytes) {call:.1f} + rest {total - pack - call:.1f}", flush=True)
    prof = np.median([[ (m0._engine().score(natives, packed, m0._lut, want_matrix=False, want_mean=True), [eng.get_option(f"server_prof_{k}") for k in range(8)])[1] ] for _ in range(100)], axis=0)[0]
    print("      inside the C call (ns since entry): checks %d, request posted %d, first answer %d, all collected %d, outputs written %d; member planes done at %d %d %d" % tuple(prof), flush=True)

# streamed (the strings packed straight into the mailbox, the request posted first) against packed-then-posted
from flexs_amd import _native as nat
for name, make, L, alpha in fams:
    rows = []
    for lo in (0, 384):
        nat.STREAM_MIN_ROWS = lo
        model = make()                                   # (a plan is built per model list: a fresh one picks the setting up)
        pool = synth.bytes_to_strings(synth.random_sequence_bytes(4096, L, alpha, 3))
        rows.append([med_us(lambda: model.get_fitness(pool[:n]), 200) for n in (384, 500, 1000, 2001, 4096)])
    print(f"{name}: get_fitness(list[str]) us at N = 384 / 500 / 1000 / 2001 / 4096", flush=True)
    print("   packed, then posted: " + "  ".join(f"{t:6.1f}" for t in rows[0]), flush=True)
    print("   streamed:            " + "  ".join(f"{t:6.1f}" for t in rows[1]), flush=True)
print("streamed calls answered:", eng.get_option("server_streamed"), " fallbacks:", eng.get_option("server_fallbacks"))
