"""Wide resident form (round 4) A/B: get_fitness(list[str]) latency by batch size under
  wide      serve_wide = 1, no fence after the answers (the default)
  wide+f    serve_wide = 1, round 3's system fence per tile
  r3        serve_wide = 0, fence per tile: round 3's geometry (<= 256 sequences served, 48 workgroups for 3 members)
  launch    serve_small = 0: a launch per call
and one Adalead round (2000 model queries) under wide / r3 / launch."""
import random, sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import rollouts
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"
MODES = {"wide": dict(serve_small=1, serve_wide=2, serve_fence=0), "adaptive": dict(serve_small=1, serve_wide=1, serve_fence=0),
         "r3": dict(serve_small=1, serve_wide=0, serve_fence=1), "launch": dict(serve_small=0, serve_wide=1, serve_fence=0)}


def set_mode(name):
    for k, v in MODES[name].items():
        eng.set_option(k, v)


def call_us(model, batch, reps=300):
    for _ in range(30):
        model.get_fitness(batch)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); model.get_fitness(batch); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, float(np.percentile(ts, 99)) * 1e6


fams = [("3xCNN L=8", lambda: flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)]), 8, "TGCA"),
        ("1xCNN L=8", lambda: bm.CNN(8, 32, 100, "TGCA", seed=0), 8, "TGCA"),
        ("3xCNN L=14", lambda: flexs_amd.Ensemble([bm.CNN(14, 32, 100, "UGCA", seed=m) for m in range(3)]), 14, "UGCA"),
        ("MLP L=14", lambda: bm.MLP(14, 100, "UGCA", seed=0), 14, "UGCA"),
        ("3xGE L=14", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(14, 100, "UGCA", seed=m) for m in range(3)]), 14, "UGCA"),
        ("8xGE L=90", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(90, 100, AAS, seed=m) for m in range(8)]), 90, AAS),
        ("GE+MLP200+CNN L=14", lambda: flexs_amd.Ensemble([bm.GlobalEpistasisModel(14, 100, "UGCA", seed=1), bm.MLP(14, 200, "UGCA", seed=2), bm.CNN(14, 32, 100, "UGCA", seed=3)]), 14, "UGCA")]
for name, make, L, alpha in fams:
    model = make()
    pool = synth.bytes_to_strings(synth.random_sequence_bytes(4096, L, alpha, 3))
    sizes = [n for n in (1, 20, 100, 256, 257, 500, 1000, 2001, 4096) if n * L <= 65536]
    print(f"== {name}: median (p99) us per get_fitness(list[str]) call", flush=True)
    print("   N      " + "".join(f"{m:>18s}" for m in MODES), flush=True)
    for n in sizes:
        row = []
        for mode in MODES:
            set_mode(mode)
            if mode == "adaptive":
                time.sleep(0.3)                                  # (the adaptive geometry remembers mid-size requests for 0.25 s)
            med, p99 = call_us(model, pool[:n], reps=300 if n <= 1000 else 150)
            row.append(f"{med:9.1f} ({p99:6.1f})")
        print(f"   {n:<6d}" + "".join(f"{r:>18s}" for r in row), flush=True)
    set_mode("wide")
    print(f"   server calls/starts/fallbacks: {eng.get_option('server_calls')} {eng.get_option('server_starts')} {eng.get_option('server_fallbacks')}", flush=True)
# one Adalead round on trained-looking members (synthetic weights): 2000 queries
ens = flexs_amd.Ensemble([bm.CNN(8, 32, 100, "TGCA", seed=m) for m in range(3)])
seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "TGCA", 3))
y = np.random.default_rng(0).random(1000)
for mode in ("adaptive", "wide", "r3", "launch"):
    set_mode(mode)
    time.sleep(0.3)
    ts = []
    for i in range(5):
        random.seed(1)
        c0 = ens.cost
        t0 = time.perf_counter()
        rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet="TGCA")
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"Adalead round, {mode}: {np.median(ts[1:]):.2f} ms (runs {[round(t, 2) for t in ts]}), {ens.cost - c0} queries", flush=True)
set_mode("adaptive")
