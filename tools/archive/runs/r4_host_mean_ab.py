"""Launched explorer-size mean-only calls (protein CNN): the mean on the host from member planes in pinned memory
(host_mean_below = 256, default) against the mean kernel (0), the completion flag (done_flag = 1, default: the kernel's last unit
raises a word in pinned host memory that the host polls) against hipStreamSynchronize, and the call's timeline inside the library."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils import population
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"


def med_us(fn, reps=200):
    for _ in range(20): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


for L, M in ((237, 3), (90, 3), (237, 8)):
    members = [bm.CNN(L, 32, 100, AAS, seed=m) for m in range(M)]
    ens = flexs_amd.Ensemble(members)
    ev = population.PopulationEvaluator(ens, AAS, L)
    rng = np.random.default_rng(0)
    out = {}
    for below in (0, 256, 257):
        eng.set_option("host_mean_below", min(below, 256))
        eng.set_option("done_flag", 0 if below < 257 else 1)
        row = {}
        for n in (1, 16, 40):
            seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, 12))
            row[f"N={n}"] = med_us(lambda: ens.get_fitness(seqs))
            ref = ens.get_fitness(seqs)
            if below == 0: out.setdefault("ref", {})[n] = ref
            else: assert np.array_equal(out["ref"][n], ref), (L, M, n, below)
        for P in (15, 40):
            x = rng.standard_normal((P, L * 20))
            row[f"P={P} step"] = med_us(lambda: ev.evaluate(x), 100)
        if below:
            seqs = synth.bytes_to_strings(synth.random_sequence_bytes(1, L, AAS, 12))
            prof = np.median([[(ens.get_fitness(seqs), [eng.get_option(f"call_prof_{k}") for k in range(4)])[1]] for _ in range(100)], axis=0)[0]
            row["timeline N=1 (ns since the packed call entered: prepared, launched, synchronised, mean taken)"] = [int(v) for v in prof]
        out[below] = row
    print(f"== {M} x CNN(32,100) L={L} A=20: us, mean kernel / host mean / host mean + completion flag", flush=True)
    for k in out[0]:
        print(f"   {k:14s} {out[0][k]:8.1f} / {out[256][k]:8.1f} / {out[257][k]:8.1f}", flush=True)
    tl = "timeline N=1 (ns since the packed call entered: prepared, launched, synchronised, mean taken)"
    print("   " + tl, out[256][tl], "with the flag:", out[257][tl], flush=True)
eng.set_option("host_mean_below", 256)
eng.set_option("done_flag", 1)
