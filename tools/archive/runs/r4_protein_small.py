"""Small protein batches (configs[4]'s explorer pattern): layer-parallel form (cnn_lp = 1, round 4) against the position-segmented
form (cnn_lp = 0, rounds 1-2): launch time from C, get_fitness latency, the CMA-ES population step."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
from flexs_amd.utils.population import PopulationEvaluator
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"

def med_us(fn, reps=100):
    for _ in range(10): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6

for L in (237, 90):
    for M in (1, 3):
        members = [bm.CNN(L, 32, 100, AAS, seed=m) for m in range(M)]
        model = flexs_amd.Ensemble(members) if M > 1 else members[0]
        ev = PopulationEvaluator(model, AAS, L)
        rng = np.random.default_rng(5)
        rows = []
        for lp in (1, 0):
            eng.set_option("cnn_lp", lp)
            r = {}
            for n in (1, 16, 40):
                seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, AAS, 12))
                r[f"N={n}"] = med_us(lambda: model.get_fitness(seqs))
                d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 12)).cuda()
                stride = 64
                d_pl = torch.empty((M, stride), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                nat = [m.native() for m in members]
                ms = eng.time_score_planes(nat, d_in.data_ptr(), n, L, members[0]._lut, d_pl.data_ptr(), stride, 200)
                r[f"kernel N={n}"] = ms / 200 * 1e3
            for P in (15, 40):
                x = rng.standard_normal((P, L * 20))
                from flexs_amd.utils import population
                for host in (True, False):
                    population.HOST_DECODE = host
                    r[f"P={P} decode+score" + (" (host argmax)" if host else " (device argmax)")] = med_us(lambda: ev.evaluate(x))
                population.HOST_DECODE = True
                xo = np.ascontiguousarray(x.reshape(P, L, 20)); out = np.empty((P, L), np.uint8)
                r[f"P={P} host argmax alone"] = med_us(lambda: _native._strpack.decode_argmax(xo, P * L, 20, AAS.encode(), out))
            rows.append(r)
        eng.set_option("cnn_lp", 1)
        print(f"== {M} x CNN(32,100) L={L} A=20: us, layer-parallel / segmented", flush=True)
        for k in rows[0]:
            print(f"   {k:24s} {rows[0][k]:8.1f} / {rows[1][k]:8.1f}", flush=True)
