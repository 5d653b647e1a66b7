"""Round-2 probe: wall time of `model.train` (20 epochs, batch 256) per explorer round, next to the scoring calls of that round."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from flexs_amd import synth
from flexs_amd.baselines import models as bm

for kind, L, alpha in (("cnn", 8, "TGCA"), ("mlp", 14, "UGCA"), ("ge", 90, "ILVAGMFYWEDQNHCRKSTP"), ("cnn", 90, "ILVAGMFYWEDQNHCRKSTP")):
    for n in (100, 1000, 5000):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 1))
        y = np.random.default_rng(0).random(n)
        m = {"cnn": lambda: bm.CNN(L, 32, 100, alpha), "mlp": lambda: bm.MLP(L, 100, alpha), "ge": lambda: bm.GlobalEpistasisModel(L, 100, alpha)}[kind]()
        m.train(seqs, y); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.train(seqs, y); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = 20 * ((n + 255) // 256)
        t1 = time.perf_counter(); m.get_fitness(seqs); t2 = time.perf_counter() - t1
        print({"what": f"train {kind} L={L} n={n}", "train_ms": round(dt * 1e3, 1), "steps": steps, "ms_per_step": round(dt * 1e3 / steps, 3), "get_fitness_ms": round(t2 * 1e3, 3)}, flush=True)
