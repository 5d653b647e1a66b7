"""Kernel trace target: explorer-size protein calls (layer-parallel form) and a wide resident request stream."""
import sys; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import synth
from flexs_amd.baselines import models as bm
AAS = "ILVAGMFYWEDQNHCRKSTP"
ens = flexs_amd.Ensemble([bm.CNN(237, 32, 100, AAS, seed=m) for m in range(3)])
for n in (1, 40):
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, 237, AAS, 12))
    for _ in range(200):
        ens.get_fitness(seqs)
