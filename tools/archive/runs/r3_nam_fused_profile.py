import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines.models import NoisyAbstractModel
exec(open("tools/archive/runs/r3_nam_fused_ab.py").read().split("def run(")[0].split("class Batched")[0].split('import sys, time; sys.path.insert(0, ".")')[1])
np.random.seed(0)
model = NoisyAbstractModel(Table(), 0.9)
model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, L, "ACGT", 5)), np.random.random(1000))
batches = [synth.bytes_to_strings(synth.random_sequence_bytes(1, L, "ACGT", 100 + c)) for c in range(2300)]
for b in batches[:300]: model.get_fitness(b)
pr = cProfile.Profile(); pr.enable()
for b in batches[300:]: model.get_fitness(b)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
