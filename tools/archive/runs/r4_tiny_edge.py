"""Tiny requests (FxMailIn::tiny) at the edges the GPU suite's L >= 8 cases do not reach: sequences of 1-2 symbols, where 48 bytes
are more than one tile's 16 sequences (those requests take the byte area) and where a tile's byte rows are shorter than the 48-byte
line.  Resident answers against the launched call, serve_tiny on and off.  Prints one line per family; exit code 1 on a difference."""
import sys; sys.path.insert(0, ".")
import numpy as np, flexs_amd
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm

eng = _native.Engine.get()
bad = 0
for kind, L, M in (("mlp", 2, 1), ("ge", 1, 2), ("ge", 2, 3), ("mlp", 1, 2), ("mlp", 3, 1), ("cnn", 8, 3)):
    alpha = "UGCA"
    mk = {"cnn": lambda s: bm.CNN(L, 32, 100, alpha, seed=s), "mlp": lambda s: bm.MLP(L, 100, alpha, seed=s),
          "ge": lambda s: bm.GlobalEpistasisModel(L, 100, alpha, seed=s)}[kind]
    try:
        members = [mk(70 + s) for s in range(M)]
        ens = flexs_amd.Ensemble(members) if M > 1 else members[0]
        sizes = [1, 2, 3, 15, 16, 17, 23, 24, 25, 40, 47, 48, 49]
        data = {n: synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 900 + n)) for n in sizes}
        eng.set_option("serve_small", 0)
        want = {n: ens.get_fitness(data[n]) for n in sizes}
        eng.set_option("serve_small", 1)
        wrong, resident = [], 0
        for tiny in (1, 0, 1):
            eng.set_option("serve_tiny", tiny)
            for _ in range(12):
                ens.get_fitness(data[1])
                if eng.get_option("server_resident") == 1:
                    break
            resident += eng.get_option("server_resident")
            for rep in range(3):
                for n in sizes + sizes[::-1]:
                    if not np.array_equal(ens.get_fitness(data[n]), want[n]):
                        wrong.append((tiny, rep, n))
        print(f"{kind} L={L} M={M}: resident in {resident}/3 legs, wrong {wrong[:8]}", flush=True)
        bad += len(wrong)
    except Exception as ex:                                  # (a shape one of the paths refuses: named, not fatal for the others)
        print(f"{kind} L={L} M={M}: {type(ex).__name__}: {ex}", flush=True)
        bad += 1
    finally:
        eng.set_option("serve_small", 1); eng.set_option("serve_tiny", 1)
sys.exit(1 if bad else 0)
