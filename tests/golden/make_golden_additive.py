#!/usr/bin/env python3
"""Generate tests/golden/additive_aav.json by RUNNING the reference's AdditiveAAVPackaging class
(flexs/landscapes/additive_aav_packaging.py) in the build container:

    python tests/golden/make_golden_additive.py

The measurement file the class opens (`data/additive_aav_packaging/AAV2_single_subs.json`) is not in the
reference checkout, so `open` is redirected, for that one path, to a synthetic file of the same schema
generated below; the fixture stores that synthetic input and the outputs the reference class produced.
"""
import builtins
import importlib
import io
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference  # noqa: E402

AAS = "ILVAGMFYWEDQNHCRKSTP"


def synthetic_single_subs(rng, lo, hi):
    data = {}
    for pos in range(lo, hi):
        present = [aa for aa in AAS if rng.random() < 0.8] or ["A"]
        if pos == lo + 3:
            present = ["K"]                                          # a position with a single measured residue
        subs = {}
        for aa in present:
            subs[aa] = {"log2_heart_v_wt": round(float(rng.normal(0, 1.5)), 4),
                        "log2_lung_v_wt": round(float(rng.normal(-0.5, 2.0)), 4),
                        "log2_packaging_v_wt": round(float(rng.normal(-2, 3)), 4)}
        if pos == lo + 5:                                            # nothing packages here: the "M" / -10 default
            for aa in subs:
                subs[aa]["log2_packaging_v_wt"] = -7.5
        data[str(pos)] = subs
    return data


def main():
    import_reference()
    rng = np.random.default_rng(20260928)
    lo, hi = 450, 482
    data = synthetic_single_subs(rng, lo, hi)
    blob = json.dumps(data)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if str(path).endswith("AAV2_single_subs.json"):
            return io.StringIO(blob)
        return real_open(path, *a, **k)

    mod = importlib.import_module("flexs.landscapes.additive_aav_packaging")
    cases = []
    for phen, mfm, start, end, noise, seed in (("heart", 1, 450, 482, 0, 1), ("lung", 1, 455, 470, 0, 2),
                                               ("heart", 0.5, 450, 482, 0.05, 3), ("lung", 2, 460, 482, 0.3, 4),
                                               ("heart", 1, 450, 482, 5, 5)):
        builtins.open = fake_open
        try:
            land = mod.AdditiveAAVPackaging(phenotype=phen, minimum_fitness_multiplier=mfm, start=start, end=end, noise=noise)
        finally:
            builtins.open = real_open
        L = end - start
        seqs = [land.wild_type, land.top_seq]
        for _ in range(40):
            s = list(land.wild_type if rng.random() < 0.5 else land.top_seq)
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, L))] = AAS[int(rng.integers(0, 20))]
            seqs.append("".join(s))
        seqs.append("X" * L)                                         # residues no position has data for
        seqs.append(land.wild_type[: L // 2])                        # shorter than the window: fewer terms
        np.random.seed(seed)
        out1 = land.get_fitness(seqs)
        out2 = land.get_fitness(seqs[:7])
        cases.append({"params": {"phenotype": phen, "minimum_fitness_multiplier": mfm, "start": start, "end": end,
                                 "noise": noise}, "seed": seed, "sequences": seqs,
                      "fitness": [float(x) for x in out1], "fitness_second_call": [float(x) for x in out2],
                      "dtype": str(out1.dtype), "cost": int(land.cost), "name": land.name, "top_seq": land.top_seq,
                      "max_possible": float(land.max_possible), "wild_type": land.wild_type,
                      "rng_next_random": float(np.random.random())})
    too_long_error = None
    builtins.open = fake_open
    try:
        land = mod.AdditiveAAVPackaging(start=450, end=460)
    finally:
        builtins.open = real_open
    try:
        land.get_fitness(["A" * 11])
    except KeyError as e:
        too_long_error = e.args[0]
    json.dump({"single_subs": data, "cases": cases, "too_long_keyerror": too_long_error,
               "registry": mod.registry()}, open(os.path.join(OUT, "additive_aav.json"), "w"), indent=0)
    print("wrote additive_aav.json", os.path.getsize(os.path.join(OUT, "additive_aav.json")), "bytes;",
          [(c["dtype"], round(c["max_possible"], 3)) for c in cases], too_long_error)


if __name__ == "__main__":
    main()
