#!/usr/bin/env python3
"""Generate tests/golden/sequence_generators.json by RUNNING the reference's sequence_utils helpers
(flexs/utils/sequence_utils.py:18-29, 69-108) under fixed `random` seeds:

    python tests/golden/make_golden_sequtils.py
"""
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, import_reference  # noqa: E402


def main():
    _, s_utils, *_ = import_reference()
    cases = {"single_mutants": [], "random_sequences": [], "random_mutant": [], "construct_mutant": []}
    for wt, alpha in (("AT", s_utils.DNAA), ("UGCAU", s_utils.RNAA), ("MKV", s_utils.AAS), ("", s_utils.DNAA)):
        cases["single_mutants"].append({"wt": wt, "alphabet": alpha, "out": s_utils.generate_single_mutants(wt, alpha)})
    for seed, length, number, alpha in ((1, 8, 5, s_utils.DNAA), (2, 14, 3, s_utils.RNAA), (3, 30, 2, s_utils.AAS), (4, 0, 2, s_utils.DNAA)):
        random.seed(seed)
        out = s_utils.generate_random_sequences(length, number, alpha)
        cases["random_sequences"].append({"seed": seed, "length": length, "number": number, "alphabet": alpha, "out": out,
                                          "next_random": random.random()})
    for seed, seq, mu, alpha in ((5, "GATTACAG", 0.3, s_utils.DNAA), (6, "GATTACAG", 1.0, s_utils.DNAA), (7, "GATTACAG", 0.0, s_utils.DNAA),
                                 (8, "MKVLAAGIW" * 5, 2.0 / 45, s_utils.AAS)):
        random.seed(seed)
        outs = [s_utils.generate_random_mutant(seq, mu, alpha) for _ in range(4)]
        cases["random_mutant"].append({"seed": seed, "sequence": seq, "mu": mu, "alphabet": alpha, "out": outs,
                                       "next_random": random.random()})
    rng = np.random.default_rng(9)
    for L, A in ((8, 4), (5, 20)):
        base = np.eye(A)[rng.integers(0, A, L)]
        sample = np.zeros((L, A))
        for i in rng.choice(L, 3, replace=False):
            sample[i, rng.integers(0, A)] = rng.uniform(0.1, 2.0)
        out = s_utils.construct_mutant_from_sample(sample, base)
        cases["construct_mutant"].append({"base": base.tolist(), "sample": sample.tolist(), "out": out.tolist(), "dtype": str(out.dtype)})
    json.dump(cases, open(os.path.join(OUT, "sequence_generators.json"), "w"), indent=0)
    print("wrote sequence_generators.json", {k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
