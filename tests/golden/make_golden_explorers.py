#!/usr/bin/env python3
"""Call-pattern traces of the reference's own explorers (build container only; needs /root/reference):

    python tests/golden/make_golden_explorers.py

Runs the reference's `Explorer.run` loop (flexs/explorer.py:115-184) with the reference's `Adalead`
(baselines/explorers/adalead.py), `Random` (random.py) and `GeneticAlgorithm` (genetic_algorithm.py) against a
recording model and landscape whose values are a pure function of the sequence text (tests/fakes.py), with every RNG
seeded, and writes explorer_traces.json:

* for every `model.get_fitness` call, in order: the sequences it was given (this is the shape of the hot path's
  input under each explorer: Adalead <= eval_batch_size per call, Random one ~2000-sequence call per round, ...);
* per round: the call index range, `model.cost` as the run log records it (`model_cost` column), the proposed
  sequences and their model scores, and the size of the training set handed to `model.train`;
* for Adalead additionally the state of Python's `random` generator after each round (a hash), so that a
  re-implementation of the roll-out loop can be held to the same random stream.

`flexs/__init__.py` cannot be imported here (TensorFlow etc.), so the needed modules are imported one by one under a
synthetic parent package, as in make_golden.py.  pandas 2 dropped `DataFrame.append`, which explorer.py:166 uses: it
is restored for the run.  The fixture holds inputs and outputs only, no reference source.
"""
import hashlib
import importlib
import json
import os
import random
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from fakes import hashed_fitnesses  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for name, sub in (("flexs", ""), ("flexs.baselines", "baselines"), ("flexs.baselines.explorers", "baselines/explorers"),
                      ("flexs.utils", "utils")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "flexs", sub)]
        sys.modules[name] = m
    flexs = sys.modules["flexs"]
    flexs.types = importlib.import_module("flexs.types")
    flexs.Landscape = importlib.import_module("flexs.landscape").Landscape
    mm = importlib.import_module("flexs.model")
    flexs.Model, flexs.LandscapeAsModel = mm.Model, mm.LandscapeAsModel
    flexs.Explorer = importlib.import_module("flexs.explorer").Explorer
    flexs.utils = sys.modules["flexs.utils"]
    flexs.utils.sequence_utils = importlib.import_module("flexs.utils.sequence_utils")
    ex = {"Adalead": importlib.import_module("flexs.baselines.explorers.adalead").Adalead,
          "Random": importlib.import_module("flexs.baselines.explorers.random").Random,
          "GeneticAlgorithm": importlib.import_module("flexs.baselines.explorers.genetic_algorithm").GeneticAlgorithm}
    return flexs, ex


def rng_digest():
    return hashlib.sha256(repr(random.getstate()).encode()).hexdigest()[:16]


def main():
    flexs, explorers = import_reference()
    pd.DataFrame.append = lambda self, other: pd.concat([self, other])        # explorer.py:166 (pandas < 2 API)

    class Recorder(flexs.Model):
        def __init__(self):
            super().__init__("recorder")
            self.calls, self.train_sizes, self.marks = [], [], []

        def train(self, sequences, labels):
            self.train_sizes.append(len(sequences))
            self.marks.append(len(self.calls))                                # first call index of the round

        def _fitness_function(self, sequences):
            seqs = [str(s) for s in sequences]
            self.calls.append(seqs)
            return hashed_fitnesses(seqs, salt=1)

    class Truth(flexs.Landscape):
        def _fitness_function(self, sequences):
            return hashed_fitnesses(sequences, salt=2)

    start, alphabet = "GCTCGAGC", "ACGT"                                      # tf_binding registry start, DNA 8-mers
    specs = {
        "Adalead": lambda m: explorers["Adalead"](m, rounds=3, sequences_batch_size=100, model_queries_per_batch=2000,
                                                  starting_sequence=start, alphabet=alphabet),
        "Adalead_recomb": lambda m: explorers["Adalead"](m, rounds=2, sequences_batch_size=40, model_queries_per_batch=600,
                                                         starting_sequence=start, alphabet=alphabet, rho=1, recomb_rate=0.2,
                                                         eval_batch_size=8, mu=2),
        "Random": lambda m: explorers["Random"](m, rounds=3, starting_sequence=start, sequences_batch_size=100,
                                                model_queries_per_batch=2000, alphabet=alphabet, seed=7),
        "GeneticAlgorithm": lambda m: explorers["GeneticAlgorithm"](
            m, rounds=3, starting_sequence=start, sequences_batch_size=100, model_queries_per_batch=2000, alphabet=alphabet,
            population_size=100, parent_selection_strategy="top-proportion", children_proportion=0.2,
            parent_selection_proportion=0.5, seed=7),
    }
    out = {"start": start, "alphabet": alphabet, "model_salt": 1, "landscape_salt": 2, "runs": {}}
    for name, make in specs.items():
        random.seed(1234)
        np.random.seed(1234)
        model, land = Recorder(), Truth("truth")
        explorer = make(model)
        digests = []
        if name.startswith("Adalead"):
            propose = explorer.propose_sequences

            def wrapped(measured, _p=propose):
                r = _p(measured)
                digests.append(rng_digest())
                return r

            explorer.propose_sequences = wrapped
        data, meta = explorer.run(land, verbose=True)
        rounds = []
        marks = model.marks + [len(model.calls)]
        for r in range(1, explorer.rounds + 1):
            rows = data[data["round"] == r]
            rounds.append({"first_call": marks[r - 1], "end_call": marks[r], "train_size": model.train_sizes[r - 1],
                           "model_cost": int(rows["model_cost"].iloc[0]), "proposed": rows["sequence"].tolist(),
                           "model_score": rows["model_score"].tolist(), "true_score": rows["true_score"].tolist(),
                           "random_state_after": digests[r - 1] if digests else None})
        out["runs"][name] = {"explorer_name": explorer.name, "params": {k: v for k, v in vars(explorer).items()
                                                                        if isinstance(v, (int, float, str)) and k != "name"},
                             "calls": model.calls, "rounds": rounds, "landscape_cost": land.cost}
        print(name, "calls", len(model.calls), "sizes", sorted({len(c) for c in model.calls})[:8], "cost per round",
              [r["model_cost"] for r in rounds])
    json.dump(out, open(os.path.join(OUT, "explorer_traces.json"), "w"))
    print("explorer_traces.json", os.path.getsize(os.path.join(OUT, "explorer_traces.json")), "bytes")


if __name__ == "__main__":
    main()
