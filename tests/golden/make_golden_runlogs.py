#!/usr/bin/env python3
"""Turn the run logs the reference ships into fixtures (build container only; needs /root/reference):

    python tests/golden/make_golden_runlogs.py

* tf_binding_runlogs.npz -- every (sequence, true_score) row of paper_code/cloud/runs/cmaes/* (27 CMA-ES runs on the
  TF-binding landscapes VAX2 / VSX1 / SIX6, 26 757 rows, 23 311 distinct (landscape, 8-mer) pairs), checked here
  against the reference's own `TFBinding` class on the shipped 8-mer files (0 mismatches), plus, per landscape, the
  8-mers on which the reference class returns its minimum 0.0 and maximum 1.0 -- so that a test can lay the pairs
  out as an 8-mer file whose min-max normalisation (tf_binding.py:33-34) is the identity and must get every logged
  score back bit for bit through the device table.
* nam_identity_log.json -- examples/robustness/adalead/1.csv (Adalead against `NoisyAbstractModel(signal_strength=1)`,
  RNA L = 14): the 496 logged rows with model_score == true_score on every scored row, i.e. alpha = 1 => f_hat = f
  (noisy_abstract_model.py:93-94), together with the round structure needed to replay the cache growth.

The fixtures hold data only (sequences, scores, round numbers), no reference source.
"""
import glob
import importlib
import json
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def reference_tf_binding():
    sys.path.insert(0, REF)
    for name, sub in (("flexs", ""), ("flexs.landscapes", "landscapes"), ("flexs.utils", "utils")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "flexs", sub)]
        sys.modules[name] = m
    flexs = sys.modules["flexs"]
    flexs.types = importlib.import_module("flexs.types")
    flexs.Landscape = importlib.import_module("flexs.landscape").Landscape
    return importlib.import_module("flexs.landscapes.tf_binding")


def main():
    tfb = reference_tf_binding()
    per_tf, rows = {}, 0
    for path in sorted(glob.glob(os.path.join(REF, "paper_code/cloud/runs/cmaes/*"))):
        tf = os.path.basename(path).split("_start")[0]
        with open(path) as fh:
            meta = json.loads(fh.readline())
            df = pd.read_csv(fh)
        assert meta["landscape_name"] == "TF_Binding"
        rows += len(df)
        d = per_tf.setdefault(tf, {})
        for s, t in zip(df["sequence"], df["true_score"]):
            assert d.setdefault(s, t) == t                      # a landscape is a function
    out, mismatches = {}, 0
    for tf, d in sorted(per_tf.items()):
        land = tfb.TFBinding(os.path.join(REF, "flexs/landscapes/data/tf_binding", f"{tf}_8mers.txt"))
        seqs = sorted(d)
        want = land.get_fitness(seqs)
        mismatches += int((np.abs(want - np.array([d[s] for s in seqs])) > 1e-12).sum())
        lo = min(land.sequences, key=land.sequences.get)
        hi = max(land.sequences, key=land.sequences.get)
        assert land.sequences[lo] == 0.0 and land.sequences[hi] == 1.0
        seqs_all = seqs + [s for s in (lo, hi) if s not in d]
        vals = np.array([d.get(s, land.sequences[s]) for s in seqs_all], np.float64)
        out[f"{tf}__sequences"] = np.array(seqs_all, dtype="S8")
        out[f"{tf}__true_scores"] = vals
        out[f"{tf}__n_logged"] = np.array(len(seqs))
    assert mismatches == 0, mismatches
    out["rows_in_logs"] = np.array(rows)
    np.savez_compressed(os.path.join(OUT, "tf_binding_runlogs.npz"), **out)
    print(f"tf_binding_runlogs.npz: {rows} log rows, {sum(len(d) for d in per_tf.values())} distinct pairs, 0 mismatches "
          f"against the reference TFBinding class")

    path = os.path.join(REF, "examples/robustness/adalead/1.csv")
    with open(path) as fh:
        meta = json.loads(fh.readline())
        df = pd.read_csv(fh)
    scored = df["model_score"].notna()
    assert meta["model_name"] == "NAMb_ss1" and (df["model_score"][scored] == df["true_score"][scored]).all()
    json.dump({"model_name": meta["model_name"], "landscape_name": meta["landscape_name"], "rounds": int(meta["rounds"]),
               "sequences": df["sequence"].tolist(), "true_score": df["true_score"].tolist(),
               "model_score": [None if np.isnan(v) else v for v in df["model_score"]],
               "round": df["round"].astype(int).tolist()},
              open(os.path.join(OUT, "nam_identity_log.json"), "w"))
    print(f"nam_identity_log.json: {len(df)} rows, model_score == true_score on all {int(scored.sum())} scored rows")


if __name__ == "__main__":
    main()
