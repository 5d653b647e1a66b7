#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz by RUNNING the reference's own Python.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference package cannot be imported whole (`flexs/__init__.py:11` pulls
TensorFlow, tf_agents, cma, tape, ViennaRNA, pyrosetta -- none installable here),
so the pure-Python modules on the hot path are imported one by one under a
synthetic `flexs` parent package.  Nothing from the reference is copied into
this repository: the fixtures hold only INPUTS and the OUTPUTS the reference
code produced for them.

`editdistance` (third-party C++, setup.py:23) is absent; a textbook DP defined
below stands in for `editdistance.eval`, so the NoisyAbstractModel fixtures pin
the reference's cache / ordering / RNG logic, while the distance function itself
is pinned separately by the `ed_N_wt` known answers the reference registries
carry (edit_distance_known.json).
"""
import importlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def import_reference():
    sys.path.insert(0, REF)
    for name, sub in (("flexs", ""), ("flexs.baselines", "baselines"), ("flexs.baselines.models", "baselines/models"),
                      ("flexs.utils", "utils"), ("flexs.landscapes", "landscapes")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "flexs", sub)]
        sys.modules[name] = m
    ed = types.ModuleType("editdistance")
    ed.eval = _lev
    sys.modules["editdistance"] = ed
    flexs = sys.modules["flexs"]
    flexs.types = importlib.import_module("flexs.types")
    flexs.Landscape = importlib.import_module("flexs.landscape").Landscape
    mm = importlib.import_module("flexs.model")
    flexs.Model, flexs.LandscapeAsModel = mm.Model, mm.LandscapeAsModel
    flexs.Ensemble = importlib.import_module("flexs.ensemble").Ensemble
    s_utils = importlib.import_module("flexs.utils.sequence_utils")
    nam = importlib.import_module("flexs.baselines.models.noisy_abstract_model")
    ada = importlib.import_module("flexs.baselines.models.adaptive_ensemble")
    tfb = importlib.import_module("flexs.landscapes.tf_binding")
    return flexs, s_utils, nam, ada, tfb


def main():
    flexs, s_utils, nam_mod, ada_mod, tfb_mod = import_reference()
    rng = np.random.default_rng(20260927)

    # ---------------------------------------------------------------- encode
    enc = {"alphabets": {"AAS": s_utils.AAS, "RNAA": s_utils.RNAA, "DNAA": s_utils.DNAA, "BA": s_utils.BA}, "cases": []}
    for alpha_name, L, n in (("DNAA", 8, 6), ("RNAA", 14, 4), ("AAS", 31, 3), ("BA", 5, 3), ("DNAA", 1, 2)):
        alpha = enc["alphabets"][alpha_name]
        for _ in range(n):
            seq = "".join(alpha[i] for i in rng.integers(0, len(alpha), L))
            oh = s_utils.string_to_one_hot(seq, alpha)
            assert oh.dtype == np.float64
            enc["cases"].append({"alphabet": alpha_name, "sequence": seq, "one_hot": oh.astype(int).tolist(),
                                 "dtype": str(oh.dtype)})
    # repeated-character alphabet: `str.index` returns the first occurrence
    oh = s_utils.string_to_one_hot("ABA", "ABAC")
    enc["cases"].append({"alphabet_literal": "ABAC", "sequence": "ABA", "one_hot": oh.astype(int).tolist(),
                         "dtype": str(oh.dtype)})
    try:
        s_utils.string_to_one_hot("ATXG", s_utils.DNAA)
        enc["bad_char_exception"] = None
    except Exception as e:  # noqa: BLE001
        enc["bad_char_exception"] = type(e).__name__
    oh = s_utils.string_to_one_hot("", s_utils.DNAA)
    enc["empty_shape"] = list(oh.shape)
    json.dump(enc, open(os.path.join(OUT, "encode.json"), "w"), indent=0)

    # ---------------------------------------------------------------- decode
    dec_in, dec_alpha, dec_out = [], [], []
    for alpha_name, L in (("DNAA", 8), ("RNAA", 14), ("AAS", 20), ("AAS", 238)):
        alpha = enc["alphabets"][alpha_name]
        for k in range(4):
            x = rng.standard_normal((L, len(alpha)))
            if k == 1:                      # ties: quantise so several maxima coincide
                x = np.round(x)
            if k == 2:                      # all-equal rows -> index 0
                x[: L // 2] = 0.0
            if k == 3:                      # extra column (DynaPPO observations have A+1), argmax in range
                x = np.abs(x)
            dec_in.append(x)
            dec_alpha.append(alpha_name)
            dec_out.append(s_utils.one_hot_to_string(x, alpha))
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **{f"x{i}": x for i, x in enumerate(dec_in)})
    json.dump({"alphabet": dec_alpha, "strings": dec_out}, open(os.path.join(OUT, "decode.json"), "w"), indent=0)

    # -------------------------------------------------------------- ensemble
    class FixedModel(flexs.Model):
        def __init__(self, name, values):
            super().__init__(name)
            self.values = values
            self.trained = 0

        def _fitness_function(self, sequences):
            return self.values[: len(sequences)]

        def train(self, sequences, labels):
            self.trained += 1

    ens = {"cases": []}
    ens_arrays = {}
    seqs = ["".join("TGCA"[i] for i in rng.integers(0, 4, 8)) for _ in range(97)]
    for ci, (M, dt) in enumerate(((1, "float32"), (2, "float32"), (3, "float32"), (5, "float32"), (8, "float32"),
                                  (11, "float32"), (16, "float32"), (3, "float64"), (8, "float64"), (130, "float32"))):
        vals = (rng.standard_normal((97, M)) * rng.choice([1e-3, 1.0, 1e3], (97, M))).astype(dt)
        models = [FixedModel(f"m{j}", np.ascontiguousarray(vals[:, j])) for j in range(M)]
        e = flexs.Ensemble(models)
        out = e.get_fitness(seqs)
        out2 = e.get_fitness(seqs[:10])
        ident = flexs.Ensemble(models, combine_with=lambda x: x).get_fitness(seqs)
        e.train(seqs, np.zeros(len(seqs)))
        ens_arrays[f"in{ci}"] = vals
        ens_arrays[f"out{ci}"] = out
        ens_arrays[f"ident{ci}"] = ident
        ens["cases"].append({"M": M, "dtype": dt, "out_dtype": str(out.dtype), "name": e.name,
                             "ens_cost": e.cost, "member_costs": [m.cost for m in models],
                             "member_trained": [m.trained for m in models], "n_second_call": len(out2)})
    # AdaptiveEnsemble: r2 weights + weighted sum (adaptive_ensemble.py:12-26,97-102)
    preds = rng.standard_normal((4, 40))
    labels = preds[0] * 0.7 + preds[2] * 0.2 + rng.standard_normal(40) * 0.3
    w = ada_mod.r2_weights(preds, labels)
    vals = rng.standard_normal((97, 4)).astype(np.float32)
    models = [FixedModel(f"a{j}", np.ascontiguousarray(vals[:, j])) for j in range(4)]
    ae = ada_mod.AdaptiveEnsemble(models)
    out_default = ae.get_fitness(seqs)
    ae.weights = w
    out_w = ae.get_fitness(seqs)
    ens_arrays.update(ada_preds=preds, ada_labels=labels, ada_w=w, ada_in=vals, ada_out_default=out_default, ada_out_w=out_w)
    ens["adaptive"] = {"name": ae.name, "default_weights": (np.ones(4) / 4).tolist(), "out_dtype": str(out_w.dtype),
                       "cost": ae.cost, "member_costs": [m.cost for m in models]}
    ens["sequences"] = seqs
    np.savez_compressed(os.path.join(OUT, "ensemble.npz"), **ens_arrays)
    json.dump(ens, open(os.path.join(OUT, "ensemble.json"), "w"), indent=0)

    # -------------------------------------------------------------- TFBinding
    tf_file = os.path.join(REF, "flexs/landscapes/data/tf_binding/SIX6_REF_R1_8mers.txt")
    tfl = tfb_mod.TFBinding(tf_file)
    all8 = ["".join("TGCA"[(i >> (2 * k)) & 3] for k in range(8)) for i in rng.choice(65536, 300, replace=False)]
    tf_fix = {"problem": "SIX6_REF_R1", "name": tfl.name,
              "tutorial_known_answer": {"sequence": "ATTATGTT", "value": float(tfl.get_fitness(["ATTATGTT"])[0])},
              "sample_sequences": all8, "sample_values": tfl.get_fitness(all8).tolist(), "cost_after": tfl.cost}
    json.dump(tf_fix, open(os.path.join(OUT, "tf_binding.json"), "w"), indent=0)

    # -------------------------------------------------- NoisyAbstractModel traces
    class TableLandscape(flexs.Landscape):
        def __init__(self, table, default_fn):
            super().__init__("Table")
            self.table = table
            self.default_fn = default_fn
            self.touched = {}

        def _fitness_function(self, sequences):
            out = []
            for s in sequences:
                s = str(s)
                v = self.table[s] if s in self.table else self.default_fn(s)
                self.touched[s] = float(v)
                out.append(v)
            return np.array(out)

    def mutate(seq, alpha, nmut, allow_indel_like=False):
        s = list(seq)
        for _ in range(nmut):
            i = int(rng.integers(0, len(s)))
            s[i] = alpha[int(rng.integers(0, len(alpha)))]
        if allow_indel_like and rng.random() < 0.5:     # rotate -> Levenshtein < Hamming
            s = s[1:] + s[:1]
        return "".join(s)

    traces = []
    for ti, (alpha, L, ss, seed, negative) in enumerate((
            ("TGCA", 8, 0.9, 11, False), ("TGCA", 8, 0.0, 12, False), ("TGCA", 8, 1.0, 13, False),
            ("UGCA", 14, 0.75, 14, False), (s_utils.AAS, 70, 0.9, 15, False), ("TGCA", 8, 0.5, 16, True),
            (s_utils.AAS, 238, 0.9, 17, False))):
        if alpha == "TGCA" and not negative:
            table = tfl.sequences

            def default_fn(s):
                raise KeyError(s)
        else:
            table = {}
            base = 1.0 if not negative else 0.2

            def default_fn(s, _b=base, _neg=negative):
                h = sum((i + 1) * ord(c) for i, c in enumerate(s)) % 1000
                v = h / 1000.0 * _b
                return v - 0.1 if _neg else v
        land = TableLandscape(table, default_fn)
        start = "".join(alpha[i] for i in rng.integers(0, len(alpha), L))
        train_seqs = [start] + [mutate(start, alpha, int(rng.integers(1, 4))) for _ in range(12)]
        train_labels = land._fitness_function(train_seqs).tolist()
        batches = []
        pool = list(train_seqs)
        for b in range(4):
            batch = []
            for _ in range(int(rng.integers(5, 12))):
                parent = pool[int(rng.integers(0, len(pool)))]
                batch.append(mutate(parent, alpha, int(rng.integers(1, 5)), allow_indel_like=True))
            if b >= 1:
                batch.append(batch[0])                  # duplicate inside a batch
                batch.append(pool[-1])                  # already-cached sequence
            batches.append(batch)
            pool.extend(batch)
        np.random.seed(seed)
        model = nam_mod.NoisyAbstractModel(land, signal_strength=ss)
        empty_first = None
        if ti == 0:
            # empty-cache special case (noisy_abstract_model.py:44-45) on a fresh model
            m0 = nam_mod.NoisyAbstractModel(land, signal_strength=ss)
            np.random.seed(seed)
            empty_first = {"query": [start], "out": m0.get_fitness([start]).tolist(), "cache_len": len(m0.cache)}
            np.random.seed(seed)
        model.train(train_seqs, train_labels)
        cost0 = land.cost
        outs, costs, cache_lens, mcosts = [], [], [], []
        for batch in batches:
            o = model.get_fitness(batch)
            outs.append(o.tolist())
            costs.append(land.cost - cost0)
            cache_lens.append(len(model.cache))
            mcosts.append(model.cost)
        after = float(np.random.random())              # RNG position after the trace
        traces.append({"alphabet": alpha, "L": L, "ss": ss, "seed": seed, "name": model.name,
                       "train_sequences": train_seqs, "train_labels": train_labels, "batches": batches,
                       "outputs": outs, "landscape_cost": costs, "cache_len": cache_lens, "model_cost": mcosts,
                       "cache_keys_in_order": list(model.cache.keys()), "rng_next_random": after,
                       "landscape_values": land.touched, "empty_first": empty_first, "out_dtype": "float64"})
    json.dump({"traces": traces}, open(os.path.join(OUT, "nam_traces.json"), "w"), indent=0)

    # ------------------------------------------- edit-distance known answers
    # Registries name their starting sequences by edit distance to wild type:
    # bert_gfp.py:36-46 (GFP, L=238), rosetta.py:207-225 (3msi L=66, 3mx7 L=90).
    import re

    def class_attrs(path, names):
        src = open(path).read()
        ns = {}
        for n in names:
            m = re.search(rf"^\s*{n}\s*=\s*(\(.*?\)|\{{.*?\}})\s*$", src, re.S | re.M)
            ns[n] = eval(m.group(1))  # noqa: S307 - literal strings/dicts from the reference source
        return ns

    gfp = class_attrs(os.path.join(REF, "flexs/landscapes/bert_gfp.py"), ["gfp_wt_sequence", "starts"])
    three2one = dict(ALA="A", ARG="R", ASN="N", ASP="D", CYS="C", GLN="Q", GLU="E", GLY="G", HIS="H", ILE="I",
                     LEU="L", LYS="K", MET="M", PHE="F", PRO="P", SER="S", THR="T", TRP="W", TYR="Y", VAL="V")

    def pdb_sequence(path):
        seq, seen = [], set()
        for line in open(path):
            if line.startswith("ATOM") and line[12:16].strip() == "CA" and line[16] in " A":
                key = (line[21], line[22:27])
                if key not in seen and line[21] == "A":
                    seen.add(key)
                    seq.append(three2one[line[17:20]])
        return "".join(seq)

    sys.modules["torch"] = importlib.import_module("torch")
    ros_src = open(os.path.join(REF, "flexs/landscapes/rosetta.py")).read()
    known = []
    for name, s in gfp["starts"].items():
        known.append({"family": "gfp", "name": name, "wt": gfp["gfp_wt_sequence"], "seq": s,
                      "named": int(name.split("_")[1])})
    for pdb in ("3msi", "3mx7"):
        wt = pdb_sequence(os.path.join(REF, f"flexs/landscapes/data/rosetta/{pdb}.pdb"))
        blk = ros_src[ros_src.index(f'"{pdb}": {{'):]
        for m in re.finditer(r'"(ed_(\d+)_wt)":\s*"([A-Z]+)"', blk[: blk.index("},\n        },") if "},\n        }," in blk else len(blk)]):
            known.append({"family": pdb, "name": m.group(1), "wt": wt, "seq": m.group(3), "named": int(m.group(2))})
    for k in known:
        k["levenshtein_dp"] = _lev(k["seq"], k["wt"])
        k["hamming"] = sum(a != b for a, b in zip(k["seq"], k["wt"])) if len(k["seq"]) == len(k["wt"]) else None
    json.dump({"known": known}, open(os.path.join(OUT, "edit_distance_known.json"), "w"), indent=0)
    print("wrote fixtures:", sorted(f for f in os.listdir(OUT) if f.endswith((".json", ".npz"))))
    for k in known:
        print(k["family"], k["name"], len(k["wt"]), len(k["seq"]), k["named"], k["levenshtein_dp"], k["hamming"])


if __name__ == "__main__":
    main()
