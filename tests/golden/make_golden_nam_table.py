#!/usr/bin/env python3
"""Generate tests/golden/nam_table_traces.json by RUNNING the reference's NoisyAbstractModel
(flexs/baselines/models/noisy_abstract_model.py) over landscapes that are COMPLETE k-mer tables.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_nam_table.py

Why a second NAM fixture: `nam_traces.json` pins the product class on landscapes that are Python
dicts; the fused device path (`fx_cache_nam_query`: neighbour search + both table look-ups + blend in
one submission) only engages when the landscape is a device table holding EVERY k-mer, which none of
those traces' landscapes is.  Here the landscape of the reference run is a table over all 4^L
sequences (values committed with the trace, so the GPU test can put the same table on the device),
and the reference's own class produces the outputs, the oracle-call counts, the cache order and the
position of NumPy's global RNG.  Data only: inputs and the outputs the reference code produced.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402  (same stubbed import of the reference's pure-Python modules)


def main():
    flexs, _s_utils, nam_mod, _ada, _tfb = import_reference()
    traces = []
    for ti, (alpha, L, ss, seed, neg_frac) in enumerate((("ACGT", 6, 0.8, 21, 0.0), ("TGCA", 5, 0.9, 22, 0.0),
                                                         ("ACGT", 5, 0.5, 23, 0.03), ("UGCA", 4, 0.0, 24, 0.0),
                                                         ("ACGT", 5, 1.0, 25, 0.0))):
        rng = np.random.default_rng(1000 + ti)
        n_all = len(alpha) ** L
        vals = rng.uniform(0.0, 1.0, n_all)
        if neg_frac:
            neg = rng.random(n_all) < neg_frac
            vals[neg] = -rng.uniform(0.1, 1.0, int(neg.sum()))
        # index of a sequence = its characters' alphabet positions, 2 bits each, FIRST character in the lowest bits
        # (the packing of fx_table_create / flexs_amd.landscapes.TFBinding)
        all_seqs = ["".join(alpha[(i >> (2 * k)) & 3] for k in range(L)) for i in range(n_all)]
        table = dict(zip(all_seqs, vals.tolist()))

        class TableLandscape(flexs.Landscape):
            def __init__(self):
                super().__init__("Table")

            def _fitness_function(self, sequences):
                return np.array([table[str(s)] for s in sequences])

        land = TableLandscape()
        order = rng.permutation(n_all)
        pool = [all_seqs[i] for i in order]
        train = [s for s in pool if table[s] >= 0][:30]
        batches = []
        at = 0
        for b in range(10):
            n = (1, 3, 20, 60, 7)[b % 5]
            batch = [pool[(40 + at + j) % n_all] for j in range(n)]     # (a small table wraps around: later batches meet cached sequences)
            at += n
            if b >= 2:
                batch = batch + [batch[0], train[b]]          # a duplicate inside the batch, an already-cached sequence
            batches.append(batch)
        batches.append([pool[(30 + j) % n_all] for j in range(70)])   # mostly cached by now
        np.random.seed(seed)
        model = nam_mod.NoisyAbstractModel(land, signal_strength=ss)
        model.train(train, [table[s] for s in train])
        outs, costs, cache_lens, mcosts = [], [], [], []
        for batch in batches:
            o = model.get_fitness(batch)
            outs.append(o.tolist())
            costs.append(land.cost)
            cache_lens.append(len(model.cache))
            mcosts.append(model.cost)
        traces.append({"alphabet": alpha, "L": L, "ss": ss, "seed": seed, "name": model.name, "table_values": vals.tolist(),
                       "train_sequences": train, "train_labels": [table[s] for s in train], "batches": batches,
                       "outputs": outs, "landscape_cost": costs, "cache_len": cache_lens, "model_cost": mcosts,
                       "cache_keys_in_order": list(model.cache.keys()), "rng_next_random": float(np.random.random()),
                       "has_negative_values": bool(neg_frac)})
    json.dump({"traces": traces, "index_rule": "index = sum(alphabet.index(seq[k]) << (2 k)), first character in the lowest bits"},
              open(os.path.join(HERE, "nam_table_traces.json"), "w"), indent=0)
    print("wrote nam_table_traces.json:", [(t["L"], t["ss"], len(t["table_values"])) for t in traces])


if __name__ == "__main__":
    main()
