"""Fixtures made by RUNNING the reference (generator scripts under tests/golden/, data only):

* explorer_traces.json (make_golden_explorers.py): every `model.get_fitness` call the reference's Explorer.run loop
  issues under Adalead / Random / GeneticAlgorithm, with the per-round cost and proposals;
* tf_binding_runlogs.npz, nam_identity_log.json (make_golden_runlogs.py): the run logs the reference ships.

CPU tests check the fixtures' own consistency and hold `flexs_amd.utils.rollouts.adalead_round` to the reference
Adalead's proposals, cost and random stream; GPU tests replay the call patterns and the logs through the engine."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import flexs_amd
from fakes import hashed_fitnesses
from flexs_amd.utils import rollouts


@pytest.fixture(scope="module")
def traces(golden_dir):
    return json.load(open(os.path.join(golden_dir, "explorer_traces.json")))


class HashedModel(flexs_amd.Model):
    def __init__(self, salt):
        super().__init__("recorder")
        self.salt, self.calls = salt, []

    def train(self, sequences, labels):
        pass

    def _fitness_function(self, sequences):
        self.calls.append([str(s) for s in sequences])
        return hashed_fitnesses(sequences, self.salt)


def test_call_patterns_are_what_survey_section_3_5_says(traces):
    """Shape of the hot path's input under each explorer, and the cost bookkeeping of explorer.py:126-175:
    model_cost in the log = running number of sequences the model was asked about."""
    runs = traces["runs"]
    for name, run in runs.items():
        sizes = [len(c) for c in run["calls"]]
        total = 0
        for rnd in run["rounds"]:
            total += sum(sizes[rnd["first_call"]:rnd["end_call"]])
            assert rnd["model_cost"] == total, name
            assert len(rnd["proposed"]) == len(rnd["model_score"])
            assert name == "Random" or len(set(rnd["proposed"])) == len(rnd["proposed"])     # random.py:92 samples with replacement
        assert run["rounds"][-1]["end_call"] == len(sizes)
    assert max(len(c) for c in runs["Adalead"]["calls"]) <= 20                      # eval_batch_size
    assert [len(c) for c in runs["Random"]["calls"]] == [2001, 2001, 2001]           # `<=` in random.py:81
    assert all(len(r["proposed"]) == 99 for r in runs["Adalead"]["rounds"])         # [: -batch : -1] keeps batch - 1
    assert all(r["train_size"] == 1 + 99 * i for i, r in enumerate(runs["Adalead"]["rounds"]))
    # the recorded model scores are the fake model's values: the fixture is self-consistent
    r0 = runs["Adalead"]["rounds"][0]
    assert np.array_equal(hashed_fitnesses(r0["proposed"], traces["model_salt"]), np.array(r0["model_score"]))
    assert np.array_equal(hashed_fitnesses(r0["proposed"], traces["landscape_salt"]), np.array(r0["true_score"]))


@pytest.mark.parametrize("run_name", ["Adalead", "Adalead_recomb"])
@pytest.mark.parametrize("fuse", [False, True])
def test_adalead_round_reproduces_the_reference_explorer(traces, run_name, fuse):
    """flexs_amd.utils.rollouts.adalead_round against the reference's Adalead.propose_sequences round by round: same
    proposals in the same order, same scores, same model.cost, same state of Python's `random` afterwards -- with the
    roots-plus-first-children fusion on and off (fused: fewer, larger calls, same everything else)."""
    run = traces["runs"][run_name]
    p = run["params"]
    random.seed(1234)
    np.random.seed(1234)
    model = HashedModel(traces["model_salt"])
    measured_seqs, measured_scores = [traces["start"]], list(hashed_fitnesses([traces["start"]], traces["landscape_salt"]))
    for rnd in run["rounds"]:
        first = len(model.calls)
        seqs, preds = rollouts.adalead_round(
            model, measured_seqs, measured_scores, sequences_batch_size=p["sequences_batch_size"],
            model_queries_per_batch=p["model_queries_per_batch"], alphabet=traces["alphabet"], mu=p["mu"],
            recomb_rate=p["recomb_rate"], threshold=p["threshold"], rho=p["rho"], eval_batch_size=p["eval_batch_size"], fuse=fuse)
        assert seqs.tolist() == rnd["proposed"] and preds.tolist() == rnd["model_score"]
        assert model.cost == rnd["model_cost"]
        assert hashlib.sha256(repr(random.getstate()).encode()).hexdigest()[:16] == rnd["random_state_after"]
        ref_calls = run["calls"][rnd["first_call"]:rnd["end_call"]]
        mine = model.calls[first:]
        if fuse:
            assert len(mine) < len(ref_calls) and sum(map(len, mine)) == sum(map(len, ref_calls))
            assert [s for c in mine for s in c] == [s for c in ref_calls for s in c]       # same sequences, same order
        else:
            assert mine == ref_calls
        measured_seqs += list(seqs)
        measured_scores += list(hashed_fitnesses(seqs, traces["landscape_salt"]))


# ------------------------------------------------------------------------------------------------ GPU replays
@pytest.mark.gpu
@pytest.mark.parametrize("run_name", ["Adalead", "Random", "GeneticAlgorithm"])
def test_replay_of_explorer_call_patterns_on_the_engine(traces, run_name):
    """The recorded calls, one by one, through a 3-member CNN ensemble on the GPU: every small call returns the bits
    the same sequences get in one big batch (zero-copy small-call path, position-segmented and bulk kernels are
    interchangeable), and the cost counters follow the log's model_cost column."""
    from flexs_amd.baselines import models as bm

    run = traces["runs"][run_name]
    members = [bm.CNN(8, 32, 100, traces["alphabet"], seed=s) for s in range(3)]
    ens = flexs_amd.Ensemble(members)
    everything = sorted({s for c in run["calls"] for s in c})
    ref = dict(zip(everything, flexs_amd.Ensemble(members).get_fitness(everything)))
    for m in members:
        m.cost = 0
    for rnd in run["rounds"]:
        for call in run["calls"][rnd["first_call"]:rnd["end_call"]]:
            got = ens.get_fitness(np.array(call) if len(call) % 2 else call)          # list and ndarray inputs alike
            assert got.dtype == np.float32 and got.tolist() == [ref[s] for s in call]
        assert ens.cost == rnd["model_cost"] and all(m.cost == rnd["model_cost"] for m in members)


@pytest.mark.gpu
def test_adalead_round_on_device_models_fused_equals_unfused(traces):
    from flexs_amd.baselines import models as bm

    p = traces["runs"]["Adalead"]["params"]
    out = []
    for fuse in (False, None):
        members = [bm.CNN(8, 32, 100, traces["alphabet"], seed=s) for s in range(3)]
        ens = flexs_amd.Ensemble(members)
        random.seed(5)
        seqs, preds = rollouts.adalead_round(ens, [traces["start"]], [0.5], sequences_batch_size=100, model_queries_per_batch=2000,
                                             alphabet=traces["alphabet"], mu=p["mu"], eval_batch_size=20, fuse=fuse)
        out.append((seqs.tolist(), preds.tolist(), ens.cost, members[0].cost, random.random()))
    assert out[0] == out[1] and len(out[0][0]) == 99


@pytest.mark.gpu
def test_tf_binding_run_logs_through_the_device_table(golden_dir, tmp_path):
    """The 26 757 (sequence, true_score) rows of the reference's CMA-ES run logs: laid out as an 8-mer file (with the
    8-mers on which the reference class returns 0.0 and 1.0, so that the min-max scaling of tf_binding.py:33-34 is
    the identity), every logged score must come back through the device-resident table."""
    fx = np.load(os.path.join(golden_dir, "tf_binding_runlogs.npz"))
    comp = str.maketrans("ACGT", "TGCA")
    assert int(fx["rows_in_logs"]) == 26757
    total = 0
    for tf in ("SIX6_REF_R1", "VAX2_REF_R1", "VSX1_REF_R1"):
        seqs = [s.decode() for s in fx[f"{tf}__sequences"]]
        vals = fx[f"{tf}__true_scores"]
        total += int(fx[f"{tf}__n_logged"])
        lines = ["8-mer\t8-mer.1\tE-score\tMedian\tZ-score"] + [f"{s}\t{s.translate(comp)[::-1]}\t{float(v)!r}\t0\t0" for s, v in zip(seqs, vals)]
        path = tmp_path / f"{tf}_8mers.txt"
        path.write_text("\n".join(lines) + "\n")
        land = flexs_amd.landscapes.TFBinding(str(path))
        got = land.get_fitness(seqs)
        host = np.array([land.sequences[s] for s in seqs])
        assert np.array_equal(got, host) and land.cost == len(seqs)                 # device table == the class's own dict
        assert np.abs(got - vals).max() <= 2e-16                                    # == the logged scores (text round trip of the file)
        # without the text round trip: the table filled from the logged values returns them bit for bit
        land.sequences = dict(zip(seqs, vals.tolist()))
        land._table = None
        order = np.random.default_rng(0).permutation(len(seqs))
        assert np.array_equal(land.get_fitness(np.array(seqs)[order]), vals[order])
        with pytest.raises(KeyError):
            land.get_fitness(["ACGTACGT" if "ACGTACGT" not in land.sequences else "AAAAAAAC"])
    assert total == 23311


@pytest.mark.gpu
def test_signal_strength_one_is_the_identity_on_the_shipped_log(golden_dir):
    """examples/robustness/adalead/1.csv: Adalead against NoisyAbstractModel(signal_strength=1); every logged
    model_score equals the true score.  Replayed round by round (train on everything measured, then score the round's
    proposals: explorer.py:157-165) through the device neighbour search and blend: alpha = 1 ** d = 1 exactly."""
    from flexs_amd.baselines.models import NoisyAbstractModel

    log = json.load(open(os.path.join(golden_dir, "nam_identity_log.json")))
    table = dict(zip(log["sequences"], log["true_score"]))

    class Logged(flexs_amd.Landscape):
        def _fitness_function(self, seqs):
            return np.array([table[str(s)] for s in seqs])

    land = Logged("rna")
    nam = NoisyAbstractModel(land, signal_strength=1)
    assert nam.name == log["model_name"]
    rounds = np.array(log["round"])
    seqs, truth = np.array(log["sequences"]), np.array(log["true_score"])
    np.random.seed(3)
    for r in range(1, log["rounds"] + 1):
        known = rounds < r
        nam.train(seqs[known], truth[known])
        batch = seqs[rounds == r]
        got = nam.get_fitness(batch)
        assert got.tolist() == [log["model_score"][i] for i in np.flatnonzero(rounds == r)]
        assert nam.cost == len(batch) * r and len(nam.cache) == known.sum() + len(batch)
