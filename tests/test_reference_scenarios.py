"""The reference's own test scenarios for this path (tests/test_models.py:36-99, tests/test_landscapes.py:8-54),
restated against `flexs_amd` with the same fakes and the same assertions.  Data files the reference ships (or
expects) are replaced by synthetic files of the same format, because nothing under /root/reference exists on
the GPU box."""
import json

import numpy as np
import pytest

import flexs_amd
from flexs_amd import baselines
from flexs_amd.utils import sequence_utils as s_utils

rng = np.random.default_rng()


class FakeModel(flexs_amd.Model):
    def _fitness_function(self, sequences):
        return rng.random(size=len(sequences))

    def train(self, *args, **kwargs):
        pass


class FakeLandscape(flexs_amd.Landscape):
    def _fitness_function(self, sequences):
        return rng.random(size=len(sequences))


class FakeConstantModel(flexs_amd.Model):
    def __init__(self, constant):
        super().__init__(name="ConstantModel")
        self.constant = constant

    def _fitness_function(self, sequences):
        return np.ones(len(sequences)) * self.constant

    def train(self, *args, **kwargs):
        pass


def test_adaptive_ensemble():
    """tests/test_models.py:36-52 (foreign members: the host-stacked path, no GPU needed)."""
    models = [FakeConstantModel(1), FakeConstantModel(2)]
    ens = baselines.models.AdaptiveEnsemble(models)
    assert np.sum(ens.weights) == 1
    assert ens.get_fitness(["ATC"]) == 1.5

    models = [FakeModel(name="FakeModel") for _ in range(2)]
    ens = baselines.models.AdaptiveEnsemble(models)
    ens.train(["ATC"] * 15, list(range(15)))
    assert np.any(ens.weights != np.ones(len(models)) / len(models))
    assert np.isclose(np.sum(ens.weights), 1)


@pytest.mark.gpu
def test_keras_models():
    """tests/test_models.py:55-77."""
    cnn = baselines.models.CNN(seq_len=3, num_filters=1, hidden_size=1, kernel_size=2, alphabet=s_utils.DNAA)
    assert cnn.get_fitness(["ATC"]).shape == (1,)
    gem = baselines.models.GlobalEpistasisModel(seq_len=3, hidden_size=1, alphabet=s_utils.DNAA)
    assert gem.get_fitness(["ATC"]).shape == (1,)
    mlp = baselines.models.MLP(seq_len=3, hidden_size=1, alphabet=s_utils.DNAA)
    assert mlp.get_fitness(["ATC"]).shape == (1,)


@pytest.mark.gpu
def test_noisy_abstract_model():
    """tests/test_models.py:80-99."""
    nam = baselines.models.NoisyAbstractModel(landscape=FakeLandscape(name="FakeLandscape"))
    assert len(nam.cache) == 0
    fitness = nam.get_fitness(["ATC"])
    assert len(nam.cache) == 1
    assert nam.get_fitness(["ATC"]) == fitness

    nam = baselines.models.NoisyAbstractModel(landscape=FakeConstantModel(2), signal_strength=1)
    assert nam.get_fitness(["ATC"]) == [2]

    nam = baselines.models.NoisyAbstractModel(landscape=FakeConstantModel(2), signal_strength=0)
    nam.get_fitness(["ATC"])
    assert nam.get_fitness(["ATG"]) != [2]            # flaky in the reference too, but extremely unlikely to fail


@pytest.mark.gpu
def test_tf_binding(tmp_path):
    """tests/test_landscapes.py:47-54 on a synthetic 8-mer file covering all 4^8 sequences."""
    import itertools

    comp = str.maketrans("ACGT", "TGCA")
    seen, lines = set(), ["8-mer\t8-mer.1\tE-score\tMedian\tZ-score"]
    for t in itertools.product("ACGT", repeat=8):
        s = "".join(t)
        rc = s.translate(comp)[::-1]
        if s in seen or rc in seen:
            continue
        seen.update((s, rc))
        lines.append(f"{s}\t{rc}\t{rng.uniform(-0.5, 0.5):.5f}\t{rng.uniform(0, 1e4):.2f}\t{rng.normal():.4f}")
    (tmp_path / "SIX6_REF_R1_8mers.txt").write_text("\n".join(lines) + "\n")
    problem = flexs_amd.landscapes.tf_binding.registry(str(tmp_path))["SIX6_REF_R1"]
    landscape = flexs_amd.landscapes.TFBinding(**problem["params"])
    test_seqs = s_utils.generate_random_sequences(8, 100, s_utils.DNAA)
    out = landscape.get_fitness(test_seqs)
    assert out.shape == (100,) and ((out >= 0) & (out <= 1)).all() and landscape.cost == 100


@pytest.mark.gpu
def test_additive_aav_packaging(tmp_path):
    """tests/test_landscapes.py:8-13 on a synthetic single-substitution file for residues 450-540."""
    from flexs_amd.landscapes import additive_aav_packaging as aav

    data = {str(pos): {aa: {"log2_heart_v_wt": float(rng.normal(0, 1.5)), "log2_packaging_v_wt": float(rng.normal(-2, 3))}
                       for aa in s_utils.AAS if rng.random() < 0.9 or aa == "A"}
            for pos in range(450, 540)}
    path = tmp_path / "AAV2_single_subs.json"
    path.write_text(json.dumps(data))
    problem = aav.registry()["heart"]
    landscape = flexs_amd.landscapes.AdditiveAAVPackaging(data_file=str(path), **problem["params"])
    test_seqs = s_utils.generate_random_sequences(90, 100, s_utils.AAS)
    out = landscape.get_fitness(test_seqs)
    assert out.shape == (100,) and (out >= 0).all() and landscape.cost == 100
