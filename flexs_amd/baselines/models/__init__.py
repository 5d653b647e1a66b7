"""`baselines.models` -- device-backed counterparts of flexs/baselines/models/__init__.py."""
from flexs_amd.baselines.models.adaptive_ensemble import AdaptiveEnsemble  # noqa: F401
from flexs_amd.baselines.models.cnn import CNN  # noqa: F401
from flexs_amd.baselines.models.dyna_ppo_ensemble import DynaPPOEnsemble  # noqa: F401
from flexs_amd.baselines.models.global_epistasis_model import GlobalEpistasisModel  # noqa: F401
from flexs_amd.baselines.models.keras_model import Architecture, KerasModel  # noqa: F401
from flexs_amd.baselines.models.mlp import MLP  # noqa: F401
from flexs_amd.baselines.models.noisy_abstract_model import NoisyAbstractModel  # noqa: F401
