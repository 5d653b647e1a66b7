"""`Model` (a landscape that can be trained) and `LandscapeAsModel` (a perfect model) --
contract of flexs/model.py:11-54."""
import abc
from typing import Any, List

from flexs_amd.landscape import Landscape, _reference_class
from flexs_amd.types import FITNESS_TYPE, SEQUENCES_TYPE


class _Model(Landscape, abc.ABC):
    """Adds `train`, which the explorer calls once per round with everything measured so far
    (flexs/explorer.py:157-160)."""

    @abc.abstractmethod
    def train(self, sequences: SEQUENCES_TYPE, labels: List[Any]):
        """Update the model from measured (sequence, fitness) pairs."""


class _LandscapeAsModel(_Model):
    """The oracle itself behind the model interface (for experiments with a perfect surrogate)."""

    def __init__(self, landscape: Landscape):
        super().__init__(name="LandscapeAsModel=" + landscape.name)
        self.landscape = landscape

    def train(self, sequences: SEQUENCES_TYPE, labels: List[Any]):
        return None           # nothing to learn

    def _fitness_function(self, sequences: SEQUENCES_TYPE) -> FITNESS_TYPE:
        # the wrapped landscape's cost is NOT charged (flexs/model.py:49-50 calls the private hook)
        return self.landscape._fitness_function(sequences)


Model = _reference_class("Model") or _Model
LandscapeAsModel = _reference_class("LandscapeAsModel") or _LandscapeAsModel
