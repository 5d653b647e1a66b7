"""`Model` / `LandscapeAsModel` -- same contract as flexs/model.py:11-54."""
import abc
import os
from typing import Any, List

import numpy as np

from flexs_amd.landscape import Landscape
from flexs_amd.types import SEQUENCES_TYPE

if os.environ.get("FLEXS_AMD_BIND_FLEXS") == "1":
    import flexs as _flexs

    Model = _flexs.Model
    LandscapeAsModel = _flexs.LandscapeAsModel
else:

    class Model(Landscape, abc.ABC):
        """Landscape + `train` (flexs/model.py:11-27)."""

        @abc.abstractmethod
        def train(self, sequences: SEQUENCES_TYPE, labels: List[Any]):
            pass

    class LandscapeAsModel(Model):
        """Wrap a landscape as a perfect model (flexs/model.py:30-54)."""

        def __init__(self, landscape: Landscape):
            super().__init__(f"LandscapeAsModel={landscape.name}")
            self.landscape = landscape

        def _fitness_function(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
            return self.landscape._fitness_function(sequences)

        def train(self, sequences: SEQUENCES_TYPE, labels: List[Any]):
            pass
