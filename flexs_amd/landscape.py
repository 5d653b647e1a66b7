"""`Landscape` base class -- same contract as flexs/landscape.py:9-45."""
import abc
import os

import numpy as np

from flexs_amd.types import SEQUENCES_TYPE

_BaseLandscape = None
if os.environ.get("FLEXS_AMD_BIND_FLEXS") == "1":      # become real flexs subclasses
    import flexs as _flexs

    _BaseLandscape = _flexs.Landscape

if _BaseLandscape is not None:
    Landscape = _BaseLandscape
else:

    class Landscape(abc.ABC):
        """
        Base class for all landscapes and for `Model`.

        Attributes:
            cost (int): Number of sequences whose fitness has been evaluated.
            name (str): Human-readable name used when logging explorer runs.
        """

        def __init__(self, name: str):
            self.cost = 0
            self.name = name

        @abc.abstractmethod
        def _fitness_function(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
            pass

        def get_fitness(self, sequences: SEQUENCES_TYPE) -> np.ndarray:
            """Score sequences: `cost += len(sequences)` then `_fitness_function`
            (flexs/landscape.py:29-45).  Not to be overridden."""
            self.cost += len(sequences)
            return self._fitness_function(sequences)
