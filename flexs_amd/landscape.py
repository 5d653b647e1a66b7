"""`Landscape`: the root of the plugin API (contract of flexs/landscape.py:9-45).

A landscape maps sequences to fitness values and counts how many it has been asked about:
`get_fitness` adds `len(sequences)` to `cost` and then defers to the subclass hook
`_fitness_function`.  Explorers read and reset `cost` (flexs/explorer.py:126) and log `name`.
"""
import abc
import os

from flexs_amd.types import FITNESS_TYPE, SEQUENCES_TYPE


def _reference_class(attr):
    """With FLEXS_AMD_BIND_FLEXS=1 the classes of this package ARE (subclasses of) the reference's,
    so `isinstance(model, flexs.Ensemble)`-style checks in reference code hold."""
    if os.environ.get("FLEXS_AMD_BIND_FLEXS") != "1":
        return None
    import flexs

    return getattr(flexs, attr)


class _Landscape(abc.ABC):
    def __init__(self, name: str):
        self.name = name      # appears in the run-log metadata
        self.cost = 0         # sequences scored so far

    @abc.abstractmethod
    def _fitness_function(self, sequences: SEQUENCES_TYPE) -> FITNESS_TYPE:
        """Subclass hook: score `sequences` (no bookkeeping here)."""

    def get_fitness(self, sequences: SEQUENCES_TYPE) -> FITNESS_TYPE:
        """Public entry point -- not meant to be overridden (flexs/landscape.py:33-35)."""
        self.cost = self.cost + len(sequences)
        return self._fitness_function(sequences)


_Landscape.__doc__ = "Base class of every landscape and model: `name`, `cost`, `get_fitness`."
Landscape = _reference_class("Landscape") or _Landscape
