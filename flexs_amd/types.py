"""Type aliases of the public API.

`SEQUENCES_TYPE` has the meaning of `flexs.types.SEQUENCES_TYPE` (flexs/types.py:6): whatever
`get_fitness` accepts -- a Python sequence of strings or a NumPy array of `str_` / `bytes_`
(the `S` dtype is this package's zero-copy fast path, see `flexs_amd._native.sequences_to_bytes`).
"""
import typing

import numpy

SEQUENCES_TYPE = typing.Union[typing.List[str], numpy.ndarray]
FITNESS_TYPE = numpy.ndarray          # (N,) float32 for the Keras-type surrogates, float64 for NoisyAbstractModel
COMBINE_TYPE = typing.Callable[[numpy.ndarray], numpy.ndarray]   # (N, M) -> (N,) ensemble reduction
