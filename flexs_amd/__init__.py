"""flexs_amd -- MI355X-native scoring engine behind the FLEXS Model / Landscape API.

Only the `get_fitness` hot path of samsinai/FLEXS is rebuilt here (SURVEY.md
section 8): string -> one-hot encode, CNN / MLP / GlobalEpistasis forward,
ensemble reduction, NoisyAbstractModel neighbour search + blend, argmax decode
-- hand-written HIP for gfx950 in libflexs_amd.so, called through a C ABI
(include/flexs_amd.h).  Explorers, landscapes and evaluation drivers remain the
reference's host Python and drop in unchanged.
"""
from flexs_amd import types  # noqa: F401
from flexs_amd.landscape import Landscape  # noqa: F401
from flexs_amd.model import LandscapeAsModel, Model  # noqa: F401
from flexs_amd.ensemble import Ensemble  # noqa: F401  isort:skip
from flexs_amd import baselines, landscapes, utils  # noqa: F401  isort:skip

__version__ = "0.1.0"
