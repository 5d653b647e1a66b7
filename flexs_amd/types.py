"""Type definitions (mirrors flexs/types.py:6)."""
from typing import List, Union

import numpy as np

SEQUENCES_TYPE = Union[List[str], np.ndarray]
