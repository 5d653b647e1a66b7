"""Baselines: device-backed surrogate models (explorers stay the reference's own)."""
from flexs_amd.baselines import models  # noqa: F401
