"""Utility modules (mirrors flexs/utils; `sequence_utils` is the one on the hot path)."""
from flexs_amd.utils import edit_distance, sequence_utils  # noqa: F401
