"""Deterministic stand-ins shared by the fixture generators (tests/golden/make_golden_explorers.py) and the tests
that replay their output: a fitness that is a pure function of the sequence text."""
import numpy as np


def hashed_fitness(sequence: str, salt: int = 0) -> float:
    """FNV-1a of the text, folded into [0, 1): the same value in the generator and in the replay."""
    h = (2166136261 ^ (salt * 0x9E3779B1)) & 0xFFFFFFFF
    for ch in str(sequence).encode("latin-1"):
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return ((h >> 8) & 0xFFFFFF) / float(1 << 24)


def hashed_fitnesses(sequences, salt: int = 0) -> np.ndarray:
    return np.array([hashed_fitness(s, salt) for s in sequences], dtype=np.float64)
