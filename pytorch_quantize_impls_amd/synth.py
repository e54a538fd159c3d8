"""Build-owned counter PRNG for synthetic inputs (bench, tests, golden vectors).

splitmix64 over a (seed, index) counter: integer-only, vectorised in numpy, identical on every
machine and numpy/torch version — so the GPU box regenerates exactly the inputs the golden
hashes were computed on.  "normal" is an Irwin-Hall(12) sum (mean 0,
variance 1, pure IEEE adds — no libm), which is all the synthetic workloads need.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def bits64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit words of stream ``stream`` of generator ``seed``."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))],
                                    dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) + base
    return _splitmix64(idx)


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1) with 53 random bits."""
    return (bits64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed: int, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(seed, n)).astype(np.float32).reshape(shape)


def normal(seed: int, shape, std: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for s in range(12):
        acc += uniform01(seed, n, stream=s + 1)
    return ((acc - 6.0) * std).astype(np.float32).reshape(shape)


def pm1(seed: int, shape) -> np.ndarray:
    """+-1 fp32 with fair coin signs."""
    n = int(np.prod(shape))
    b = (bits64(seed, n) >> np.uint64(63)).astype(np.float32)
    return (1.0 - 2.0 * b).reshape(shape)
