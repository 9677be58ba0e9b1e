"""Deterministic synthetic inputs shared by the golden generator and the tests.

Integer-hash based so that every platform reproduces the same float32 SE tiles without
storing them in the fixtures (only IEEE add/mul/convert are used; no libm calls).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_u01(seed: int, idx: np.ndarray) -> np.ndarray:
    """Uniform [0,1) float64 from (seed, integer index array)."""
    with np.errstate(over="ignore"):
        c = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx.astype(np.uint64)) & _M64
    z = _splitmix64(_splitmix64(c))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def se_tile(seed: int, t: int, n_ues: int, n_rbs: int, low_se_every: int = 0) -> np.ndarray:
    """(U,R) float32 spectral efficiencies following the MimicQuadriga law
    (channels/mimic_quadriga.py:37-56): per-UE mean in ~[1,19], per-RB spread ~1.5,
    absolute value.  Pseudo-normal = sum of 4 uniforms (exactly reproducible)."""
    u = np.arange(n_ues, dtype=np.uint64)
    mu = 1.0 + 18.0 * hash_u01(seed * 7919 + 1, u)                                   # (U,)
    idx = (np.uint64(t) * np.uint64(n_ues * n_rbs)
           + np.arange(n_ues * n_rbs, dtype=np.uint64)).reshape(n_ues, n_rbs)
    z = np.zeros((n_ues, n_rbs))
    for k in range(4):
        z += hash_u01(seed * 104729 + 17 + k, idx)
    z = (z - 2.0) * 1.7320508075688772                                               # var 1
    se = np.abs(mu[:, None] + 1.5 * z)
    if low_se_every:
        se[::low_se_every] *= 0.02
    return se.astype(np.float32)


def se_pool(seed: int, n_tiles: int, n_ues: int, n_rbs: int) -> np.ndarray:
    return np.stack([se_tile(seed, t, n_ues, n_rbs) for t in range(n_tiles)])
