"""Platform-independent deterministic pseudo-random tensors.

Golden fixtures store only outputs; inputs and parameters are regenerated from
(tag, shape) with exact integer arithmetic so that the authoring container and
the GPU box produce bit-identical float32 values without shipping megabytes of
parameters.
"""
import zlib

import numpy as np
import torch


def _u01(tag: str, n: int) -> np.ndarray:
    """n float64 values in [0,1), exact: a 32-bit multiplicative hash / 2^32."""
    seed = np.uint64(zlib.crc32(tag.encode()) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    x = (idx + seed * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    # three rounds of an integer mixer (all ops exact in uint64, masked to 32 bit)
    for mul in (0x85EBCA6B, 0xC2B2AE35, 0x27D4EB2F):
        x ^= x >> np.uint64(15)
        x = (x * np.uint64(mul)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(13)
    return x.astype(np.float64) / 4294967296.0


def uniform(tag: str, shape, lo=0.0, hi=1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    v = lo + (hi - lo) * _u01(tag, n)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def triangular(tag: str, shape, scale=1.0) -> torch.Tensor:
    """Zero-mean, roughly bell-shaped (sum of 4 uniforms), exact arithmetic only."""
    n = int(np.prod(shape))
    v = sum(_u01(f"{tag}/{k}", n) for k in range(4)) - 2.0
    return torch.from_numpy((v * (scale * 1.7320508)).astype(np.float32).reshape(shape))
