"""Portable synthetic-input generator (no torch RNG, bit-identical everywhere).

SURVEY.md section 8(d): values in [-1, 1) from a counter hash so that this
container (where the reference can be imported to make golden vectors) and the
GPU box (where it cannot) regenerate exactly the same tensors.

    value(seed, i) = top24(splitmix64(seed * 2^40 + i)) / 2^23 - 1

Every value has at most 24 significant bits, so it is exact in float32.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0,
                 dtype=np.float32) -> np.ndarray:
    """Deterministic array in [lo, hi); default [-1, 1) with 24-bit mantissas."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) << np.uint64(40)) + idx
    top = (_splitmix64(key) >> np.uint64(40)).astype(np.float64)  # 24 bits
    unit = top / float(1 << 23) - 1.0                              # [-1, 1)
    if lo != -1.0 or hi != 1.0:
        unit = (unit + 1.0) * (0.5 * (hi - lo)) + lo
    return unit.astype(dtype).reshape(shape)


# HRNetV2-W32 feature pyramid seen by the flow head at 1024x512 input
# (SURVEY.md section 3c): (C, H, W), low resolution first.
W32_PYRAMID_1024x512 = ((256, 16, 32), (128, 32, 64), (64, 64, 128),
                        (32, 128, 256))


def pyramid_shapes(width: int = 1024, height: int = 512, base_ch: int = 32):
    """(C,H,W) per level, coarse to fine, for an HRNetV2-W<base_ch> backbone
    (strides 32,16,8,4; channels 8w,4w,2w,w)."""
    return tuple((base_ch * (8 >> lvl), height // (32 >> lvl),
                  width // (32 >> lvl)) for lvl in range(4))


def fill_parameters(module, seed: int = 1000) -> None:
    """Deterministic, framework-RNG-free weights: parameter i (in named_parameters order) is
    hash_uniform(seed + i) scaled by 0.7/sqrt(fan_in) (weights) or 0.05 (biases).  Used to put
    identical weights into the reference's modules (in-container, for goldens) and into this
    repository's counterparts (anywhere)."""
    import math
    import torch
    for i, (_, p) in enumerate(module.named_parameters()):
        if p.dim() > 1:
            fan_in = int(np.prod(p.shape[1:]))
            scale = 0.7 / math.sqrt(fan_in)
        else:
            scale = 0.05
        with torch.no_grad():
            p.copy_(torch.from_numpy(hash_uniform(tuple(p.shape), seed + i) * scale))
