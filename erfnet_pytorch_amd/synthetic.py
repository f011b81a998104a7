"""Deterministic synthetic weights, images and labels.

The pretrained checkpoints of the reference are absent from its checkout
(/root/reference/.MISSING_LARGE_BLOBS:1-2), so every parity test, the smoke
test and the benchmark run on closed-form weights that any machine can
regenerate bit-for-bit from a state_dict's (key, shape) list alone: pure
integer hashing in numpy, no libm, no torch RNG.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    """32-bit integer finaliser (murmur3-style) on a uint64 carrier."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def unit_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (index, seed)."""
    idx = np.arange(n, dtype=np.uint64)
    h = _mix32(idx * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF))
    h = _mix32(h ^ np.uint64((seed * 0x7FEB352D) & 0xFFFFFFFF))
    return (h.astype(np.float64) / 2147483648.0) - 1.0


def key_seed(key: str, salt: int = 0) -> int:
    return (zlib.crc32(key.encode()) + 0x632BE5AB * (salt + 1)) & 0xFFFFFFFF


def fill_tensor(key: str, shape, salt: int = 0) -> torch.Tensor:
    """Closed-form value for one state_dict entry, chosen by the key's suffix.

    conv / linear weights: uniform(-a, a), a = gain * sqrt(6 / fan_in)
    biases:               uniform(-0.1, 0.1)
    BN weight (gamma):    uniform(0.6, 1.4);  BN bias (beta): uniform(-0.2, 0.2)
    running_mean:         uniform(-0.3, 0.3); running_var:   uniform(0.5, 1.5)
    num_batches_tracked:  0
    """
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit_uniform(n, key_seed(key, salt))
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        v = 0.3 * u
    elif leaf == "running_var":
        v = 1.0 + 0.5 * u
    elif leaf == "weight" and len(shape) == 1:        # BatchNorm gamma
        v = 1.0 + 0.4 * u
    elif leaf == "bias":
        parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
        v = (0.2 if parent.startswith("bn") else 0.1) * u
    elif leaf == "weight" and len(shape) >= 2:
        # Conv2d [Cout,Cin,kh,kw]: fan_in = Cin*kh*kw.  ConvTranspose2d
        # [Cin,Cout,kh,kw] at stride 2: each output pixel sees ~Cin*kh*kw/4 taps.
        fan_in = int(np.prod(shape[1:]))
        is_transposed = (".conv.weight" in key and "decoder.layers" in key and len(shape) == 4
                         and shape[2] == 3 and shape[3] == 3) or key.endswith("decoder.output_conv.weight")
        if is_transposed:
            fan_in = shape[0] * shape[2] * shape[3] // 4
        a = 0.75 * np.sqrt(6.0 / max(fan_in, 1))
        v = a * u
    else:
        v = 0.1 * u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(template: dict, salt: int = 0) -> dict:
    """A new state_dict with the template's keys/shapes and closed-form values."""
    return {k: fill_tensor(k, v.shape, salt) for k, v in template.items()}


def load_synthetic(module: torch.nn.Module, salt: int = 0) -> torch.nn.Module:
    sd = module.state_dict()
    module.load_state_dict(fill_state_dict(sd, salt))
    return module


def synthetic_images(n: int, h: int, w: int, seed: int = 0, c: int = 3) -> torch.Tensor:
    """N(0,1)-ish images (sum of 4 uniforms, unit variance), fp32 NCHW."""
    m = n * c * h * w
    acc = np.zeros(m, dtype=np.float64)
    for k in range(4):
        acc += unit_uniform(m, key_seed("image", seed * 4 + k))
    acc *= np.sqrt(3.0) / 2.0
    return torch.from_numpy(acc.astype(np.float32).reshape(n, c, h, w))


def synthetic_labels(n: int, h: int, w: int, num_classes: int = 20, seed: int = 0,
                     ignore_frac: float = 0.05, ignore_label: int = 19) -> torch.Tensor:
    """int64 labels [N,1,H,W] uniform over the classes, ~ignore_frac forced to ignore_label."""
    m = n * h * w
    u = (unit_uniform(m, key_seed("label", seed)) + 1.0) * 0.5
    lab = np.minimum((u * num_classes).astype(np.int64), num_classes - 1)
    if ignore_frac > 0 and ignore_label < num_classes:
        g = (unit_uniform(m, key_seed("ignore", seed)) + 1.0) * 0.5
        lab[g < ignore_frac] = ignore_label
    return torch.from_numpy(lab.reshape(n, 1, h, w))
