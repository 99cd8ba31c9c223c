"""Deterministic synthetic data: weights, inputs and noise by *recipe*, not by file.

There are no checkpoints or datasets on the build/GPU boxes, and a random-init reference UNet
outputs exactly 0 (every ResBlock out-conv, attention proj_out and the head conv are
`zero_module`'d, MToV/models/ddpm/unet.py:159,242,289,974).  So every tensor -- including the
zero-initialised ones and the GroupNorm affine parameters -- is filled from a pure-arithmetic
counter hash keyed on (state-dict key, flat index).  The same recipe runs in the golden-vector
generator (with the reference), in the tests (oracle and HIP path) and in bench.py, so 0.5 GB of
weights never has to be stored and nothing depends on a torch RNG stream.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24 random bits each (exact in float32)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15) ^ (stream * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(base + idx * np.uint64(0x2545F4914F6CDD1D))
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def uniform_pm1(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(key, n, seed)
    return torch.from_numpy((2.0 * u - 1.0).astype(np.float32)).reshape(shape)


def normal(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """N(0,1) by Box-Muller on two hashed uniform streams (float64 math, float32 result)."""
    n = int(np.prod(shape))
    u1 = uniform01(key, n, seed, stream=1)
    u2 = uniform01(key, n, seed, stream=2)
    u1 = (u1 * (1 << 24) + 1.0) / float((1 << 24) + 1)     # (0,1]: log is finite
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy(z.astype(np.float32)).reshape(shape)


def fill_tensor(key: str, shape: Tuple[int, ...], seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    """Weight recipe.
      >=2-D (conv / conv1d / linear weight): U(-1,1) * gain * sqrt(3 / fan_in)   (unit-variance preserving)
      1-D '...weight' (GroupNorm gamma)    : 1 + 0.2 * U(-1,1)
      1-D '...bias'                        : 0.2 * U(-1,1)
    """
    u = uniform_pm1(key, tuple(shape), seed)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return u * float(gain * math.sqrt(3.0 / fan_in))
    if key.endswith("weight"):
        return 1.0 + 0.2 * u
    return 0.2 * u


def fill_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    return {k: fill_tensor(k, tuple(s), seed, gain) for k, s in shapes}


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0, gain: float = 1.0, skip_prefixes: Tuple[str, ...] = ()) -> None:
    """Overwrite every parameter/buffer-backed state_dict entry of `module` in place with the recipe."""
    sd = module.state_dict()
    for k, v in sd.items():
        if not torch.is_floating_point(v) or any(k.startswith(p) or ("." + p) in k for p in skip_prefixes):
            continue
        v.copy_(fill_tensor(k, tuple(v.shape), seed, gain).to(v.device, v.dtype))


# ViTAutoencoder (the steps either side of the denoising loop): the recipe above applied to every parameter -- the
# rotary-frequency buffers keep the values the reference computes -- with two per-key gains so that the sigmoid / tanh
# output layers stay in their sensitive range (at gain 1 a random 8-layer transformer saturates them, and a saturated
# output would hide errors from a parity test).
AE_BUFFERS = ("inv_freqs", "scales", "coords")
AE_KEY_GAINS = {"to_pixel.1.weight": 0.1, "pre_xy.weight": 0.1, "pre_yt.weight": 0.1, "pre_xt.weight": 0.1}


@torch.no_grad()
def fill_autoencoder_(module: torch.nn.Module, seed: int = 0) -> None:
    fill_module_(module, seed=seed, skip_prefixes=AE_BUFFERS)
    sd = module.state_dict()
    for k, g in AE_KEY_GAINS.items():
        if k in sd:
            sd[k].mul_(g)


def synthetic_inputs(batch: int, res: int, frames: int, seed: int = 0, tag: str = "clip"):
    """x [B,4,L], cond [B,8,L], image_cond [B,4,R*R], all U(-1,1) (AE latents are tanh outputs)."""
    L = res * res + 2 * frames * res
    x = uniform_pm1(f"{tag}.x", (batch, 4, L), seed)
    cond = uniform_pm1(f"{tag}.cond", (batch, 8, L), seed)
    image_cond = uniform_pm1(f"{tag}.image_cond", (batch, 4, res * res), seed)
    return x, cond, image_cond


def noise_list(n: int, shape: Tuple[int, ...], seed: int = 0, tag: str = "noise"):
    return [normal(f"{tag}.{i}", tuple(shape), seed) for i in range(n)]
