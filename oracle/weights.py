"""Deterministic synthetic weights (test infrastructure).

No pretrained checkpoint is available offline, and with default init the reference's UNet output is exactly 0
(zero_module on every ResBlock out-conv, transformer proj_out, TemporalConvBlock.conv4, fps_embedding[-1], final
out conv: openaimodel3d.py:179,269-270,381-382,545; attention.py:288-290,360-362). Fixtures therefore fill EVERY
tensor from a per-name seeded generator, so that (a) zero-initialised tensors become non-trivial, (b) the same
name+shape gives the same tensor on the reference side (make_golden.py) and on the build side (tests), with no
dependence on parameter order and without committing gigabytes of weights.
"""
import zlib

import torch


def tensor_for(name, shape, seed=0):
    """Value of parameter `name` with `shape`: N(0, s^2) with s chosen by role so activations stay O(1)."""
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) == 0:                       # learnable scalar (CrossAttention.alpha)
        return torch.randn((), generator=g) * 0.5
    if len(shape) == 1:
        if name.endswith("bias"):
            return torch.randn(shape, generator=g) * 0.05
        return 1.0 + 0.1 * torch.randn(shape, generator=g)     # norm scales
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if name.endswith("latents"):              # Resampler.latents [1, n, d]
        return torch.randn(shape, generator=g) * (shape[-1] ** -0.5)
    return torch.randn(shape, generator=g) * (0.8 / fan_in ** 0.5)


def fill_state_dict(shapes, seed=0, dtype=torch.float32):
    """shapes: mapping name -> shape. Returns name -> tensor (order-independent)."""
    return {k: tensor_for(k, s, seed).to(dtype) for k, s in shapes.items()}
