"""TEST INFRASTRUCTURE -- deterministic parameter values for the EquiformerV2 parity fixtures (same role as oracle/escn_params.py)."""
import math
import zlib

import torch


def make_tensor(name, shape, seed):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    r = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if name.endswith(".bias") or name.endswith("affine_bias"):
        return 0.1 * r
    if name.endswith("embedding.weight"):
        return r if name.startswith("sphere_embedding") else 0.5 * r
    if name.endswith("affine_weight") or len(shape) == 1:     # layer-norm scales
        return 1.0 + 0.1 * r
    if name.endswith("alpha_dot"):
        return r / math.sqrt(shape[-1])
    return r / math.sqrt(shape[-1])                           # Linear [out, in] and SO3_LinearV2 [l, out, in]: fan-in scaling


def make_state(named_shapes, seed):
    return {name: make_tensor(name, shape, seed) for name, shape in named_shapes}


def probe_direction(name, shape, seed):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(("probe:" + name).encode()) ^ (seed * 40503)) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float64)
