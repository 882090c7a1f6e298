"""TEST INFRASTRUCTURE -- deterministic parameter values for the eSCN parity fixtures (same role as oracle/gemnet_params.py)."""
import math
import zlib

import torch


def make_tensor(name, shape, seed):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    r = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if name.endswith(".bias"):
        return 0.1 * r
    if name.endswith("embedding.weight"):
        return r if name.startswith("sphere_embedding") else 0.5 * r
    return r / math.sqrt(shape[1])


def make_state(named_shapes, seed):
    return {name: make_tensor(name, shape, seed) for name, shape in named_shapes}


def probe_direction(name, shape, seed):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(("probe:" + name).encode()) ^ (seed * 40503)) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float64)
