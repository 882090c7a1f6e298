"""TEST INFRASTRUCTURE -- deterministic parameter values for the GemNet-OC parity fixtures (the role oracle/qhnet_params.py plays for QHNet).

The config/model/gemnet-oc.yaml network has 37.8 M parameters; the fixture generator (oracle/make_golden_gemnet.py, container-only, loads them into
the REAL reference model) and the tests (load them into nabladft_amd.gemnet_oc.GemNetOC) both call ``make_state`` with the same names / shapes / seed.
Scales follow each tensor's initialiser (gemnet_oc/initializers.py:26-45: variance 1/fan_in, fan_in = in_features or, for the 3-index BasisEmbedding
weights, the product of the first two extents; embeddings: unit variance).  ScaleFactor parameters (scale_factor.py:37-58; 0 = "not fitted" = identity)
are set to non-trivial values only when ``fit_scales`` is given.
"""
import math
import zlib

import torch


def make_tensor(name, shape, seed, fit_scales=False):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if name.endswith("scale_factor"):
        if not fit_scales:
            return torch.tensor(0.0)
        return torch.tensor(0.75 + 0.5 * float(torch.rand((), generator=g, dtype=torch.float32)), dtype=torch.float32)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if name.endswith("embeddings.weight"):
        return r
    if len(shape) == 3:
        return r / math.sqrt(shape[0] * shape[1])
    return 0.8 * r / math.sqrt(shape[1])          # 0.8: keeps the 100-layer random network from amplifying (the reference relies on fitted ScaleFactors for that)


def make_state(named_shapes, seed, fit_scales=False):
    return {name: make_tensor(name, shape, seed, fit_scales) for name, shape in named_shapes}


def probe_direction(name, shape, seed):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(("probe:" + name).encode()) ^ (seed * 40503)) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float64)
