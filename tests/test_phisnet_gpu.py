"""GPU: PhiSNet block mirrors (nabladft_amd/phisnet.py: residual stacks, SphericalLinear, InteractionBlock, ModularBlock) against golden
vectors from the REAL reference ModularBlock (oracle/make_golden_phisnet.py --blocks).  Tolerance 5e-5 relative (fp32, deep composition)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, rel_err
from tests.so3_helpers import FixtureCG

pytestmark = pytest.mark.gpu
TOL = 5e-5


@pytest.mark.parametrize("tag", ["mb2", "mb1ssp"])
def test_modular_block_matches_reference(tag):
    from nabladft_amd import phisnet as PH
    fx = np.load(os.path.join(GOLDEN, "phisnet_blocks.npz"))
    order, F, K, N, act = (int(v) for v in fx[tag + ":cfg"])
    m = PH.ModularBlock(order, F, K, 1, 1, 1, 1, 1, 1, FixtureCG(), True, "swish" if act == 0 else "ssp").cuda()
    ref_sd = {k.split(":p:")[1]: torch.tensor(fx[k]) for k in fx.files if k.startswith(tag + ":p:")}
    assert sorted(n for n, _ in m.named_parameters()) == sorted(ref_sd)                    # same parameter surface as the reference block
    m.load_state_dict(ref_sd)
    xs = [torch.tensor(fx[f"{tag}:x_{l}"]).cuda().requires_grad_(True) for l in range(order + 1)]
    sph = [torch.tensor(fx[f"{tag}:sph_{l}"]).cuda() for l in range(order + 1)]
    rbf = torch.tensor(fx[tag + ":rbf"]).cuda().requires_grad_(True)
    idx_i, idx_j = torch.tensor(fx[tag + ":idx_i"]).cuda(), torch.tensor(fx[tag + ":idx_j"]).cuda()
    xo, yo = m(xs, rbf, sph, idx_i, idx_j)
    for l in range(order + 1):
        assert rel_err(xo[l].detach().cpu().numpy(), fx[f"{tag}:xo_{l}"]) < TOL and rel_err(yo[l].detach().cpu().numpy(), fx[f"{tag}:yo_{l}"]) < TOL, l
    loss = sum((t * torch.tensor(fx[f"{tag}:wx_{l}"]).cuda()).sum() for l, t in enumerate(xo)) + \
        sum((t * torch.tensor(fx[f"{tag}:wy_{l}"]).cuda()).sum() for l, t in enumerate(yo))
    loss.backward()
    for l in range(order + 1):
        assert rel_err(xs[l].grad.cpu().numpy(), fx[f"{tag}:gx_{l}"]) < TOL, ("gx", l)
    assert rel_err(rbf.grad.cpu().numpy(), fx[tag + ":grbf"]) < TOL
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        g = fx[f"{tag}:g:{n}"]
        e = float(np.abs(p.grad.cpu().numpy().astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-6))
        worst = max(worst, (n, e), key=lambda t: t[1])
        assert e < 4 * TOL, (n, e)
    print(f"{tag}: {len(ref_sd)} parameters, worst gradient error {worst[1]:.2e} ({worst[0]})")
