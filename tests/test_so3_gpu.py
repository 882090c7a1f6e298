"""GPU: SO(3) mixing kernels (csrc/so3.hip via nabladft_amd.so3) against golden vectors produced by the REAL reference modules
(phisnet PairMixing / SelfMixing with the reference's Clebsch-Gordan table; oracle/make_golden_phisnet.py).
Tolerance 1e-5 relative (fp32; the reference sums the 5-D broadcast products in a different order)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, rel_err
from tests.so3_helpers import FixtureCG

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("tag", ["pm222", "pm444", "pm214"])
def test_pair_mixing_matches_reference(tag):
    from nabladft_amd import so3
    fx = np.load(os.path.join(GOLDEN, "phisnet_mixing.npz"))
    o1, o2, oy, K, F, rows = (int(v) for v in fx[tag + ":cfg"])
    m = so3.PairMixing(o1, o2, oy, K, F, FixtureCG()).cuda()
    m.load_state_dict({k.split(":p:")[1]: torch.tensor(fx[k]) for k in fx.files if k.startswith(tag + ":p:")})
    x1s = [torch.tensor(fx[f"{tag}:x1_{l}"]).cuda().requires_grad_(True) for l in range(o1 + 1)]
    x2s = [torch.tensor(fx[f"{tag}:x2_{l}"]).cuda().requires_grad_(True) for l in range(o2 + 1)]
    rbf = torch.tensor(fx[f"{tag}:rbf"]).cuda().requires_grad_(True)
    ys = m(x1s, x2s, rbf)
    assert len(ys) == oy + 1
    for L, y in enumerate(ys):
        assert y.shape == fx[f"{tag}:y_{L}"].shape and rel_err(y.detach().cpu().numpy(), fx[f"{tag}:y_{L}"]) < TOL, L
    sum((y * torch.tensor(fx[f"{tag}:w_{L}"]).cuda()).sum() for L, y in enumerate(ys)).backward()
    for l, t in enumerate(x1s):
        assert rel_err(t.grad.cpu().numpy(), fx[f"{tag}:gx1_{l}"]) < TOL, ("gx1", l)
    for l, t in enumerate(x2s):
        assert rel_err(t.grad.cpu().numpy(), fx[f"{tag}:gx2_{l}"]) < TOL, ("gx2", l)
    assert rel_err(rbf.grad.cpu().numpy(), fx[f"{tag}:grbf"]) < TOL
    for n, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx[f"{tag}:g:{n}"]) < TOL, n


@pytest.mark.parametrize("tag", ["sm44", "sm23", "sm31"])
def test_self_mixing_matches_reference(tag):
    from nabladft_amd import so3
    fx = np.load(os.path.join(GOLDEN, "phisnet_mixing.npz"))
    oi, oo, F, rows = (int(v) for v in fx[tag + ":cfg"])
    m = so3.SelfMixing(oi, oo, F, FixtureCG()).cuda()
    m.load_state_dict({k.split(":p:")[1]: torch.tensor(fx[k]) for k in fx.files if k.startswith(tag + ":p:")})
    xs = [torch.tensor(fx[f"{tag}:x_{l}"]).cuda().requires_grad_(True) for l in range(oi + 1)]
    ys = m(xs)
    for L, y in enumerate(ys):
        assert y.shape == fx[f"{tag}:y_{L}"].shape and rel_err(y.detach().cpu().numpy(), fx[f"{tag}:y_{L}"]) < TOL, L
    sum((y * torch.tensor(fx[f"{tag}:w_{L}"]).cuda()).sum() for L, y in enumerate(ys)).backward()
    for l, t in enumerate(xs):
        assert rel_err(t.grad.cpu().numpy(), fx[f"{tag}:gx_{l}"]) < TOL, ("gx", l)
    for n, p in m.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx[f"{tag}:g:{n}"]) < TOL, n


def test_pair_mixing_throughput_on_a_phisnet_sized_call():
    """All ordered pairs of 16 conformers of 42 atoms (27.5 k rows), order 4, F = 128, K = 128: one forward + backward."""
    import time
    from nabladft_amd import so3
    rows, F, K = 16 * 42 * 41, 128, 128
    m = so3.PairMixing(4, 4, 4, K, F, FixtureCG()).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x1s = [torch.randn(1, rows, 2 * l + 1, F, device="cuda", generator=g).requires_grad_(True) for l in range(5)]
    x2s = [torch.randn(1, rows, 2 * l + 1, F, device="cuda", generator=g).requires_grad_(True) for l in range(5)]
    rbf = torch.randn(1, rows, 1, K, device="cuda", generator=g).requires_grad_(True)
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ys = m(x1s, x2s, rbf)
        sum(y.sum() for y in ys).backward()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    flops = rows * F * 2052 * 2 * (1 + 3)          # forward (mul + fma per non-zero) and ~3x that in the reverse kernel
    print(f"PairMixing order 4, {rows} pairs, F={F}, K={K}: fwd+bwd {ms:.2f} ms  (CG contraction ~{flops / ms / 1e9:.1f} TFLOP/s incl. GEMMs and packing)")
    assert all(torch.isfinite(y).all() for y in ys)


def test_spherical_harmonics_and_bernstein_rbf_match_reference():
    """Golden vectors from the real reference code: PhiSNet spherical_harmonics(4, u); ExponentialBernsteinRadialBasisFunctions of PhiSNet
    (K = 128, cutoff 15) and QHNet (K = 32, cutoff 12), values at / beyond the cutoff included, gradient w.r.t. _alpha."""
    from nabladft_amd import so3
    fx = np.load(os.path.join(GOLDEN, "geometry_bases.npz"))
    ys = so3.spherical_harmonics(4, torch.tensor(fx["u"]).cuda())
    for l, y in enumerate(ys):
        assert y.shape == fx[f"Y_{l}"].shape and np.abs(y.cpu().numpy() - fx[f"Y_{l}"]).max() < 5e-6, l
    for tag in ("phisnet128", "qhnet32", "small"):
        K, cutoff, ini = fx[f"{tag}:cfg"]
        m = so3.ExponentialBernsteinRadialBasisFunctions(int(K), float(cutoff), float(ini)).cuda()
        assert abs(float(m._alpha) - float(fx[f"{tag}:_alpha"])) < 1e-6 and np.allclose(m.logc.cpu().numpy(), fx[f"{tag}:logc"], rtol=1e-6, atol=1e-5)
        out = m(torch.tensor(fx[f"{tag}:r"]).cuda())
        assert out.shape == fx[f"{tag}:rbf"].shape
        assert rel_err(out.detach().cpu().numpy(), fx[f"{tag}:rbf"]) < 2e-5, tag
        assert float(out[-1].abs().max()) == 0.0 and float(out[-2].abs().max()) == 0.0          # r >= cutoff -> exactly 0
        (out * torch.tensor(fx[f"{tag}:w"]).cuda()).sum().backward()
        assert abs(float(m._alpha.grad) - float(fx[f"{tag}:g_alpha"])) < 5e-5 * max(1.0, abs(float(fx[f"{tag}:g_alpha"]))), tag


def test_other_radial_bases_match_reference():
    """gaussian / exp-gaussian / bernstein (the other choices of NeuralNetwork, neural_network.py:210-221) and overlap-bernstein: values, exact zeros at /
    beyond the cutoff, dL/d_alpha and the state_dict surface against the real reference modules."""
    from nabladft_amd import so3
    fx = np.load(os.path.join(GOLDEN, "geometry_bases.npz"))
    for tag in ("gaussian", "exp-gaussian", "overlap-bernstein", "bernstein"):
        args = fx[f"rb:{tag}:args"]
        m = so3.RADIAL_BASES[tag](int(args[0]), *[float(a) for a in args[1:]]).cuda()
        assert list(m.state_dict().keys()) == list(fx[f"rb:{tag}:state_keys"]), tag
        out = m(torch.tensor(fx[f"rb:{tag}:r"]).cuda())
        ref = fx[f"rb:{tag}:rbf"]
        inside = ~np.isnan(ref).any(axis=-1)       # the reference's plain Bernstein basis is 0 * NaN = NaN at r >= cutoff (log of a negative number); 0 here
        assert inside.sum() >= 30 and (tag == "bernstein" or inside.all())
        assert out.shape == ref.shape and rel_err(out.detach().cpu().numpy()[inside], ref[inside]) < 1e-5, tag
        assert float(out[-1].abs().max()) == 0.0 and float(out[-2].abs().max()) == 0.0
        if tag in ("exp-gaussian", "overlap-bernstein"):
            (out * torch.tensor(fx[f"rb:{tag}:w"]).cuda()).sum().backward()
            ref = float(fx[f"rb:{tag}:g_alpha"])
            assert abs(float(m._alpha.grad) - ref) < 2e-5 * max(1.0, abs(ref)), (tag, float(m._alpha.grad), ref)
