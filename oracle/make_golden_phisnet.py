"""TEST INFRASTRUCTURE (container-only): golden vectors for the SO(3) mixing modules, produced by the REAL reference modules
nablaDFT/phisnet/nn/modules/{pair_mixing,self_mixing,clebsch_gordan}.py (pure torch; loaded file by file, the package __init__ chain
is not executed).  Writes tests/golden/phisnet_mixing.npz (inputs, seeded parameters, outputs, gradients) and
tests/golden/phisnet_cg_l4.npz (the reference's Clebsch-Gordan table restricted to l <= 4: data held by the reference's module).
    python oracle/make_golden_phisnet.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
MOD = "/root/reference/nablaDFT/phisnet/nn/modules"


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_phisnet_" + name, os.path.join(MOD, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    torch.manual_seed(0)
    CG = _load("clebsch_gordan").ClebschGordan()
    PairMixing = _load("pair_mixing").PairMixing
    SelfMixing = _load("self_mixing").SelfMixing
    raw = np.load(os.path.join(MOD, "clebsch_gordan_coefficients_L10.npz"), allow_pickle=True)["cg"][()]
    np.savez_compressed(os.path.join(OUT, "phisnet_cg_l4.npz"), **{"cg_%d_%d_%d" % k: v for k, v in raw.items() if max(k) <= 4})
    fx = {}
    rng = np.random.Generator(np.random.PCG64(3))
    for tag, (o1, o2, oy, K, F, rows) in {"pm222": (2, 2, 2, 8, 64, 10), "pm444": (4, 4, 4, 16, 64, 6), "pm214": (2, 1, 3, 8, 64, 5)}.items():
        m = PairMixing(o1, o2, oy, K, F, CG)
        x1s = [torch.tensor(rng.normal(size=(1, rows, 2 * l + 1, F)).astype(np.float32), requires_grad=True) for l in range(o1 + 1)]
        x2s = [torch.tensor(rng.normal(size=(1, rows, 2 * l + 1, F)).astype(np.float32), requires_grad=True) for l in range(o2 + 1)]
        rbf = torch.tensor(rng.normal(size=(1, rows, 1, K)).astype(np.float32), requires_grad=True)
        ys = m(x1s, x2s, rbf)
        ws = [torch.tensor(rng.normal(size=tuple(y.shape)).astype(np.float32)) for y in ys]
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        fx[tag + ":cfg"] = np.array([o1, o2, oy, K, F, rows])
        for l, t in enumerate(x1s):
            fx[f"{tag}:x1_{l}"], fx[f"{tag}:gx1_{l}"] = t.detach().numpy(), t.grad.numpy()
        for l, t in enumerate(x2s):
            fx[f"{tag}:x2_{l}"], fx[f"{tag}:gx2_{l}"] = t.detach().numpy(), t.grad.numpy()
        fx[f"{tag}:rbf"], fx[f"{tag}:grbf"] = rbf.detach().numpy(), rbf.grad.numpy()
        for L, (y, w) in enumerate(zip(ys, ws)):
            fx[f"{tag}:y_{L}"], fx[f"{tag}:w_{L}"] = y.detach().numpy(), w.numpy()
        for n, p in m.named_parameters():
            fx[f"{tag}:p:{n}"], fx[f"{tag}:g:{n}"] = p.detach().numpy(), p.grad.numpy()
    for tag, (oi, oo, F, rows) in {"sm44": (4, 4, 64, 7), "sm23": (2, 3, 64, 5), "sm31": (3, 1, 64, 4)}.items():
        m = SelfMixing(oi, oo, F, CG)
        xs = [torch.tensor(rng.normal(size=(1, rows, 2 * l + 1, F)).astype(np.float32), requires_grad=True) for l in range(oi + 1)]
        ys = m(xs)
        ws = [torch.tensor(rng.normal(size=tuple(y.shape)).astype(np.float32)) for y in ys]
        sum((y * w).sum() for y, w in zip(ys, ws)).backward()
        fx[tag + ":cfg"] = np.array([oi, oo, F, rows])
        for l, t in enumerate(xs):
            fx[f"{tag}:x_{l}"], fx[f"{tag}:gx_{l}"] = t.detach().numpy(), t.grad.numpy()
        for L, (y, w) in enumerate(zip(ys, ws)):
            fx[f"{tag}:y_{L}"], fx[f"{tag}:w_{L}"] = y.detach().numpy(), w.numpy()
        for n, p in m.named_parameters():
            fx[f"{tag}:p:{n}"], fx[f"{tag}:g:{n}"] = p.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "phisnet_mixing.npz"), **fx)
    print("phisnet_mixing.npz:", len(fx), "arrays")


if __name__ == "__main__" and "--bases" not in sys.argv and "--blocks" not in sys.argv:
    main()


def geometry_bases():
    """Spherical harmonics and exponential-Bernstein radial bases from the real reference code (PhiSNet: fp64 buffers cast with .float();
    QHNet's fp32 copy of the same class via oracle/qhnet_import.py)."""
    sys.path.insert(0, os.path.dirname(HERE))
    import types
    # phisnet/nn/spherical_harmonics is a plain package of two files
    pkg = types.ModuleType("ref_sh")
    pkg.__path__ = ["/root/reference/nablaDFT/phisnet/nn/spherical_harmonics"]
    sys.modules["ref_sh"] = pkg
    sh = importlib.import_module("ref_sh.spherical_harmonics")
    fpkg = types.ModuleType("ref_phisnet_nn")
    fpkg.__path__ = ["/root/reference/nablaDFT/phisnet/nn"]
    sys.modules["ref_phisnet_nn"] = fpkg
    mpkg = types.ModuleType("ref_phisnet_nn.modules")
    mpkg.__path__ = [MOD]
    sys.modules["ref_phisnet_nn.modules"] = mpkg
    rbf_mod = importlib.import_module("ref_phisnet_nn.modules.exponential_bernstein_radial_basis_functions")
    from oracle.qhnet_import import load_qhnet
    qlayers = sys.modules.get("nablaDFT.qhnet.layers") or (load_qhnet() and sys.modules["nablaDFT.qhnet.layers"])
    rng = np.random.Generator(np.random.PCG64(17))
    fx = {}
    u = rng.normal(size=(37, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    u[0] = [0, 0, 1]
    ut = torch.tensor(u.astype(np.float32))
    fx["u"] = ut.numpy()
    for l, y in enumerate(sh.spherical_harmonics(4, ut)):
        fx[f"Y_{l}"] = y.numpy()
    for tag, (cls, K, cutoff, ini) in {"phisnet128": (rbf_mod.ExponentialBernsteinRadialBasisFunctions, 128, 15.0, 0.5),
                                      "qhnet32": (qlayers.ExponentialBernsteinRadialBasisFunctions, 32, 12.0, 0.5),
                                      "small": (qlayers.ExponentialBernsteinRadialBasisFunctions, 8, 5.0, 1.3)}.items():
        m = cls(K, cutoff, ini).float()
        r = torch.tensor(np.concatenate([rng.uniform(0.3, cutoff * 0.999, size=29), [cutoff * 0.9999, cutoff, cutoff * 1.2]]).astype(np.float32)).view(-1, 1)
        out = m(r)
        w = torch.tensor(rng.normal(size=tuple(out.shape)).astype(np.float32))
        (out * w).sum().backward()
        fx[f"{tag}:cfg"] = np.array([K, cutoff, ini])
        fx[f"{tag}:r"], fx[f"{tag}:rbf"], fx[f"{tag}:w"] = r.numpy(), out.detach().numpy(), w.numpy()
        fx[f"{tag}:g_alpha"], fx[f"{tag}:_alpha"] = m._alpha.grad.numpy(), m._alpha.detach().numpy()
        fx[f"{tag}:logc"] = m.logc.numpy()
    np.savez_compressed(os.path.join(OUT, "geometry_bases.npz"), **fx)
    print("geometry_bases.npz:", len(fx), "arrays")


if __name__ == "__main__" and "--bases" in sys.argv:
    geometry_bases()


def blocks():
    """ModularBlock (residual stacks + InteractionBlock) of the real reference on a small full graph, seeded non-trivial parameters (the
    reference zero-initialises the second linear of every residual block, which would hide half of the graph), outputs + all gradients."""
    mpkg = __import__("types").ModuleType("ref_pm")
    mpkg.__path__ = [MOD]
    sys.modules["ref_pm"] = mpkg
    mb = importlib.import_module("ref_pm.modular_block")
    CG = importlib.import_module("ref_pm.clebsch_gordan").ClebschGordan().float()
    torch.manual_seed(1)
    rng = np.random.Generator(np.random.PCG64(23))
    fx = {}
    for tag, (order, F, K, N, act) in {"mb2": (2, 32, 8, 5, "swish"), "mb1ssp": (1, 32, 6, 4, "ssp")}.items():
        m = mb.ModularBlock(order, F, K, 1, 1, 1, 1, 1, 1, CG, True, act)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.abs().max() == 0 or n.endswith("alpha") or n.endswith("beta"):
                    p.add_(torch.tensor(rng.normal(0, 0.3, size=tuple(p.shape)).astype(np.float32)))
        ii, jj = zip(*[(i, j) for i in range(N) for j in range(N) if i != j])
        idx_i, idx_j = torch.tensor(ii), torch.tensor(jj)
        P = len(ii)
        xs = [torch.tensor(rng.normal(0, 0.7, size=(1, N, 2 * l + 1, F)).astype(np.float32), requires_grad=True) for l in range(order + 1)]
        rbf = torch.tensor(rng.uniform(0, 1, size=(1, P, 1, K)).astype(np.float32), requires_grad=True)
        sph = [torch.tensor(rng.normal(size=(1, P, 2 * l + 1, 1)).astype(np.float32)) for l in range(order + 1)]
        xo, yo = m(xs, rbf, sph, idx_i, idx_j)
        ws = [torch.tensor(rng.normal(size=tuple(t.shape)).astype(np.float32)) for t in xo + yo]
        sum((t * w).sum() for t, w in zip(xo + yo, ws)).backward()
        fx[tag + ":cfg"] = np.array([order, F, K, N, 0 if act == "swish" else 1])
        fx[tag + ":idx_i"], fx[tag + ":idx_j"] = idx_i.numpy(), idx_j.numpy()
        for l in range(order + 1):
            fx[f"{tag}:x_{l}"], fx[f"{tag}:gx_{l}"], fx[f"{tag}:sph_{l}"] = xs[l].detach().numpy(), xs[l].grad.numpy(), sph[l].numpy()
            fx[f"{tag}:xo_{l}"], fx[f"{tag}:yo_{l}"] = xo[l].detach().numpy(), yo[l].detach().numpy()
            fx[f"{tag}:wx_{l}"], fx[f"{tag}:wy_{l}"] = ws[l].numpy(), ws[order + 1 + l].numpy()
        fx[tag + ":rbf"], fx[tag + ":grbf"] = rbf.detach().numpy(), rbf.grad.numpy()
        for n, p in m.named_parameters():
            fx[f"{tag}:p:{n}"], fx[f"{tag}:g:{n}"] = p.detach().numpy(), p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "phisnet_blocks.npz"), **fx)
    print("phisnet_blocks.npz:", len(fx), "arrays")


if __name__ == "__main__" and "--blocks" in sys.argv:
    blocks()
