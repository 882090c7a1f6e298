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


if __name__ == "__main__" and not ({"--bases", "--blocks", "--matrix", "--network", "--forces", "--checkpoint"} & set(sys.argv)):
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
    # the other radial bases NeuralNetwork can be built with (neural_network.py:210-221) + the overlap variant, from the real reference modules
    others = {"gaussian": ("gaussian_radial_basis_functions", "GaussianRadialBasisFunctions", (16, 6.0)),
              "exp-gaussian": ("exponential_gaussian_radial_basis_functions", "ExponentialGaussianRadialBasisFunctions", (16, 6.0, 0.7)),
              "overlap-bernstein": ("overlap_bernstein_radial_basis_functions", "OverlapBernsteinRadialBasisFunctions", (12, 7.0, 0.9)),
              "bernstein": ("bernstein_radial_basis_functions", "BernsteinRadialBasisFunctions", (12, 5.0))}
    for tag, (modname, clsname, args) in others.items():
        cls = getattr(importlib.import_module("ref_phisnet_nn.modules." + modname), clsname)
        m = cls(*args).float()
        cutoff = args[1]
        r = torch.tensor(np.concatenate([rng.uniform(0.3, cutoff * 0.999, size=29), [cutoff * 0.9999, cutoff, cutoff * 1.2]]).astype(np.float32)).view(-1, 1)
        out = m(r)
        w = torch.tensor(rng.normal(size=tuple(out.shape)).astype(np.float32))
        if out.requires_grad:
            (out * w).sum().backward()
        fx[f"rb:{tag}:args"] = np.array(args, dtype=np.float64)
        fx[f"rb:{tag}:r"], fx[f"rb:{tag}:rbf"], fx[f"rb:{tag}:w"] = r.numpy(), out.detach().numpy(), w.numpy()
        has_alpha = hasattr(m, "_alpha")
        fx[f"rb:{tag}:g_alpha"] = np.float64(m._alpha.grad.item() if has_alpha and m._alpha.grad is not None else 0.0)
        fx[f"rb:{tag}:state_keys"] = np.array(list(m.state_dict().keys()))
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


def matrix_assembly():
    """irreps -> matrix with the REAL NeuralNetwork.compute_matrix_irreps / matrix_block / generate_matrix_from_irreps (called unbound on a
    stand-in ``self`` that carries only the real ClebschGordan table); the irreps-collection loops of forward (neural_network.py:859-918) are
    inline code there and are restated here (which feature row / index an irrep is read from)."""
    import types
    from collections import defaultdict
    pkg = types.ModuleType("ref_phisnet_nn3")
    pkg.__path__ = ["/root/reference/nablaDFT/phisnet/nn"]
    sys.modules["ref_phisnet_nn3"] = pkg
    NN = importlib.import_module("ref_phisnet_nn3.neural_network").NeuralNetwork
    CG = importlib.import_module("ref_phisnet_nn3.modules.clebsch_gordan").ClebschGordan().float()
    fake = types.SimpleNamespace(clebsch_gordan=CG)
    fake.matrix_block = types.MethodType(NN.matrix_block, fake)
    rng = np.random.Generator(np.random.PCG64(31))
    atom2orb = {1: ((1, 0), (1, 0), (1, 1)), 6: ((6, 0), (6, 0), (6, 0), (6, 1), (6, 1), (6, 2)), 8: ((8, 0), (8, 0), (8, 0), (8, 1), (8, 1), (8, 2))}
    elements = sorted(atom2orb)
    # index dictionaries exactly as NeuralNetwork.__init__ builds them (neural_network.py:368-417) from the per-element orbital lists
    number_L = [0] * 5
    irreps_ii = {}
    for zz in elements:
        irreps_ii, number_L = NN.compute_matrix_irreps(atom2orb[zz], atom2orb[zz], irreps_ii, number_L)
    n_ii = max(number_L)
    number_L = [0] * 5
    irreps_ij = {}
    for za in elements:
        for zb in elements:
            irreps_ij, number_L = NN.compute_matrix_irreps(atom2orb[za], atom2orb[zb], irreps_ij, number_L)    # incl. same-element pairs
    n_ij = max(number_L)
    Fo = max(n_ii, n_ij)
    sizes = [3, 1, 4]
    zs = np.array([8, 1, 1, 6, 6, 1, 8, 1])
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    N = int(ptr[-1])
    orbitals = [atom2orb[int(a)] for a in zs]
    idx_i, idx_j = [], []
    for b in range(len(sizes)):
        for i in range(ptr[b], ptr[b + 1]):
            for j in range(ptr[b], ptr[b + 1]):
                if i != j:
                    idx_i.append(i), idx_j.append(j)
    P = len(idx_i)
    f_ii = [torch.tensor(rng.normal(size=(1, N, 2 * L + 1, Fo)).astype(np.float32), requires_grad=True) for L in range(5)]
    f_ij = [torch.tensor(rng.normal(size=(1, P, 2 * L + 1, Fo)).astype(np.float32), requires_grad=True) for L in range(5)]
    begin, end, Norb = {}, {}, 0
    for i in range(N):
        begin[i] = Norb
        Norb += sum(2 * l + 1 for _, l in orbitals[i])
        end[i] = Norb
    mask = torch.block_diag(*(torch.ones(s, s) for s in sizes))
    irreps = defaultdict(list)
    idx = 0
    for i in range(N):                                                     # restated collection loops (neural_network.py:859-918)
        for j in range(N):
            cur = []
            if i != j and not mask[i, j]:
                continue
            for n_i, (z_i, l_i) in enumerate(orbitals[i]):
                for n_j, (z_j, l_j) in enumerate(orbitals[j]):
                    for L in range(abs(l_i - l_j), l_i + l_j + 1):
                        if i == j:
                            cur.append(f_ii[L][:, i, :, irreps_ii[(z_i, z_j, n_i, n_j, L)]])
                        else:
                            cur.append(f_ij[L][:, idx, :, irreps_ij[(z_i, z_j, n_i, n_j, L)]])
            if i != j:
                idx += 1
            irreps[(orbitals[i][0][0], orbitals[j][0][0])].append((cur, i, j))
    H0 = NN.generate_matrix_from_irreps(fake, irreps, Norb, atom2orb, begin, end, 1, "cpu", torch.float32)
    H = H0 + H0.transpose(-2, -1)
    w = torch.tensor(rng.normal(size=tuple(H.shape)).astype(np.float32))
    (H * w).sum().backward()
    eye = torch.eye(Norb).unsqueeze(0)
    S = (1 - eye) * H + eye
    fx = dict(z=zs, ptr=ptr, idx_i=np.array(idx_i), idx_j=np.array(idx_j), Fo=np.int64(Fo), H_unsym=H0[0].detach().numpy(), H=H[0].detach().numpy(),
              overlap=S[0].detach().numpy(), w=w[0].numpy(),
              f_ii=np.concatenate([t.detach().numpy()[0] for t in f_ii], axis=1), f_ij=np.concatenate([t.detach().numpy()[0] for t in f_ij], axis=1),
              g_ii=np.concatenate([t.grad.numpy()[0] for t in f_ii], axis=1), g_ij=np.concatenate([t.grad.numpy()[0] for t in f_ij], axis=1),
              ii_keys=np.array(list(irreps_ii.keys())), ii_vals=np.array(list(irreps_ii.values())),
              ij_keys=np.array(list(irreps_ij.keys())), ij_vals=np.array(list(irreps_ij.values())))
    np.savez_compressed(os.path.join(OUT, "phisnet_matrix.npz"), **fx)
    print("phisnet_matrix.npz: Norb", Norb, "pairs", P, "features", Fo, "irreps", len(irreps_ii), len(irreps_ij))


if __name__ == "__main__" and "--matrix" in sys.argv:
    matrix_assembly()


def _network_setup():
    """The REAL NeuralNetwork end to end (embedding -> modules -> pair features -> irreps -> matrices) on a small batch.  CAVEAT, also in
    DESIGN.md: the reference reads its pair-of-pairs table from modules/pindex_dict.npy, which its tree does not contain; the load is answered
    here with the inferred table (nabladft_amd.phisnet.inferred_pair_of_pairs), so this fixture pins everything except that table's content."""
    import types
    sys.path.insert(0, os.path.dirname(HERE))
    from nabladft_amd.phisnet import inferred_pair_of_pairs
    pkg = types.ModuleType("ref_phisnet_nn4")
    pkg.__path__ = ["/root/reference/nablaDFT/phisnet/nn"]
    sys.modules["ref_phisnet_nn4"] = pkg
    real_load = np.load

    def load(path, *a, **k):
        if str(path).endswith("pindex_dict.npy"):
            return np.array({n: tuple(t.numpy() for t in inferred_pair_of_pairs(n)) for n in range(2, 9)}, dtype=object)
        return real_load(path, *a, **k)

    np.load = load
    try:
        nnmod = importlib.import_module("ref_phisnet_nn4.neural_network")
        shells = {1: (0, 0, 1), 6: (0, 0, 0, 1, 1, 2), 8: (0, 0, 0, 1, 1, 2)}
        # one entry per list position; elements repeated so that same-element pairs have off-diagonal irreps (neural_network.py:407-417)
        max_orbitals = tuple(tuple((zz, l) for l in shells[zz]) for zz in (1, 1, 6, 6, 8, 8))
        torch.manual_seed(5)
        hp = dict(max_orbitals=max_orbitals, order=2, num_features=32, num_basis_functions=8, num_modules=2, num_residual_pre_x=1, num_residual_post_x=1,
                  num_residual_pre_vi=1, num_residual_pre_vj=1, num_residual_post_v=1, num_residual_output=1, num_residual_pc=1, num_residual_pn=1,
                  num_residual_ii=1, num_residual_ij=1, num_residual_full_ii=1, num_residual_full_ij=1, num_residual_core_ii=1, num_residual_core_ij=1,
                  num_residual_over_ij=1, basis_functions="exp-bernstein", cutoff=8.0, activation="swish")
        m = nnmod.NeuralNetwork(**hp).float()
    finally:
        np.load = real_load
    rng = np.random.Generator(np.random.PCG64(41))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.abs().max() == 0 or n.endswith("alpha") or n.endswith("beta"):
                p.add_(torch.tensor(rng.normal(0, 0.2, size=tuple(p.shape)).astype(np.float32)))
    sizes = [3, 2, 4]
    zs = np.array([8, 1, 1, 1, 1, 6, 8, 1, 6])
    pos = np.concatenate([rng.normal(0, 1.1, size=(s, 3)) + 0.0 for s in sizes]).astype(np.float32)
    batch = dict(positions=torch.tensor(pos).view(1, -1, 3), atomic_numbers=torch.tensor(zs), orbitals=[tuple((int(a), l) for l in shells[int(a)]) for a in zs],
                 molecule_size=torch.tensor(sizes))
    return m, batch, hp, rng, zs, pos, sizes


def network():
    """Fixture phisnet_network.npz: matrices, energy and all parameter gradients of the REAL NeuralNetwork (see _network_setup)."""
    m, batch, hp, rng, zs, pos, sizes = _network_setup()
    m.predict_energy = True
    import copy
    m64 = copy.deepcopy(m).double()             # the same network evaluated in float64: the truth both fp32 evaluations are measured against
    out = m(batch)
    batch64 = dict(batch, positions=batch["positions"].double())
    out64 = m64(batch64)
    fx = dict(z=zs, positions=pos, sizes=np.array(sizes), electron_config=m.embedding.embedding.electron_config.numpy(),
              hp=np.array([hp["order"], hp["num_features"], hp["num_basis_functions"], hp["num_modules"]]), cutoff=np.float64(hp["cutoff"]))
    loss = loss64 = 0
    for k in ("full_hamiltonian", "core_hamiltonian", "overlap_matrix"):
        w = torch.tensor(rng.normal(size=tuple(out[k].shape)).astype(np.float32))
        fx[k], fx["w_" + k] = out[k][0].detach().numpy(), w[0].numpy()
        fx["f64:" + k] = out64[k][0].detach().numpy()
        loss = loss + (out[k] * w).sum()
        loss64 = loss64 + (out64[k] * w.double()).sum()
    w = torch.tensor(rng.normal(size=tuple(out["energy"].shape)).astype(np.float32))
    fx["energy"], fx["w_energy"] = out["energy"].detach().numpy(), w.numpy()
    fx["f64:energy"] = out64["energy"].detach().numpy()
    loss = loss + (out["energy"] * w).sum()
    loss64 = loss64 + (out64["energy"] * w.double()).sum()
    loss.backward()
    loss64.backward()
    p64 = dict(m64.named_parameters())
    for n, p in m.named_parameters():
        fx["p:" + n] = p.detach().numpy()
        fx["g:" + n] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        fx["g64:" + n] = (p64[n].grad.numpy() if p64[n].grad is not None else np.zeros(tuple(p.shape))).astype(np.float32)   # fp64 result, stored rounded
        fx["rg:" + n] = np.bool_(p.requires_grad)
    np.savez_compressed(os.path.join(OUT, "phisnet_network.npz"), **fx)
    print("phisnet_network.npz:", len(fx), "arrays; Norb", fx["full_hamiltonian"].shape, "params", sum(p.numel() for p in m.parameters()))


def forces():
    """Fixture phisnet_forces.npz: the REAL NeuralNetwork with predict_energy = calculate_forces = True (neural_network.py:92-93, :737, :981-984:
    forces = -autograd.grad(sum(energy), R)), create_graph = False (inference: no second-order graph), evaluated in float32 and in float64."""
    import copy
    m, batch, hp, rng, zs, pos, sizes = _network_setup()
    m.predict_energy = m.calculate_forces = True
    m.create_graph = False
    m64 = copy.deepcopy(m).double()
    out = m(dict(batch, positions=batch["positions"].clone()))
    out64 = m64(dict(batch, positions=batch["positions"].double()))
    fx = dict(z=zs, positions=pos, sizes=np.array(sizes), hp=np.array([hp["order"], hp["num_features"], hp["num_basis_functions"], hp["num_modules"]]),
              cutoff=np.float64(hp["cutoff"]), energy=out["energy"].detach().numpy(), forces=out["forces"].detach().numpy()[0])
    fx["f64:energy"], fx["f64:forces"] = out64["energy"].detach().numpy(), out64["forces"].detach().numpy()[0]
    for n, p in m.named_parameters():
        fx["p:" + n] = p.detach().numpy()
    assert np.abs(fx["forces"]).max() > 0
    np.savez_compressed(os.path.join(OUT, "phisnet_forces.npz"), **fx)
    print("phisnet_forces.npz: |F|max", np.abs(fx["forces"]).max(), "fp32 vs fp64", np.abs(fx["forces"] - fx["f64:forces"]).max() / np.abs(fx["f64:forces"]).max())


def checkpoint():
    """Fixture phisnet_checkpoint.pt: the file the REAL NeuralNetwork.save (neural_network.py:470-503) writes for the network of _network_setup -- what
    `load_from` reads.  The Clebsch-Gordan buffers of the reference module (fp64 tables up to L = 10, constants of the code, tens of MB) are dropped from the
    saved state_dict to keep the fixture small; everything else is the reference's own output (tensors, tuples, numbers, strings: no code objects)."""
    m, batch, hp, rng, zs, pos, sizes = _network_setup()
    path = os.path.join(OUT, "phisnet_checkpoint.pt")
    m.save(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    n0 = len(ck["state_dict"])
    ck["state_dict"] = {k: v for k, v in ck["state_dict"].items() if "clebsch_gordan" not in k}
    torch.save(ck, path)
    print("phisnet_checkpoint.pt:", n0, "->", len(ck["state_dict"]), "state entries;", os.path.getsize(path) // 1024, "kB; keys", sorted(k for k in ck if k != "state_dict"))


if __name__ == "__main__" and "--network" in sys.argv:
    network()
if __name__ == "__main__" and "--checkpoint" in sys.argv:
    checkpoint()
if __name__ == "__main__" and "--forces" in sys.argv:
    forces()
