"""BASELINE.json configs[4] as written: "mixed molecule sizes 10-90 atoms (load-balance stress)".  The golden fixtures hold 26-54-atom molecules; here the
yaml configurations of eSCN, EquiformerV2 and GemNet-OC run on a 10 / 90 / U{10..90}-atom mix and are checked through properties that do not need a CPU run
of a 90-atom molecule (one molecule = one graph):
  * energies and forces of a molecule do not depend on what else is in the batch (90-atom rows stress the per-atom LDS tables and the host-split row tiles);
  * the gradient of a loss that is a sum over molecules equals the sum of the per-molecule gradients (every backward kernel, same stress);
  * everything is finite and the neighbour caps really bind on the large molecules.
Tolerance: 2e-5 of the largest magnitude (float32 sums in a different association: batch rows change the split of the weight-gradient contractions) on the
exact-f32 GEMM engine, whose rows do not depend on the batch.  With the split-bf16 engine (default for large products) the whole batch and the single
molecules run on DIFFERENT engines, i.e. every intermediate differs in its last f32 bit, and autograd forces of a random-weight model amplify that: the
yardstick there is what a pure f32 reordering does to the same outputs (the exact engine against the generic exact-f32 kernels, which sum over k in another
order); it is measured and reported, but the bounds are fixed per model (VERDICT r3 item 8 ii)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


class Batch:
    pass


def _mix(dev, sizes_fixed=(10, 90), n_random=3, seed=11):
    """Conformers of 10 and 90 atoms plus `n_random` of U{10..90}, as one batch and as single-molecule batches."""
    from nabladft_amd.synth import gen_conformers, take_conformers
    parts = [gen_conformers(seed + i, 1, size=int(n)) for i, n in enumerate(sizes_fixed)] + [gen_conformers(seed + 50, n_random, size=(10, 90))]
    pos, z, batch, off = [], [], [], 0
    for p, zz, b, _, _ in parts:
        pos.append(p); z.append(zz); batch.append(b + off); off += int(b.max()) + 1
    pos, z, batch = torch.cat(pos), torch.cat(z), torch.cat(batch)
    n_mol = off
    y, f = torch.zeros(n_mol), torch.zeros(pos.shape[0], 3)

    def mk(idx):
        p, zz, b, _, _ = take_conformers(pos, z, batch, y, f, idx)
        o = Batch()
        o.pos, o.z, o.batch = p.to(dev), zz.to(dev), b.to(dev)
        cnt = torch.bincount(b, minlength=len(idx))
        o.ptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)]).to(dev)
        o.natoms = cnt.to(dev)
        return o
    return mk(list(range(n_mol))), [mk([i]) for i in range(n_mol)], torch.bincount(batch).tolist()


def _check(net, dev, weight_seed=0, tol=2e-5, grad_tol=5e-5, default_tol=None, default_grad_tol=None):
    """Strict (`tol`) on the exact-f32 engine, whose rows do not depend on the batch.  On the default engines the whole batch and the single molecules run on
    DIFFERENT engines (split-bf16 above 192 tiles, exact f32 below), so every intermediate differs in its last f32 bit: the bound there is FIXED per model
    (`default_tol`, stated at the call site with the measurement behind it) -- not derived from anything measured inside the test.  The f32 reordering
    sensitivity (exact engine vs the generic exact-f32 kernels, which sum over k in another order) is still measured and written to the report."""
    from nabladft_amd import _lib
    lib = _lib.load()
    default_tol = tol if default_tol is None else default_tol
    default_grad_tol = grad_tol if default_grad_tol is None else default_grad_tol
    whole, _, _ = _mix(dev)
    try:
        lib.nq_set_gemm_variant(1 | 32)
        out = _check_engine(net, dev, weight_seed, tol, grad_tol)
        Ex, Fx = [t.detach().clone() for t in net(whole)[:2]]
        lib.nq_set_gemm_variant(1 | 16)
        Eg, Fg = [t.detach() for t in net(whole)[:2]]
    finally:
        lib.nq_set_gemm_variant(1)
    reorder = max(float((Ex - Eg).abs().max()) / float(Ex.abs().max()), float((Fx - Fg).abs().max()) / float(Fx.abs().max()))
    try:
        _check_engine(net, dev, weight_seed, default_tol, default_grad_tol)
    finally:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "mixed_sizes_report.txt"), "a") as f:
            f.write(f"{type(net).__name__}: f32 reordering sensitivity of E / F on the 10-90 atom mix {reorder:.3e} (exact engine vs generic exact kernels; reported, "
                    f"not used as a bound); fixed bound on the default engines {default_tol:.1e}\n")
    return out


def _check_engine(net, dev, weight_seed, tol, grad_tol):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    whole, singles, sizes = _mix(dev)
    assert min(sizes) == 10 and max(sizes) == 90
    g = torch.Generator().manual_seed(weight_seed)
    wF = torch.randn(sum(sizes), 3, generator=g).to(dev)
    wE = torch.randn(len(sizes), generator=g).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]

    def run(b, wE_b, wF_b):
        for p in params:
            p.grad = None
        E, F = net(b)[:2]
        ((E.reshape(-1) * wE_b).sum() + (F * wF_b).sum()).backward()
        return E.detach().reshape(-1), F.detach(), [None if p.grad is None else p.grad.detach().clone() for p in params]

    E, F, G = run(whole, wE, wF)
    assert bool(torch.isfinite(E).all()) and bool(torch.isfinite(F).all())
    off = 0
    Gsum = [None] * len(params)
    worst_e = worst_f = 0.0
    for i, (b, n) in enumerate(zip(singles, sizes)):
        Ei, Fi, Gi = run(b, wE[i:i + 1], wF[off:off + n])
        worst_e = max(worst_e, float((Ei - E[i:i + 1]).abs().max()) / max(float(E.abs().max()), 1e-12))
        worst_f = max(worst_f, float((Fi - F[off:off + n]).abs().max()) / max(float(F.abs().max()), 1e-12))
        for k, gk in enumerate(Gi):
            if gk is not None:
                Gsum[k] = gk if Gsum[k] is None else Gsum[k] + gk
        off += n
    with open(os.path.join(ROOT, "gpurun_out", "mixed_sizes_report.txt"), "a") as f:
        f.write(f"{type(net).__name__}: whole batch vs single molecules: E {worst_e:.3e}  F {worst_f:.3e}  (tolerance {tol:.3e})\n")
    assert worst_e < tol and worst_f < tol, (worst_e, worst_f)
    num = den = 0.0
    for a, b in zip(G, Gsum):
        if a is None and b is None:
            continue
        a = torch.zeros_like(b) if a is None else a
        b = torch.zeros_like(a) if b is None else b
        assert bool(torch.isfinite(a).all())
        num += float(((a - b).double() ** 2).sum()); den += float((b.double() ** 2).sum())
    assert (num / max(den, 1e-300)) ** 0.5 < grad_tol, (num / max(den, 1e-300)) ** 0.5
    return whole, sizes


def test_escn_yaml_configuration_on_10_to_90_atoms():
    from nabladft_amd.escn import eSCN
    from tests.test_escn_cpu import FULL
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = eSCN(**FULL).to(dev)
    # eSCN's force head sums a nearly constant scalar field times the unit vectors of 128 sphere points (escn.py:437-457): with random initial weights the sum
    # cancels to ~1e-3 of its terms.  Rounds 3-4 carried that reduction in f32 and needed a bound of 1e-4 (measured 4.0e-5 whole batch vs single molecules);
    # since round 5 the one reduction runs with float64 accumulation (nabladft_amd/escn.py: _SphereSumFn): measured 1.6e-5 (profiles/r05_mixed_sizes_report.txt),
    # held to the same 2e-5 as the other models.
    whole, sizes = _check(net, dev, default_tol=2e-5, default_grad_tol=1e-4)
    G = net.build_graph(whole)
    deg = torch.maximum(torch.bincount(G.dst.cpu(), minlength=sum(sizes)), torch.bincount(G.src.cpu(), minlength=sum(sizes)))
    assert int(deg.max()) >= FULL["max_neighbors"]                     # the cap of 40 binds on the 90-atom molecule ...
    assert int(deg[:10].max()) <= 9                                    # ... and cannot on the 10-atom one


def test_equiformer_v2_yaml_configuration_on_10_to_90_atoms():
    from nabladft_amd.equiformer_v2 import EquiformerV2_OC20
    from tests.test_equiformer_cpu import FULL
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    net = EquiformerV2_OC20(**FULL).to(dev).eval()                     # eval: no drop-path / attention dropout (stochastic per batch)
    _check(net, dev)


def test_gemnet_oc_yaml_configuration_on_10_to_90_atoms():
    from nabladft_amd.gemnet_oc import GemNetOC
    from tests.test_gemnet_gpu import FULL
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    net = GemNetOC(**FULL).to(dev)
    _check(net, dev)
