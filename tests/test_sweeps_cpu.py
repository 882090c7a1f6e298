"""CPU: the hand-derived sweeps (oracle/painn_sweeps.py; the math the HIP engine implements)
against the autograd oracle, in fp64 (derivation check, tol 1e-9) and fp32."""
import numpy as np
import pytest
import torch

from oracle import painn_ref as R
from oracle.painn_sweeps import Sweeps, loss_and_seeds
from tests.helpers import rel_err


def _case(dtype, seed=3):
    cfg = R.PaiNNConfig(hidden_channels=32, num_layers=3, num_rbf=12, cutoff=3.5, max_neighbors=100)
    P = R.make_params(cfg, seed=11, dtype=dtype)
    pos, z, batch, y, ft = R.gen_conformers(seed, 3, size=(5, 14), dtype=dtype)
    ei, _, _ = R.build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    return cfg, P, pos, z, batch, y, ft, ei


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 3e-4)])
def test_sweeps_match_autograd(dtype, tol):
    cfg, P, pos, z, batch, y, ft, ei = _case(dtype)
    e_ref, f_ref, loss_ref, g_ref = R.train_step(P, cfg, pos, z, batch, y, ft, ei)
    sw = Sweeps(P, cfg, pos, z, batch, ei)
    energy, forces = sw.energy_forces()
    assert rel_err(energy.numpy(), e_ref.numpy()) < tol
    assert rel_err(forces.numpy(), f_ref.numpy()) < tol
    loss, gE, gF = loss_and_seeds(energy, forces, y, ft)
    assert abs(float(loss) - float(loss_ref)) < tol * abs(float(loss_ref))
    G = sw.backward(gE, gF)
    for k in P:
        assert rel_err(G[k].numpy(), g_ref[k].numpy()) < tol * 10, k


def test_tangent_is_directional_derivative():
    cfg, P, pos, z, batch, y, ft, ei = _case(torch.float64)
    sw = Sweeps(P, cfg, pos, z, batch, ei)
    sw.energy_forces()
    v = torch.randn(pos.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    edot = sw.tangent(v)
    # Edot = v . dE/dpos = -v . F
    assert abs(float(edot) + float((v * sw.ws["forces"]).sum())) < 1e-9 * max(1.0, abs(float(edot)))


def test_tangent_adjoints_of_the_second_order_sweep_are_the_force_sweep_adjoints():
    """The identity csrc/engine.hip relies on since round 6: in the dual reverse sweep the adjoints of the TANGENT variables obey the force
    sweep's recursion with the same seeds (E_dot = sum of the atom energy tangents is linear in them), so they equal the force sweep's adjoints at every layer
    boundary -- whatever direction the tangent sweep took and whatever the energy seeds are.  float64, 1e-12."""
    cfg, P, pos, z, batch, y, ft, ei = _case(torch.float64)
    sw = Sweeps(P, cfg, pos, z, batch, ei)
    energy, forces = sw.energy_forces()
    lam = {k: v.clone() for k, v in sw.ws.items() if k.startswith("g_x_in") or k.startswith("g_vec_in")}      # left by the force sweep
    assert len(lam) == 2 * cfg.num_layers
    _, gE, gF = loss_and_seeds(energy, forces, y, ft)
    sw.backward(gE, gF)                                                                                         # tangent sweep along -gF, then the dual sweep
    for k, v in lam.items():
        gt = sw.ws["gt_" + k[2:]]
        assert float((gt - v).abs().max()) <= 1e-12 * max(1.0, float(v.abs().max())), k
        assert float((sw.ws[k] - v).abs().max()) > 1e-6 * float(v.abs().max()), k                               # (the primal adjoints are something else)
