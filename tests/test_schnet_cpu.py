"""CPU: schnetpack-SchNet restatement (oracle/spk_schnet_ref.py, PARITY UNPINNED): the four hand-derived sweeps the HIP engine
runs must reproduce the autograd restatement (energies, forces = -dE/dR, loss, every parameter gradient) in fp64."""
import torch

from oracle import painn_ref as R
from oracle import spk_schnet_ref as S
from oracle.spk_painn_ref import full_neighbor_list
from tests.helpers import rel_err


def test_schnet_sweeps_match_autograd_fp64():
    cfg = S.SchNetConfig(n_atom_basis=32, n_interactions=3, n_rbf=14, cutoff=3.5, max_z=20)
    P = S.make_schnet_params(cfg, seed=5, dtype=torch.float64)
    pos, z, batch, y, ft = R.gen_conformers(4, 3, size=(5, 12), dtype=torch.float64)
    e_ref, f_ref, loss_ref, g_ref = S.schnet_train_step(P, cfg, pos, z, batch, y, ft)
    ii, jj = full_neighbor_list(pos, batch, cfg.cutoff)
    sw = S.SchNetSweeps(P, cfg, pos, z, batch, ii, jj)
    e, f, loss, G = sw.train_step(y, ft)
    assert rel_err(e.numpy(), e_ref.numpy()) < 1e-12
    assert rel_err(f.numpy(), f_ref.numpy()) < 1e-11
    assert abs(float(loss) - float(loss_ref)) < 1e-12 * abs(float(loss_ref))
    for k in g_ref:
        assert rel_err(G[k].numpy(), g_ref[k].numpy()) < 1e-9, k


def test_schnet_restatement_invariances():
    """rotation + translation invariance of the energy, equivariance of the forces, permutation of molecules."""
    cfg = S.SchNetConfig(n_atom_basis=16, n_interactions=2, n_rbf=10, cutoff=3.0, max_z=20)
    P = S.make_schnet_params(cfg, seed=6, dtype=torch.float64)
    pos, z, batch, y, ft = R.gen_conformers(9, 2, size=(4, 9), dtype=torch.float64)
    e0, f0, _, _ = S.schnet_train_step(P, cfg, pos, z, batch, y, ft)
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1)))
    e1, f1, _, _ = S.schnet_train_step(P, cfg, pos @ q.T + 0.7, z, batch, y, ft)
    assert rel_err(e1.numpy(), e0.numpy()) < 1e-12 and rel_err(f1.numpy(), (f0 @ q.T).numpy()) < 1e-11
