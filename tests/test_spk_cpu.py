"""CPU: schnetpack-PaiNN restatement (oracle/spk_painn_ref.py, PARITY UNPINNED) and its mapping onto the engine's math:
spk parameters gathered into the painn_pyg layout + filter_mode="spk" must reproduce the spk energies/forces/gradients
(fp64, derivation check), and the hand-derived sweeps must hold in that mode too."""
import numpy as np
import pytest
import torch

from oracle import painn_ref as R
from oracle import spk_painn_ref as S
from oracle.painn_sweeps import Sweeps
from tests.helpers import rel_err


def _setup(dtype):
    scfg = S.SpkPaiNNConfig(n_atom_basis=32, n_interactions=3, n_rbf=14, cutoff=3.5, max_z=20)
    P = S.make_spk_params(scfg, seed=8, dtype=dtype)
    pos, z, batch, y, ft = R.gen_conformers(2, 3, size=(5, 12), dtype=dtype)
    return scfg, P, pos, z, batch, y, ft


def _to_engine(scfg, P):
    idx = S.spk_to_engine_index(scfg)
    flat = torch.cat([P[k].reshape(-1) for k, _ in S.spk_param_shapes(scfg)])[idx]
    cfg = R.PaiNNConfig(hidden_channels=scfg.n_atom_basis, num_layers=scfg.n_interactions, num_rbf=scfg.n_rbf, cutoff=scfg.cutoff,
                        max_neighbors=10 ** 6, num_elements=scfg.max_z - 1, filter_mode="spk")
    out, o = {}, 0
    for name, shp in R.param_shapes(cfg):
        n = int(np.prod(shp))
        out[name] = flat[o:o + n].view(shp)
        o += n
    assert o == flat.numel()
    return cfg, out, idx


def test_spk_maps_onto_engine_math_fp64():
    scfg, P, pos, z, batch, y, ft = _setup(torch.float64)
    e_s, f_s, loss_s, g_s = S.spk_train_step(P, scfg, pos, z, batch, y, ft)
    cfg, Pe, idx = _to_engine(scfg, P)
    ei, _, _ = R.build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    ii, jj = S.full_neighbor_list(pos, batch, scfg.cutoff)
    assert ei.shape[1] == ii.numel()                      # same edge set as the ASE-style list
    e_p, f_p = R.energy_forces(Pe, cfg, pos, z, batch, ei)
    assert rel_err(e_p.numpy(), e_s.numpy()) < 1e-12 and rel_err(f_p.numpy(), f_s.numpy()) < 1e-11
    # gradients: engine-layout gradient scattered back through the gather index == spk gradient (MSE loss seeds)
    sw = Sweeps(Pe, cfg, pos, z, batch, ei)
    energy, forces = sw.energy_forces()
    gE = 2 * (energy - y) / energy.numel()
    gF = 2 * (forces - ft) / forces.numel()
    G = sw.backward(gE, gF)
    g_engine = torch.cat([G[k].reshape(-1) for k, _ in R.param_shapes(cfg)])
    g_spk = torch.zeros(sum(int(np.prod(s)) for _, s in S.spk_param_shapes(scfg)), dtype=torch.float64).index_add_(0, idx, g_engine)
    ref = torch.cat([g_s[k].reshape(-1) for k, _ in S.spk_param_shapes(scfg)])
    # embedding row 0 (padding) is not part of the engine layout; its gradient is zero in spk as no atom has Z = 0
    assert rel_err(g_spk.numpy(), ref.numpy()) < 1e-9


def test_spk_rejects_unbuilt_options():
    from nabladft_amd import spk
    with pytest.raises(NotImplementedError):
        spk.GaussianRBF(20, 5.0, trainable=True)
    with pytest.raises(NotImplementedError):
        spk.PaiNN(128, 3, spk.GaussianRBF(20, 5.0), spk.CosineCutoff(5.0), shared_filters=True)
    pot = spk.NeuralNetworkPotential(spk.PaiNN(64, 1, spk.GaussianRBF(20, 5.0), spk.CosineCutoff(5.0), max_z=10), [spk.PairwiseDistances()],
                                     [spk.Atomwise(n_in=64, output_key="energy"), spk.Forces()])
    names = set(pot.state_dict())
    assert {k for k, _ in S.spk_param_shapes(S.SpkPaiNNConfig(64, 1, 20, 5.0, 10))} <= names
    with pytest.raises(RuntimeError):      # product has no CPU path
        pot({"_positions": torch.zeros(2, 3), "_atomic_numbers": torch.ones(2, dtype=torch.long), "_idx_m": torch.zeros(2, dtype=torch.long)})
