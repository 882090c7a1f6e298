"""GPU: the schnetpack-shaped SchNet potential (nabladft_amd/spk.py -> csrc/schnet.hip) against the fp64 CPU restatement
oracle/spk_schnet_ref.py.  PARITY UNPINNED (no schnetpack here): pins the HIP path to the restatement only.
Tolerances (fp32 engine vs fp64 restatement): energy 2e-6, forces 2e-5, parameter gradients 2e-4 relative."""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _potential(scfg):
    from nabladft_amd import spk
    return spk.NeuralNetworkPotential(
        representation=spk.SchNet(n_atom_basis=scfg.n_atom_basis, n_interactions=scfg.n_interactions,
                                  radial_basis=spk.GaussianRBF(n_rbf=scfg.n_rbf, cutoff=scfg.cutoff),
                                  cutoff_fn=spk.CosineCutoff(scfg.cutoff), max_z=scfg.max_z),
        input_modules=[spk.PairwiseDistances()],
        output_modules=[spk.Atomwise(n_in=scfg.n_atom_basis, output_key="energy"), spk.Forces(energy_key="energy", force_key="forces")],
        postprocessors=[spk.AddOffsets("energy", add_mean=True)])


@pytest.mark.parametrize("F,L,R,cutoff,n_mol,size", [(64, 2, 20, 4.0, 5, (3, 14)), (128, 6, 100, 5.0, 6, (8, 30)), (256, 1, 50, 5.0, 3, (5, 20)),
                                                     (128, 2, 100, 5.0, 40, "drug")])
def test_schnet_potential_matches_restatement(F, L, R, cutoff, n_mol, size):
    from oracle import painn_ref as PR
    from oracle import spk_schnet_ref as S
    scfg = S.SchNetConfig(n_atom_basis=F, n_interactions=L, n_rbf=R, cutoff=cutoff, max_z=20)
    P = S.make_schnet_params(scfg, seed=F + L)
    pos, z, batch, y, ft = PR.gen_conformers(41 + L, n_mol, size=size)
    P64 = {k: v.double() for k, v in P.items()}
    e_ref, f_ref, loss_ref, g_ref = S.schnet_train_step(P64, scfg, pos.double(), z, batch, y.double(), ft.double())

    pot = _potential(scfg)
    missing, unexpected = pot.load_state_dict(P, strict=False)
    assert not unexpected and all(("radial_basis" in m or "cutoff_fn" in m or "postprocessors" in m) for m in missing), (missing, unexpected)
    pot = pot.cuda().train()
    inputs = {"_positions": pos.cuda(), "_atomic_numbers": z.cuda(), "_idx_m": batch.cuda()}
    out = pot(dict(inputs))
    loss = torch.nn.functional.mse_loss(out["energy"], y.cuda()) + torch.nn.functional.mse_loss(out["forces"], ft.cuda())
    loss.backward()
    e_err, f_err = rel_err(out["energy"].detach().cpu().numpy(), e_ref.numpy()), rel_err(out["forces"].detach().cpu().numpy(), f_ref.numpy())
    assert e_err < 2e-6 and f_err < 2e-5, (e_err, f_err)
    assert abs(loss.item() - loss_ref.item()) < 2e-5 * abs(loss_ref.item())
    worst = 0.0
    for name, p in pot.named_parameters():
        g = g_ref[name].numpy()
        assert p.grad is not None, name
        scale = max(np.abs(g).max(), np.sqrt((g ** 2).mean()) + 1e-30)
        e = float(np.abs(p.grad.cpu().numpy().astype(np.float64) - g).max() / scale)
        worst = max(worst, e)
        assert e < 2e-4, (name, e)
    print(f"schnet F={F} L={L} R={R} atoms={pos.shape[0]}: energy {e_err:.2e} forces {f_err:.2e} worst grad {worst:.2e}")
    # bitwise reproducible (gather-only reverse sweeps)
    pot.zero_grad()
    out2 = pot(dict(inputs))
    (torch.nn.functional.mse_loss(out2["energy"], y.cuda()) + torch.nn.functional.mse_loss(out2["forces"], ft.cuda())).backward()
    assert torch.equal(out2["energy"], out["energy"]) and torch.equal(out2["forces"], out["forces"])


def test_schnet_fused_step_equals_autograd_plus_torch_adamw():
    import nabladft_amd as nq
    from oracle import painn_ref as PR
    from oracle import spk_schnet_ref as S
    scfg = S.SchNetConfig(n_atom_basis=64, n_interactions=2, n_rbf=20, cutoff=4.0, max_z=20)
    P = S.make_schnet_params(scfg, seed=3)
    pos, z, batch, y, ft = PR.gen_conformers(78, 6, size=(4, 16))
    a = _potential(scfg)
    a.load_state_dict(P, strict=False)
    a = a.cuda().train()
    b = copy.deepcopy(a)
    fs = nq.FusedTrainStep(a, lr=1e-3, weight_decay=0.01, max_grad_norm=0.0)
    opt = torch.optim.AdamW(b.parameters(), lr=1e-3, weight_decay=0.01)
    bt = nq.Batch(pos, z, batch, y, ft).to("cuda")
    for it in range(2):
        loss_a = float(fs(bt))
        opt.zero_grad()
        out = b({"_positions": bt.pos, "_atomic_numbers": bt.z, "_idx_m": bt.batch})
        loss_b = torch.nn.functional.mse_loss(out["energy"], bt.y) + torch.nn.functional.mse_loss(out["forces"], bt.forces)
        loss_b.backward()
        opt.step()
        assert abs(loss_a - loss_b.item()) <= 1e-5 * abs(loss_b.item()), (it, loss_a, loss_b.item())
        if it == 0:
            assert torch.equal(fs.energy, out["energy"].detach()) and torch.equal(fs.forces, out["forces"].detach())
    fs.writeback()
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.allclose(pa, pb, rtol=0, atol=2e-5), (na, (pa - pb).abs().max().item())


def test_schnet_and_spk_painn_handle_isolated_atoms_and_dimers():
    """Edge cases of the neighbour list: a single-atom molecule (no pair at all), a dimer and a molecule whose atoms are partly beyond the
    cutoff of each other -- both schnetpack-shaped models against their restatements."""
    from oracle import painn_ref as PR
    from oracle import spk_painn_ref as SP
    from oracle import spk_schnet_ref as S
    from nabladft_amd import spk
    pp, zz, bb = [], [], []
    for m, n in enumerate([1, 2, 7, 1, 12]):
        p, zc, _, _, _ = PR.gen_conformers(500 + m, 1, size=n)
        pp.append(p * (1.6 if m == 4 else 1.0)), zz.append(zc), bb.append(torch.full((n,), m, dtype=torch.long))
    pos, z, batch = torch.cat(pp), torch.cat(zz), torch.cat(bb)
    g = torch.Generator().manual_seed(4)
    y, ft = torch.randn(5, generator=g), 0.05 * torch.randn(pos.shape[0], 3, generator=g)
    inputs = {"_positions": pos.cuda(), "_atomic_numbers": z.cuda(), "_idx_m": batch.cuda()}
    for kind in ("schnet", "painn"):
        if kind == "schnet":
            scfg = S.SchNetConfig(n_atom_basis=64, n_interactions=2, n_rbf=20, cutoff=3.0, max_z=20)
            P = S.make_schnet_params(scfg, seed=9)
            ref = S.schnet_train_step({k: v.double() for k, v in P.items()}, scfg, pos.double(), z, batch, y.double(), ft.double())
            pot = _potential(scfg)
        else:
            scfg = SP.SpkPaiNNConfig(n_atom_basis=64, n_interactions=2, n_rbf=20, cutoff=3.0, max_z=20)
            P = SP.make_spk_params(scfg, seed=9)
            ref = SP.spk_train_step({k: v.double() for k, v in P.items()}, scfg, pos.double(), z, batch, y.double(), ft.double())
            pot = spk.NeuralNetworkPotential(
                representation=spk.PaiNN(n_atom_basis=64, n_interactions=2, radial_basis=spk.GaussianRBF(n_rbf=20, cutoff=3.0), cutoff_fn=spk.CosineCutoff(3.0), max_z=20),
                input_modules=[spk.PairwiseDistances()], output_modules=[spk.Atomwise(n_in=64, output_key="energy"), spk.Forces()])
        pot.load_state_dict(P, strict=False)
        pot = pot.cuda().train()
        out = pot(dict(inputs))
        loss = torch.nn.functional.mse_loss(out["energy"], y.cuda()) + torch.nn.functional.mse_loss(out["forces"], ft.cuda())
        loss.backward()
        e_ref, f_ref, _, g_ref = ref
        assert rel_err(out["energy"].detach().cpu().numpy(), e_ref.numpy()) < 2e-6, kind
        assert rel_err(out["forces"].detach().cpu().numpy(), f_ref.numpy()) < 2e-5, kind
        assert float(out["forces"][0].abs().max()) == 0.0 and float(out["forces"][10].abs().max()) == 0.0       # isolated atoms feel no force
        for name, p in pot.named_parameters():
            gr = g_ref[name].numpy()
            scale = max(np.abs(gr).max(), np.sqrt((gr ** 2).mean()) + 1e-30)
            assert float(np.abs(p.grad.cpu().numpy().astype(np.float64) - gr).max() / scale) < 2e-4, (kind, name)
