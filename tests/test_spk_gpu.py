"""GPU: the schnetpack-shaped potential (nabladft_amd/spk.py -> engine filter_mode=1) against the CPU restatement
oracle/spk_painn_ref.py in fp64.  PARITY UNPINNED (no schnetpack here, see the oracle's header): this pins the HIP path to the
restatement only.  Tolerances: energy 2e-6, forces 2e-5, parameter gradients 2e-4 relative (fp32 engine vs fp64 restatement),
the same bars as the pinned painn_pyg golden tests."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _potential(scfg):
    from nabladft_amd import spk
    return spk.NeuralNetworkPotential(
        representation=spk.PaiNN(n_atom_basis=scfg.n_atom_basis, n_interactions=scfg.n_interactions,
                                 radial_basis=spk.GaussianRBF(n_rbf=scfg.n_rbf, cutoff=scfg.cutoff),
                                 cutoff_fn=spk.CosineCutoff(scfg.cutoff), max_z=scfg.max_z),
        input_modules=[spk.PairwiseDistances()],
        output_modules=[spk.Atomwise(n_in=scfg.n_atom_basis, output_key="energy"), spk.Forces(energy_key="energy", force_key="forces")],
        postprocessors=[spk.AddOffsets("energy", add_mean=True)])


@pytest.mark.parametrize("gw", ["pair_rows", "per_molecule"])
@pytest.mark.parametrize("F,L,R,cutoff,n_mol,size", [(64, 2, 20, 4.0, 5, (3, 14)), (128, 6, 100, 5.0, 6, (8, 30)), (256, 1, 50, 5.0, 3, (5, 20))])
def test_spk_potential_matches_restatement(F, L, R, cutoff, n_mol, size, gw, monkeypatch):
    """gw: filter-network weight gradient from gphi / gpsi pair rows (small batches) or from node rows staged per molecule in LDS (csrc/molpair.hip, default
    from 4096 atoms per step; forced here).  schnetpack mode exercises the cosine-cutoff bias multiplier (beta, beta') of the window record."""
    if gw == "per_molecule":
        monkeypatch.setenv("NQ_MOLGW", "1")
    else:
        monkeypatch.delenv("NQ_MOLGW", raising=False)
    from oracle import painn_ref as PR
    from oracle import spk_painn_ref as S
    scfg = S.SpkPaiNNConfig(n_atom_basis=F, n_interactions=L, n_rbf=R, cutoff=cutoff, max_z=20)
    P = S.make_spk_params(scfg, seed=F + L)
    pos, z, batch, y, ft = PR.gen_conformers(31 + L, n_mol, size=size)
    P64 = {k: v.double() for k, v in P.items()}
    e_ref, f_ref, loss_ref, g_ref = S.spk_train_step(P64, scfg, pos.double(), z, batch, y.double(), ft.double())

    pot = _potential(scfg)
    missing, unexpected = pot.load_state_dict(P, strict=False)
    assert not unexpected and all(("radial_basis" in m or "cutoff_fn" in m or "postprocessors" in m) for m in missing), (missing, unexpected)
    pot = pot.cuda().train()
    inputs = {"_positions": pos.cuda(), "_atomic_numbers": z.cuda(), "_idx_m": batch.cuda()}
    out = pot(dict(inputs))
    assert set(out) == {"energy", "forces"}
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
    print(f"spk F={F} L={L} R={R}: energy {e_err:.2e} forces {f_err:.2e} worst grad {worst:.2e}")

    # inference: post-processing adds mean * n_atoms (train mode skips it)
    pot.eval()
    pot.postprocessors[0].mean.fill_(-3.5)
    with torch.no_grad():
        out2 = pot(dict(inputs))
    n_atoms = torch.bincount(batch).double()
    assert rel_err(out2["energy"].cpu().numpy(), (e_ref + (-3.5) * n_atoms).numpy()) < 2e-6
    assert torch.equal(out2["forces"], out["forces"].detach())          # same kernels, bitwise deterministic


def test_spk_fused_step_equals_autograd_plus_torch_adamw():
    """FusedTrainStep on the spk-shaped potential (engine-layout training + writeback) == autograd boundary + torch MSE losses +
    torch.optim.AdamW on the spk-shaped parameters, for two consecutive steps."""
    import copy
    import nabladft_amd as nq
    from oracle import painn_ref as PR
    from oracle import spk_painn_ref as S
    scfg = S.SpkPaiNNConfig(n_atom_basis=64, n_interactions=2, n_rbf=20, cutoff=4.0, max_z=20)
    P = S.make_spk_params(scfg, seed=3)
    pos, z, batch, y, ft = PR.gen_conformers(77, 6, size=(4, 16))
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
        if it == 0:      # same kernels on the same parameters: bitwise; afterwards the two AdamW implementations differ by rounding
            assert torch.equal(fs.energy, out["energy"].detach()) and torch.equal(fs.forces, out["forces"].detach())
        else:
            assert torch.allclose(fs.energy, out["energy"].detach(), rtol=1e-5, atol=1e-6) and torch.allclose(fs.forces, out["forces"].detach(), rtol=1e-4, atol=1e-6)
    fs.writeback()
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb
        assert torch.allclose(pa, pb, rtol=0, atol=2e-6), (na, (pa - pb).abs().max().item())
