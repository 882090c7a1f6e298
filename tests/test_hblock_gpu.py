"""GPU: Hamiltonian block assembly kernels (csrc/hblock.hip via nabladft_amd.hamiltonian) against golden vectors produced by the
REAL reference methods (QHNet.build_final_matrix, H + H^T, HamiltonianLoss + autograd; oracle/make_golden_qhnet.py).
Assembly is pure data movement + one add: bit-exact.  Loss / gradients: 1e-6 relative (fp32)."""
import os
import time

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, rel_err
from tests.test_hblock_cpu import ORBITALS

pytestmark = pytest.mark.gpu


def test_assembly_loss_and_gradients_match_reference():
    from nabladft_amd import hamiltonian as HM
    fx = dict(np.load(os.path.join(GOLDEN, "qhnet_blocks.npz")))
    asm = HM.BlockAssembler(ORBITALS)
    z, ptr, ei = torch.tensor(fx["z"]).cuda(), torch.tensor(fx["ptr"]).cuda(), torch.tensor(fx["edge_index"]).cuda()
    diag = torch.tensor(fx["diag"]).cuda().requires_grad_(True)
    nondiag = torch.tensor(fx["nondiag"]).cuda().requires_grad_(True)
    plan = asm.plan(z, ptr, ei)
    packed = asm.assemble(plan, diag, nondiag, symmetrize=True)
    asm.check(plan)
    H = asm.to_dense(plan, packed.detach())
    assert np.array_equal(H.cpu().numpy(), fx["H"])                                        # bit-exact
    H0 = asm.to_dense(plan, asm.assemble(plan, diag, nondiag, symmetrize=False).detach())
    assert np.array_equal(H0.cpu().numpy(), fx["H_unsym"])
    data = type("D", (), dict(z=z, ptr=ptr, full_edge_index=ei))
    assert torch.equal(asm.build_final_matrix(data, diag, nondiag), H0)                    # drop-in signature
    assert torch.equal(asm.from_dense(plan, H), packed.detach())
    target = asm.from_dense(plan, torch.tensor(fx["target"]).cuda())
    loss = HM.HamiltonianLoss()(packed, target)
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 1e-6 * float(fx["loss"])
    assert rel_err(diag.grad.cpu().numpy(), fx["g_diag"]) < 2e-6 and rel_err(nondiag.grad.cpu().numpy(), fx["g_nondiag"]) < 2e-6
    assert abs(HM.masked_mae(packed, target).item() - float(fx["masked_mae"])) < 1e-6 * float(fx["masked_mae"])
    # a pair list that is not the full graph is reported (the reference fails on .item())
    bad = asm.plan(z, ptr, ei[:, :-1])
    asm.assemble(bad, diag.detach(), nondiag.detach()[:-1])
    with pytest.raises(IndexError):
        asm.check(bad)
    with pytest.raises(KeyError):
        asm.plan(torch.full_like(z, 3), ptr, ei)                                           # Z = 3 has no orbitals entry


def test_assembly_matches_restatement_on_a_drug_sized_batch_and_is_fast():
    """16 conformers of ~40 atoms (M ~ 400 orbitals each): kernel vs the vectorised CPU restatement, and the time of one call."""
    from nabladft_amd import hamiltonian as HM
    from oracle import hblock_ref as HB
    rng = np.random.Generator(np.random.PCG64(5))
    sizes = rng.integers(30, 50, size=16)
    zs = rng.choice([1, 1, 1, 6, 6, 7, 8, 9, 16, 17, 35], size=int(sizes.sum()))
    z = torch.tensor(zs, dtype=torch.long)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long)
    asm = HM.BlockAssembler(ORBITALS)
    ei = HM.full_pair_index(ptr.cuda())
    N, P = int(ptr[-1]), int(ei.shape[1])
    diag = torch.tensor(rng.normal(size=(N, 32, 32)).astype(np.float32)).cuda()
    nondiag = torch.tensor(rng.normal(size=(P, 32, 32)).astype(np.float32)).cuda()
    plan = asm.plan(z.cuda(), ptr.cuda(), ei)
    packed = asm.assemble(plan, diag, nondiag, symmetrize=True)
    asm.check(plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        plan = asm.plan(z.cuda(), ptr.cuda(), ei)
        packed = asm.assemble(plan, diag, nondiag, symmetrize=True)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) * 100
    masks, *_ = HB.orbital_masks(ORBITALS)
    t0 = time.perf_counter()
    Href = HB.build_final_matrix(z, ptr, ei.cpu(), masks, diag.cpu(), nondiag.cpu(), symmetrize=True)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    assert torch.equal(asm.to_dense(plan, packed).cpu(), Href)
    print(f"hblock: {N} atoms, {P} pairs, {plan.m_total} orbitals: GPU plan+assemble {gpu_ms:.3f} ms, CPU restatement {cpu_ms:.0f} ms")


def test_phisnet_irreps_to_matrix_matches_reference():
    """PhiSNet irreps -> Hamiltonian / overlap matrix (row a24, assembly part): golden vectors from the REAL compute_matrix_irreps /
    matrix_block / generate_matrix_from_irreps with the reference's Clebsch-Gordan table (oracle/make_golden_phisnet.py --matrix)."""
    from nabladft_amd import hamiltonian as HM
    from tests.so3_helpers import FixtureCG
    fx = dict(np.load(os.path.join(GOLDEN, "phisnet_matrix.npz")))
    atom2orb = {1: ((1, 0), (1, 0), (1, 1)), 6: ((6, 0), (6, 0), (6, 0), (6, 1), (6, 1), (6, 2)), 8: ((8, 0), (8, 0), (8, 0), (8, 1), (8, 1), (8, 2))}
    # the index dictionaries are the model's; the host mirror of compute_matrix_irreps rebuilds them identically
    number_L, irreps_ii = [0] * 5, {}
    for zz in sorted(atom2orb):
        irreps_ii, number_L = HM.compute_matrix_irreps(atom2orb[zz], atom2orb[zz], irreps_ii, number_L)
    number_L, irreps_ij = [0] * 5, {}
    for za in sorted(atom2orb):
        for zb in sorted(atom2orb):
            irreps_ij, number_L = HM.compute_matrix_irreps(atom2orb[za], atom2orb[zb], irreps_ij, number_L)
    assert {tuple(int(v) for v in k): int(x) for k, x in zip(fx["ii_keys"], fx["ii_vals"])} == irreps_ii
    assert {tuple(int(v) for v in k): int(x) for k, x in zip(fx["ij_keys"], fx["ij_vals"])} == irreps_ij
    asm = HM.IrrepsAssembler(atom2orb, irreps_ii, irreps_ij, FixtureCG())
    z, ptr = torch.tensor(fx["z"]).cuda(), torch.tensor(fx["ptr"]).cuda()
    plan = asm.plan(z, ptr, torch.tensor(fx["idx_i"]).cuda(), torch.tensor(fx["idx_j"]).cuda())
    f_ii = torch.tensor(fx["f_ii"]).cuda().requires_grad_(True)
    f_ij = torch.tensor(fx["f_ij"]).cuda().requires_grad_(True)
    H0 = asm.to_dense(plan, asm.assemble(plan, f_ii, f_ij, symmetrize=False).detach())
    asm.check(plan)
    packed = asm.assemble(plan, f_ii, f_ij, symmetrize=True)
    H = asm.to_dense(plan, packed.detach())
    S = asm.to_dense(plan, asm.assemble(plan, f_ii, f_ij, symmetrize=True, unit_diagonal=True).detach())
    assert rel_err(H0.cpu().numpy(), fx["H_unsym"]) < 2e-6 and rel_err(H.cpu().numpy(), fx["H"]) < 2e-6 and rel_err(S.cpu().numpy(), fx["overlap"]) < 2e-6
    (packed * asm.from_dense(plan, torch.tensor(fx["w"]).cuda())).sum().backward()
    # the reference's dense weights also cover the zero off-molecule region; only the in-molecule part of w matters
    assert rel_err(f_ii.grad.cpu().numpy(), fx["g_ii"]) < 2e-6 and rel_err(f_ij.grad.cpu().numpy(), fx["g_ij"]) < 2e-6
