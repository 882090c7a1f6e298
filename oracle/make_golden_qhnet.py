"""TEST INFRASTRUCTURE (container-only): golden vectors for the Hamiltonian block assembly, produced by the REAL reference methods
(QHNet._get_mask, QHNet.build_final_matrix, the transpose-index code of QHNet.build_graph, HamiltonianLoss) imported through
oracle/qhnet_import.py.  Writes tests/golden/qhnet_blocks.npz.  Run:  python oracle/make_golden_qhnet.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.hblock_ref import full_graph  # noqa: E402
from oracle.qhnet_import import load_qhnet  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ORBITALS = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 7: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2], 9: [0, 0, 0, 1, 1, 2],
            16: [0, 0, 0, 0, 1, 1, 1, 2], 17: [0, 0, 0, 0, 1, 1, 1, 2], 35: [0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]}   # config/model/qhnet.yaml:14-22


def main():
    ref = load_qhnet()
    QHNet = ref["qhnet"].QHNet
    masks, s_max, p_max, d_max = QHNet._get_mask(None, ORBITALS)
    assert (s_max, p_max, d_max) == (5, 4, 3)
    S = s_max + 3 * p_max + 5 * d_max
    rng = np.random.Generator(np.random.PCG64(11))
    sizes = [3, 1, 6, 4]                                       # includes a single-atom molecule (no off-diagonal block)
    zs = rng.choice(list(ORBITALS.keys()), size=sum(sizes))
    zs[0], zs[4] = 35, 1                                       # make sure the largest and the smallest layouts occur
    z = torch.tensor(zs, dtype=torch.long)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long)
    ei = full_graph(ptr)
    N, P = int(ptr[-1]), ei.shape[1]
    diag = torch.tensor(rng.normal(size=(N, S, S)).astype(np.float32), requires_grad=True)
    nondiag = torch.tensor(rng.normal(size=(P, S, S)).astype(np.float32), requires_grad=True)
    fake_self = types.SimpleNamespace(orbital_mask=masks)
    data = types.SimpleNamespace(ptr=ptr, z=z, full_edge_index=ei)
    H0 = QHNet.build_final_matrix(fake_self, data, diag, nondiag)          # qhnet.py:293-321
    H = H0 + H0.transpose(-1, -2)                                           # qhnet.py:237
    # transpose index: the loop of QHNet.build_graph (qhnet.py:273-283) verbatim on this graph
    start, tr = 0, []
    for g in range(ptr.shape[0] - 1):
        n = ptr[g + 1] - ptr[g]
        gei = ei[:, start:start + n * (n - 1)]
        sub = gei - ptr[g]
        bias = (sub[0] < sub[1]).type(torch.int)
        tr.append(sub[0] * (n - 1) + sub[1] - bias + start)
        start = start + n * (n - 1)
    tr = torch.cat(tr, dim=-1)
    # loss (qhnet.py:366-377 builds target and mask as block_diag) and its gradient w.r.t. the blocks
    norb = [len(masks[int(a)]) for a in z]
    mol_orb = [sum(norb[int(ptr[b]):int(ptr[b + 1])]) for b in range(len(sizes))]
    tblocks = [torch.tensor(rng.normal(size=(m, m)).astype(np.float32)) for m in mol_orb]
    target = torch.block_diag(*tblocks)
    mask = torch.block_diag(*[torch.ones_like(t) for t in tblocks])
    loss = ref["loss"].HamiltonianLoss()(H, target, mask)
    loss.backward()
    np.savez_compressed(
        os.path.join(OUT, "qhnet_blocks.npz"), z=z.numpy(), ptr=ptr.numpy(), edge_index=ei.numpy(), S=np.int64(S),
        mask_keys=np.array(sorted(masks)), mask_vals=np.array([np.pad(masks[k].numpy(), (0, S - len(masks[k])), constant_values=-1) for k in sorted(masks)]),
        smax_pmax_dmax=np.array([s_max, p_max, d_max]), diag=diag.detach().numpy(), nondiag=nondiag.detach().numpy(), H=H.detach().numpy(),
        H_unsym=H0.detach().numpy(), transpose_index=tr.numpy(), target=target.numpy(), loss=np.float64(loss.item()),
        g_diag=diag.grad.numpy(), g_nondiag=nondiag.grad.numpy(),
        masked_mae=np.float64((torch.abs(H.detach() - target).sum() / torch.count_nonzero(target)).item()))
    print("qhnet_blocks.npz:", N, "atoms,", P, "pairs, M =", H.shape[0], "loss", float(loss))


if __name__ == "__main__":
    main()
