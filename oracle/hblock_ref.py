"""TEST INFRASTRUCTURE -- CPU restatement of the Hamiltonian block assembly of QHNet (SURVEY.md section 8, rows a19 / a20 and the
transpose index of a13).  Pinned: tests/golden/qhnet_blocks.npz is produced by the *real* reference methods
(oracle/make_golden_qhnet.py -> oracle/qhnet_import.py) and tests/test_hblock_cpu.py checks this restatement against it bit for bit.

  orbital_masks        QHNet._get_mask                 /root/reference/nablaDFT/qhnet/qhnet.py:323-342
  transpose_index      QHNet.build_graph (tail)        qhnet.py:273-283
  build_final_matrix   QHNet.build_final_matrix        qhnet.py:293-321  (+ H + H^T of qhnet.py:233-238)
  hamiltonian_loss     HamiltonianLoss.forward         /root/reference/nablaDFT/qhnet/loss.py:9-16
  masked_mae           MaskedMeanAbsoluteError.update  /root/reference/nablaDFT/qhnet/masked_mae.py:12-20
"""
import numpy as np
import torch


def orbital_masks(orbitals):
    """{Z: list of slots} inside the padded per-atom block, (s_max, p_max, d_max).  qhnet.py:323-342: the block is laid out as
    s_max s-slots, 3*p_max p-slots, 5*d_max d-slots (the largest atom type sets the padding); an atom uses the first slots of each."""
    max_z = max(orbitals.keys())
    _, counts = np.unique(orbitals[max_z], return_counts=True)
    s_max, p_max, d_max = (int(c) for c in counts)
    starts = [0, s_max, s_max + 3 * p_max]
    mult = [1, 3, 5]
    masks = {}
    for z, orb in orbitals.items():
        _, cnt = np.unique(orb, return_counts=True)
        m = []
        for l, c in enumerate(cnt):
            m.extend(range(starts[l], starts[l] + int(c) * mult[l]))
        masks[int(z)] = m
    return masks, s_max, p_max, d_max


def full_graph(ptr):
    """torch_cluster.radius_graph on a cutoff larger than every molecule (max_num_neighbors = num_nodes): per centre (row 1,
    ascending) all other atoms of the molecule ascending (row 0).  QHNet names row 0 ``dst`` and row 1 ``src`` (qhnet.py:262)."""
    r0, r1 = [], []
    for b in range(len(ptr) - 1):
        a0, a1 = int(ptr[b]), int(ptr[b + 1])
        for c in range(a0, a1):
            for j in range(a0, a1):
                if j != c:
                    r0.append(j), r1.append(c)
    return torch.tensor([r0, r1], dtype=torch.long)


def transpose_index(edge_index, ptr):
    """qhnet.py:273-283 -- position of the reverse edge, assuming the per-molecule full graph in radius_graph order."""
    out, start = [], 0
    for b in range(len(ptr) - 1):
        n = int(ptr[b + 1] - ptr[b])
        sub = edge_index[:, start:start + n * (n - 1)] - ptr[b]
        bias = (sub[0] < sub[1]).to(torch.int)
        out.append(sub[0] * (n - 1) + sub[1] - bias + start)
        start += n * (n - 1)
    return torch.cat(out)


def build_final_matrix(z, ptr, edge_index, masks, diag, nondiag, symmetrize=True):
    """Dense block-diagonal [M, M] matrix: block (dst atom rows, src atom columns) = diag[i] (dst == src) or nondiag[e] with
    edge_index[0][e] == dst and edge_index[1][e] == src, restricted to the atoms' orbital slots; then H + H^T."""
    norb = torch.tensor([len(masks[int(a)]) for a in z])
    optr = torch.cat([norb.new_zeros(1), norb.cumsum(0)])
    M = int(optr[-1])
    H = torch.zeros(M, M, dtype=diag.dtype)
    look = {(int(d), int(s)): e for e, (d, s) in enumerate(zip(edge_index[0], edge_index[1]))}
    for b in range(len(ptr) - 1):
        for s in range(int(ptr[b]), int(ptr[b + 1])):
            ms = torch.tensor(masks[int(z[s])])
            for d in range(int(ptr[b]), int(ptr[b + 1])):
                md = torch.tensor(masks[int(z[d])])
                blk = diag[s] if s == d else nondiag[look[(d, s)]]
                H[optr[d]:optr[d + 1], optr[s]:optr[s + 1]] = blk[md][:, ms]
    return H + H.T if symmetrize else H


def hamiltonian_loss(pred, target, mask):
    diff = pred - target
    mse = torch.mean(diff ** 2) * (pred.numel() / mask.sum())
    mae = torch.mean(torch.abs(diff)) * (pred.numel() / mask.sum())
    return torch.sqrt(mse) + mae


def masked_mae(pred, target):
    return torch.abs(pred - target).sum() / torch.count_nonzero(target)
