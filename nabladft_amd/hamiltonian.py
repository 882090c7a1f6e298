"""Hamiltonian block assembly for QHNet-style models on MI355X (transpose index of a13).

Mirrors, with the reference's names and argument meaning:
  orbital_masks(orbitals)              <-> QHNet._get_mask              (nablaDFT/qhnet/qhnet.py:323-342)
  transpose_index(ptr)                 <-> tail of QHNet.build_graph    (qhnet.py:273-283)
  BlockAssembler.build_final_matrix    <-> QHNet.build_final_matrix     (qhnet.py:293-321)  [+ H + H^T, qhnet.py:237]
  HamiltonianLoss                      <-> nablaDFT.qhnet.loss.HamiltonianLoss (qhnet/loss.py:5-16)
  masked_mae                           <-> MaskedMeanAbsoluteError.update (qhnet/masked_mae.py:12-20)

The reference walks every (molecule, src atom, dst atom) in Python with ``.item()`` and ``torch.where`` per block (seconds per
batch); here one table-driven gather kernel writes the result (csrc/hblock.hip).  The native result is PACKED (the diagonal
blocks, molecule after molecule); ``dense=True`` materialises the block_diag matrix the reference returns.  GPU only.
"""
import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib


def orbital_masks(orbitals: Dict[int, List[int]]) -> Tuple[Dict[int, List[int]], int, int, int]:
    """Slots of the padded per-atom block used by each atom type: s_max s-slots, then 3*p_max p-slots, then 5*d_max d-slots, sized by
    the largest atom type; an atom uses the first slots of each kind (qhnet.py:323-342)."""
    max_z = max(orbitals.keys())
    _, counts = np.unique(orbitals[max_z], return_counts=True)
    s_max, p_max, d_max = (int(c) for c in counts)
    starts, mult = [0, s_max, s_max + 3 * p_max], [1, 3, 5]
    masks = {}
    for zt, orb in orbitals.items():
        _, cnt = np.unique(orb, return_counts=True)
        m: List[int] = []
        for l, c in enumerate(cnt):
            m.extend(range(starts[l], starts[l] + int(c) * mult[l]))
        masks[int(zt)] = m
    return masks, s_max, p_max, d_max


def full_pair_index(ptr: torch.Tensor) -> torch.Tensor:
    """[2, P] ordered pairs of every molecule in the order of radius_graph on a full graph (centre = row 1 ascending, the other atoms
    ascending in row 0) -- what ``data.full_edge_index`` holds for QHNet (qhnet.py:262, 296).  Vectorised, on ptr's device."""
    n = (ptr[1:] - ptr[:-1]).long()
    cnt = n * (n - 1)
    P = int(cnt.sum())
    mol = torch.repeat_interleave(torch.arange(n.numel(), device=ptr.device), cnt, output_size=P)
    start = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])[:-1]
    k = torch.arange(P, device=ptr.device) - start[mol]
    nm = n[mol]
    c = k // (nm - 1)
    j = k % (nm - 1)
    j = j + (j >= c).long()
    return torch.stack([j + ptr[mol], c + ptr[mol]])


def transpose_index(ptr: torch.Tensor) -> torch.Tensor:
    """Index of the reverse pair for the full-graph order (qhnet.py:273-283): t = a*(n-1) + b - [a < b] with (a, b) = (row0, row1)."""
    ei = full_pair_index(ptr)
    n = (ptr[1:] - ptr[:-1]).long()
    cnt = n * (n - 1)
    mol = torch.repeat_interleave(torch.arange(n.numel(), device=ptr.device), cnt, output_size=ei.shape[1])
    start = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])[:-1]
    a, b = ei[0] - ptr[mol], ei[1] - ptr[mol]
    return a * (n[mol] - 1) + b - (a < b).long() + start[mol]


class _Plan:
    """Per-batch index tables (device)."""


class _Assemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, asm, plan, symmetrize, diag, nondiag):
        lib = _lib.load()
        dev = diag.device
        d32 = diag.detach().to(torch.float32).contiguous()
        n32 = nondiag.detach().to(torch.float32).contiguous()
        out = torch.empty(plan.total, device=dev, dtype=torch.float32)
        _lib.check(lib.nq_hblock_assemble(_lib.ptr(d32), _lib.ptr(n32), _lib.ptr(plan.mol_ptr), _lib.ptr(plan.pair_base), _lib.ptr(plan.pack_ptr),
                                          _lib.ptr(plan.mol_orb_ptr), _lib.ptr(plan.orb_atom), _lib.ptr(plan.orb_slot), _lib.ptr(plan.look), plan.B,
                                          asm.S, int(symmetrize), plan.total, _lib.ptr(out), _lib.ptr(plan.err), _lib.stream_ptr()))
        ctx.asm, ctx.plan, ctx.symmetrize, ctx.shapes = asm, plan, symmetrize, (diag.shape, nondiag.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        asm, plan = ctx.asm, ctx.plan
        g = g.to(torch.float32).contiguous()
        gd = torch.empty(ctx.shapes[0], device=g.device, dtype=torch.float32)
        gn = torch.empty(ctx.shapes[1], device=g.device, dtype=torch.float32)
        _lib.check(lib.nq_hblock_assemble_backward(_lib.ptr(g), _lib.ptr(plan.z), _lib.ptr(plan.atom_mol), _lib.ptr(plan.orb_ptr), _lib.ptr(plan.pack_ptr),
                                                   _lib.ptr(plan.mol_orb_ptr), _lib.ptr(asm._inv(g.device)), _lib.ptr(plan.e_dst), _lib.ptr(plan.e_src),
                                                   plan.N, plan.P, asm.S, int(ctx.symmetrize), _lib.ptr(gd), _lib.ptr(gn), _lib.stream_ptr()))
        return None, None, None, gd, gn


class BlockAssembler:
    """Holds the orbital-mask tables of one basis set (``orbitals`` as in config/model/qhnet.yaml:14-22) and assembles batches."""

    def __init__(self, orbitals: Dict[int, List[int]]):
        self.masks, self.s_max, self.p_max, self.d_max = orbital_masks(orbitals)
        self.S = self.s_max + 3 * self.p_max + 5 * self.d_max
        zt = max(self.masks) + 1
        table = torch.zeros(zt, self.S, dtype=torch.int32)
        count = torch.zeros(zt, dtype=torch.int32)
        inv = torch.full((zt, self.S), -1, dtype=torch.int32)
        for zz, m in self.masks.items():
            table[zz, :len(m)] = torch.tensor(m, dtype=torch.int32)
            count[zz] = len(m)
            inv[zz, torch.tensor(m, dtype=torch.long)] = torch.arange(len(m), dtype=torch.int32)
        self._host = (table, count, inv)
        self._dev = {}

    def _tables(self, device):
        if device not in self._dev:
            self._dev[device] = tuple(t.to(device) for t in self._host)
        return self._dev[device]

    def _inv(self, device):
        return self._tables(device)[2]

    def plan(self, z: torch.Tensor, ptr: torch.Tensor, full_edge_index: torch.Tensor) -> _Plan:
        """Index tables of one batch: orbital -> (atom, slot), ordered pair -> pair-block index, packed offsets."""
        if not z.is_cuda:
            raise RuntimeError("nabladft_amd.hamiltonian runs on MI355X only (no CPU fallback): move the batch to cuda")
        lib = _lib.load()
        dev = z.device
        table, count, _ = self._tables(dev)
        p = _Plan()
        z64 = z.long()
        if int(z64.max()) >= count.numel() or bool((count[z64] == 0).any()):
            raise KeyError("atomic number without an entry in `orbitals`")        # the reference raises KeyError on orbital_mask[z]
        p.z = z.to(torch.int32).contiguous()
        p.N, p.B = int(z.shape[0]), int(ptr.numel() - 1)
        p.mol_ptr = ptr.to(device=dev, dtype=torch.int32).contiguous()
        n = (ptr[1:] - ptr[:-1]).to(dev).long()
        p.atom_mol = torch.repeat_interleave(torch.arange(p.B, device=dev), n, output_size=p.N).to(torch.int32)
        norb = count[z64].long()
        p.orb_ptr = torch.cat([norb.new_zeros(1), norb.cumsum(0)])
        p.mol_orb_ptr = p.orb_ptr[ptr.to(dev).long()].contiguous()
        m = p.mol_orb_ptr[1:] - p.mol_orb_ptr[:-1]
        p.pack_ptr = torch.cat([m.new_zeros(1), (m * m).cumsum(0)])
        p.total, p.m_total = int(p.pack_ptr[-1]), int(p.mol_orb_ptr[-1])
        nn2 = n * n
        p.pair_base = torch.cat([nn2.new_zeros(1), nn2.cumsum(0)])
        look_count = int(p.pair_base[-1])
        p.e_dst, p.e_src = full_edge_index[0].to(dev).long().contiguous(), full_edge_index[1].to(dev).long().contiguous()
        p.P = int(p.e_dst.shape[0])
        p.orb_atom = torch.empty(p.m_total, device=dev, dtype=torch.int32)
        p.orb_slot = torch.empty(p.m_total, device=dev, dtype=torch.int32)
        p.look = torch.empty(max(look_count, 1), device=dev, dtype=torch.int32)
        p.err = torch.zeros(1, device=dev, dtype=torch.int32)
        _lib.check(lib.nq_hblock_tables(_lib.ptr(p.z), _lib.ptr(p.atom_mol), _lib.ptr(p.mol_ptr), p.N, p.B, _lib.ptr(p.orb_ptr), _lib.ptr(p.pair_base),
                                        _lib.ptr(p.e_dst), _lib.ptr(p.e_src), p.P, _lib.ptr(table), _lib.ptr(count), self.S, _lib.ptr(p.orb_atom),
                                        _lib.ptr(p.orb_slot), _lib.ptr(p.look), look_count, _lib.ptr(p.err), _lib.stream_ptr()))
        return p

    def assemble(self, plan: _Plan, diagonal_matrix: torch.Tensor, non_diagonal_matrix: torch.Tensor, symmetrize: bool = True) -> torch.Tensor:
        """Packed result [plan.total] (differentiable w.r.t. both block tensors)."""
        out = _Assemble.apply(self, plan, symmetrize, diagonal_matrix, non_diagonal_matrix)
        return out

    def check(self, plan: _Plan):
        """Raises if the pair list was not the full graph of every molecule (the reference fails on ``torch.where(...)[0].item()``)."""
        code = int(plan.err.item())
        if code:
            raise IndexError("full_edge_index does not hold every ordered atom pair of every molecule" if code & 2 else "a pair joins two molecules")

    def to_dense(self, plan: _Plan, packed: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        dense = torch.empty(plan.m_total, plan.m_total, device=packed.device, dtype=torch.float32)
        _lib.check(lib.nq_hblock_packed_dense(_lib.ptr(packed), _lib.ptr(dense), _lib.ptr(plan.pack_ptr), _lib.ptr(plan.mol_orb_ptr), plan.B, plan.total,
                                              plan.m_total, 1, _lib.stream_ptr()))
        return dense

    def from_dense(self, plan: _Plan, dense: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        packed = torch.empty(plan.total, device=dense.device, dtype=torch.float32)
        d = dense.to(torch.float32).contiguous()
        _lib.check(lib.nq_hblock_packed_dense(_lib.ptr(packed), _lib.ptr(d), _lib.ptr(plan.pack_ptr), _lib.ptr(plan.mol_orb_ptr), plan.B, plan.total,
                                              plan.m_total, 0, _lib.stream_ptr()))
        return packed

    def pack_targets(self, plan: _Plan, hamiltonians: List[np.ndarray]) -> torch.Tensor:
        """``batch.hamiltonian`` (list of per-molecule numpy matrices, dataset/pyg_datasets.py:198-222) -> packed device tensor."""
        flat = np.concatenate([np.asarray(h, dtype=np.float32).reshape(-1) for h in hamiltonians])
        if flat.size != plan.total:
            raise ValueError(f"targets hold {flat.size} elements, the batch has {plan.total}")
        return torch.from_numpy(flat).to(plan.z.device)

    def build_final_matrix(self, data, diagonal_matrix, non_diagonal_matrix, symmetrize: bool = False) -> torch.Tensor:
        """Drop-in for QHNet.build_final_matrix(data, diag, nondiag) (qhnet.py:293-321): dense block-diagonal [M, M].  ``symmetrize=True``
        also applies qhnet.py:237 (H + H^T).  Not differentiable through the dense copy; training uses ``assemble`` + HamiltonianLoss."""
        plan = self.plan(data.z, data.ptr, data.full_edge_index)
        packed = self.assemble(plan, diagonal_matrix, non_diagonal_matrix, symmetrize)
        self.check(plan)
        return self.to_dense(plan, packed.detach())


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.load()
        p = pred.detach().to(torch.float32).contiguous().view(-1)
        t = target.detach().to(torch.float32).contiguous().view(-1)
        stats = torch.empty(3, device=p.device, dtype=torch.float32)
        grad = torch.empty_like(p)
        scratch = torch.empty(512, device=p.device, dtype=torch.float64)
        _lib.check(lib.nq_hamiltonian_loss(_lib.ptr(p), _lib.ptr(t), p.numel(), 1.0, _lib.ptr(stats), _lib.ptr(grad), _lib.ptr(scratch), _lib.stream_ptr()))
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None


class HamiltonianLoss(nn.Module):
    """rmse + mae over the block-diagonal support (qhnet/loss.py:9-16).  ``pred`` / ``target`` are PACKED tensors (the diagonal blocks);
    for those ``mask.sum() == numel`` so the reference's ``numel / mask.sum()`` rescaling of the dense means is already applied."""

    packed = True          # QHNetLightning hands packed predictions / targets to losses that declare this

    def forward(self, pred, target, mask=None):
        if not pred.is_cuda:
            raise RuntimeError("nabladft_amd.hamiltonian.HamiltonianLoss runs on MI355X only")
        return _LossFn.apply(pred, target)


def masked_mae(pred_packed: torch.Tensor, target_packed: torch.Tensor) -> torch.Tensor:
    """sum |pred - target| / count_nonzero(target)  (MaskedMeanAbsoluteError: qhnet/masked_mae.py:12-20)."""
    lib = _lib.load()
    p, t = pred_packed.detach().float().contiguous().view(-1), target_packed.detach().float().contiguous().view(-1)
    stats = torch.empty(3, device=p.device, dtype=torch.float32)
    scratch = torch.empty(512, device=p.device, dtype=torch.float64)
    _lib.check(lib.nq_hamiltonian_loss(_lib.ptr(p), _lib.ptr(t), p.numel(), 1.0, _lib.ptr(stats), None, _lib.ptr(scratch), _lib.stream_ptr()))
    return stats[2] / torch.count_nonzero(t)


# ---- PhiSNet: irreducible representations -> matrix (the assembly part) -----------------------------------------
def compute_matrix_irreps(orbitals_i, orbitals_j, irreps, number_L):
    """Same contract as NeuralNetwork.compute_matrix_irreps (phisnet/nn/neural_network.py:610-621): assigns the next free feature index of
    order L to every new key (z_i, z_j, n_i, n_j, L)."""
    for n_i, (z_i, l_i) in enumerate(orbitals_i):
        for n_j, (z_j, l_j) in enumerate(orbitals_j):
            for L in range(abs(l_i - l_j), l_i + l_j + 1):
                key = (z_i, z_j, n_i, n_j, L)
                if key not in irreps:
                    irreps[key] = number_L[L]
                    number_L[L] += 1
    return irreps, number_L


class _IrAssemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, asm, plan, symmetrize, unit_diagonal, f_ii, f_ij):
        lib = _lib.load()
        dev = f_ii.device
        a, b = f_ii.detach().to(torch.float32).contiguous(), f_ij.detach().to(torch.float32).contiguous()
        t = asm._tables(dev)
        out = torch.empty(plan.total, device=dev, dtype=torch.float32)
        _lib.check(lib.nq_irreps_assemble(
            _lib.ptr(a), _lib.ptr(b), _lib.ptr(plan.z), _lib.ptr(plan.mol_ptr), _lib.ptr(plan.pair_base), _lib.ptr(plan.pack_ptr), _lib.ptr(plan.mol_orb_ptr),
            _lib.ptr(plan.orb_ptr), _lib.ptr(plan.orb_atom), _lib.ptr(plan.orb_slot), _lib.ptr(plan.look), plan.B, a.shape[1], a.shape[2],
            _lib.ptr(t["tz"]), _lib.ptr(t["sh_n"]), _lib.ptr(t["sh_l"]), _lib.ptr(t["sh_m"]), _lib.ptr(t["sh_off"]), _lib.ptr(t["idx_ii"]), _lib.ptr(t["idx_ij"]),
            _lib.ptr(t["cgt"]), asm.T, asm.S, int(symmetrize), int(unit_diagonal), plan.total, _lib.ptr(out), _lib.ptr(plan.err), _lib.stream_ptr()))
        ctx.asm, ctx.plan, ctx.flags, ctx.shapes = asm, plan, (symmetrize, unit_diagonal), (f_ii.shape, f_ij.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        asm, plan = ctx.asm, ctx.plan
        t = asm._tables(g.device)
        g = g.to(torch.float32).contiguous()
        g_ii = torch.empty(ctx.shapes[0], device=g.device, dtype=torch.float32)
        g_ij = torch.empty(ctx.shapes[1], device=g.device, dtype=torch.float32)
        inv_ii, inv_ij = asm._inverse(ctx.shapes[0][2], g.device)
        _lib.check(lib.nq_irreps_assemble_backward(
            _lib.ptr(g), _lib.ptr(plan.z), _lib.ptr(plan.atom_mol), _lib.ptr(plan.e_dst), _lib.ptr(plan.e_src), _lib.ptr(plan.pack_ptr), _lib.ptr(plan.mol_orb_ptr),
            _lib.ptr(plan.orb_ptr), _lib.ptr(inv_ii), _lib.ptr(inv_ij), plan.N, plan.P, ctx.shapes[0][1], ctx.shapes[0][2], _lib.ptr(t["tz"]), _lib.ptr(t["sh_n"]),
            _lib.ptr(t["sh_l"]), _lib.ptr(t["sh_m"]), _lib.ptr(t["sh_off"]), _lib.ptr(t["cgt"]), asm.T, asm.S, int(ctx.flags[0]), int(ctx.flags[1]),
            _lib.ptr(g_ii), _lib.ptr(g_ij), _lib.stream_ptr()))
        return None, None, None, None, g_ii, g_ij


class IrrepsAssembler:
    """Builds Hamiltonian / overlap matrices from PhiSNet's irreducible-representation features (the loops of NeuralNetwork.forward at
    phisnet/nn/neural_network.py:859-918 + generate_matrix_from_irreps / matrix_block, :636-706) in one kernel launch.

    ``atom2orbitals``: {Z: ((Z, l), ...)} shells of every element (l <= 2);  ``irreps_ii`` / ``irreps_ij``: the model's index dictionaries
    {(z_i, z_j, n_i, n_j, L): feature index} (``compute_matrix_irreps``);  ``clebsch_gordan``: the model's CG provider (its sign convention is
    used as is).  Features: f_ii [N, (Lout+1)^2, Fo] per atom, f_ij [P, (Lout+1)^2, Fo] per ordered pair (idx_i, idx_j) -- the concatenation
    over L of the reference's lists ``fii_*[L]`` / ``fij_*[L]``.  Result: packed block-diagonal matrix (see BlockAssembler)."""

    MAXORB, NL = 32, 5

    def __init__(self, atom2orbitals, irreps_ii, irreps_ij, clebsch_gordan):
        self.types = sorted(int(zz) for zz in atom2orbitals)
        self.T = len(self.types)
        tindex = {zz: i for i, zz in enumerate(self.types)}
        self.S = max(len(v) for v in atom2orbitals.values())
        if self.S > self.MAXORB or max(l for v in atom2orbitals.values() for _, l in v) > 2:
            raise NotImplementedError("IrrepsAssembler: up to 32 shells per atom with l <= 2 are built")
        zt = max(self.types) + 1
        tz = torch.full((zt,), -1, dtype=torch.int32)
        sh_n = torch.zeros(self.T, self.MAXORB, dtype=torch.int32)
        sh_l, sh_m, sh_off = torch.zeros_like(sh_n), torch.zeros_like(sh_n), torch.zeros_like(sh_n)
        self.norb = {}
        count = torch.zeros(zt, dtype=torch.int32)
        for zz, shells in atom2orbitals.items():
            ti = tindex[int(zz)]
            tz[int(zz)] = ti
            o = 0
            for n, (_, l) in enumerate(shells):
                sh_off[ti, n] = o
                for m in range(2 * l + 1):
                    if o >= self.MAXORB:
                        raise NotImplementedError("IrrepsAssembler: more than 32 orbitals per atom")
                    sh_n[ti, o], sh_l[ti, o], sh_m[ti, o] = n, l, m
                    o += 1
            self.norb[int(zz)] = o
            count[int(zz)] = o
        idx_ii = torch.full((self.T, self.S, self.S, self.NL), -1, dtype=torch.int32)
        idx_ij = torch.full((self.T, self.T, self.S, self.S, self.NL), -1, dtype=torch.int32)
        for (zi, zj, ni, nj, L), v in irreps_ii.items():
            if zi == zj and int(zi) in tindex:
                idx_ii[tindex[int(zi)], ni, nj, L] = v
        for (zi, zj, ni, nj, L), v in irreps_ij.items():
            if int(zi) in tindex and int(zj) in tindex:
                idx_ij[tindex[int(zi)], tindex[int(zj)], ni, nj, L] = v
        cgt = torch.zeros(3, 3, self.NL, 5, 5, 9, dtype=torch.float32)
        for li in range(3):
            for lj in range(3):
                for L in range(abs(li - lj), li + lj + 1):
                    c = torch.as_tensor(clebsch_gordan(li, lj, L)).detach().to(torch.float32).cpu() * (2 * L + 1) ** 0.5
                    cgt[li, lj, L, :2 * li + 1, :2 * lj + 1, :2 * L + 1] = c
        self._host = dict(tz=tz, sh_n=sh_n, sh_l=sh_l, sh_m=sh_m, sh_off=sh_off, idx_ii=idx_ii, idx_ij=idx_ij, cgt=cgt)
        self._count = count
        self._identity = torch.arange(self.MAXORB, dtype=torch.int32).repeat(zt, 1)
        self._dev, self._inv = {}, {}
        # the per-batch tables (orbital -> atom / local orbital, pair lookup) are those of the QHNet assembler with identity "masks"
        self._plan_helper = BlockAssembler.__new__(BlockAssembler)
        self._plan_helper.S = self.MAXORB
        self._plan_helper._host = (self._identity, count, torch.full((zt, self.MAXORB), -1, dtype=torch.int32))
        self._plan_helper._dev = {}

    def _tables(self, device):
        if device not in self._dev:
            self._dev[device] = {k: v.to(device).contiguous() for k, v in self._host.items()}
        return self._dev[device]

    def _inverse(self, Fo, device):
        key = (Fo, device)
        if key not in self._inv:
            inv_ii = torch.full((self.T, self.NL, Fo), -1, dtype=torch.int32)
            inv_ij = torch.full((self.T, self.T, self.NL, Fo), -1, dtype=torch.int32)
            ii, ij = self._host["idx_ii"], self._host["idx_ij"]
            for t in range(self.T):
                for ni in range(self.S):
                    for nj in range(self.S):
                        for L in range(self.NL):
                            v = int(ii[t, ni, nj, L])
                            if 0 <= v < Fo:
                                inv_ii[t, L, v] = ni * self.S + nj
                            for t2 in range(self.T):
                                w = int(ij[t, t2, ni, nj, L])
                                if 0 <= w < Fo:
                                    inv_ij[t, t2, L, w] = ni * self.S + nj
            self._inv[key] = (inv_ii.to(device), inv_ij.to(device))
        return self._inv[key]

    def plan(self, z, ptr, idx_i, idx_j):
        """Per-batch tables; (idx_i, idx_j) = the ordered pairs whose features are the rows of f_ij (fill_idx, neural_network.py:515-547)."""
        return self._plan_helper.plan(z, ptr, torch.stack([idx_i, idx_j]))

    def assemble(self, plan, f_ii, f_ij, symmetrize=True, unit_diagonal=False):
        return _IrAssemble.apply(self, plan, symmetrize, unit_diagonal, f_ii, f_ij)

    def check(self, plan):
        code = int(plan.err.item())
        if code & 4:
            raise KeyError("an (element pair, shells, L) combination has no entry in irreps_ii / irreps_ij")       # the reference raises KeyError
        if code:
            raise IndexError("the pair list does not hold every ordered atom pair of every molecule" if code & 2 else "a pair joins two molecules")

    to_dense = BlockAssembler.to_dense
    from_dense = BlockAssembler.from_dense
