"""eSCN on the HIP kernels of csrc/escn.hip and the fp32 MFMA GEMMs -- host-side mirror of the reference's ``nablaDFT.escn.eSCN``
(escn/escn.py:36-1003; config/model/escn-oc.yaml): same constructor arguments, same parameter / buffer names (``state_dict`` keys and order equal),
same outputs ``(energy [B], forces [N, 3])``.

Constants the reference takes from e3nn (not installed anywhere) or from its data file are computed here from first principles:
  * real spherical harmonics in e3nn's convention (degree-(l+1) from degree-l by the 3j coupling with Y_1, this package's own Racah 3j tensors, cg.py);
  * the matrices J_l of so3.py:18-21 (``Jd.pt``) as the representation of the fixed orthogonal map (x, y, z) -> (-y, -x, z) on the degree-l harmonics --
    checked against the reference's data file for l <= 6 in the tests;
  * the S2 grids of SO3_Grid (so3.py:428-497): equiangular latitudes, integral-normalised harmonics, Driscoll-Healy quadrature weights
    (from_grid(to_grid(x)) = x to 1e-15); the exact grid / weights of e3nn 0.5.1's ToS2Grid / FromS2Grid are restated from memory: PARITY UNPINNED for
    them, like the e3nn arithmetic under QHNet;
  * the coefficient orderings of CoefficientMapping (so3.py:23-118).
Edge frames use a deterministic helper axis where the reference draws a random vector (escn.py:447): the SO(2) convolution commutes with rotations
about the edge, so the model output is the same function; ``forward(data, edge_rot_mat=...)`` accepts given frames (parity tests).
Only the options of the yaml are built (gaussian distance expansion, use_grid, one resolution, no periodic cells).  No CPU path.
"""
import ctypes as C
import math
from typing import List

import numpy as np
import torch

from . import _lib, cg
from . import gemnet_oc as _gemnet
from .gemnet_oc import _DenseFn, _MulFn, _SO2GatedPairFn, _SegSumFn, _gather_raw, _new, _segsum_raw, _st, fused_pairs_available, lin
from .qhnet import _ActFn, _LinearBiasFn, _MatmulFn, _f32


# ---- constants -------------------------------------------------------------------------------------------------------------------------------
def _sh_e3nn(lmax, xyz):
    """Real harmonics in e3nn's convention, 'integral' normalisation, as homogeneous polynomials of xyz [n, 3] (float64): list over l of [n, 2l+1]."""
    xyz = np.asarray(xyz, dtype=np.float64)
    pole = np.array([0.0, 1.0, 0.0])
    ys, ref = [np.ones((xyz.shape[0], 1)), xyz.copy()], [np.ones(1), pole.copy()]
    for l in range(1, lmax):
        w = np.asarray(cg.wigner_3j_e3nn(l, 1, l + 1))
        y = np.einsum("ijk,ni,nj->nk", w, ys[l], xyz)
        r = np.einsum("ijk,i,j->k", w, ref[l], pole)
        c = np.linalg.norm(r)                               # a harmonic polynomial built from unit-norm pieces has constant norm on the sphere
        ys.append(y / c)
        ref.append(r / c)
    return [ys[l] * math.sqrt((2 * l + 1) / (4 * math.pi)) for l in range(lmax + 1)]


_J_MAP = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])


def j_matrices(lmax):
    """J_l with Y_l(g x) = J_l Y_l(x) for g = _J_MAP (the l = 1 entry of the reference's Jd.pt IS this matrix); least squares on random points."""
    rng = np.random.Generator(np.random.PCG64(0))
    x = rng.normal(size=(4 * (2 * lmax + 1) + 16, 3))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    Y, Yg = _sh_e3nn(lmax, x), _sh_e3nn(lmax, x @ _J_MAP.T)
    out = []
    for l in range(lmax + 1):
        Jt, *_ = np.linalg.lstsq(Y[l], Yg[l], rcond=None)
        J = Jt.T
        J[np.abs(J) < 1e-12] = 0.0
        out.append(J)
    return out


def s2_grids(lmax, mmax):
    """(to_grid [B*A, (lmax+1)^2], from_grid [B*A, (lmax+1)^2]) of SO3_Grid(lmax, mmax) (so3.py:436-470), all (l, m) columns in e3nn order."""
    nb = 2 * (lmax + 1)
    na = 2 * (mmax + 1) + 1 if lmax == mmax else 2 * mmax + 1
    beta = (np.arange(nb) + 0.5) / nb * math.pi
    alpha = np.arange(na) / na * 2 * math.pi
    b, a = np.meshgrid(beta, alpha, indexing="ij")
    xyz = np.stack([np.sin(b) * np.sin(a), np.cos(b), np.sin(b) * np.cos(a)], axis=-1).reshape(-1, 3)
    T = np.concatenate(_sh_e3nn(lmax, xyz), axis=1)
    half = nb // 2
    k = np.arange(half)
    w = np.array([(2.0 / half) * math.sin(t) * np.sum(np.sin((2 * k + 1) * t) / (2 * k + 1)) for t in beta])
    w *= 2.0 / w.sum()
    F = T * np.repeat(w * (2 * math.pi / na), na)[:, None]
    return T, F


def sphere_points(n):
    """CalcSpherePoints (escn/sampling.py:15-36): Fibonacci points weighted by their local density."""
    golden = (1 + 5 ** 0.5) / 2
    i = torch.arange(n).view(-1, 1)
    theta = 2 * math.pi * i / golden
    phi = torch.arccos(1 - 2 * (i + 0.5) / n)
    pts = torch.cat([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=1)
    d = ((pts.view(1, -1, 3) - pts.view(-1, 1, 3)) ** 2).sum(dim=2)
    s = 1.0 / torch.exp(-d / (0.5 * 0.3)).sum(dim=1)
    s = n * s / s.sum()
    return pts * s.view(-1, 1)


class CoefficientOrder:
    """The index bookkeeping of CoefficientMapping (so3.py:23-118) for one resolution."""

    def __init__(self, lmax, mmax):
        self.lmax, self.mmax = lmax, mmax
        full = [(l, m) for l in range(lmax + 1) for m in range(-l, l + 1)]
        self.n_full = len(full)
        self.red_l_primary = [l * l + l + m for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]      # coefficient_idx(lmax, mmax)
        order = [(l, 0) for l in range(lmax + 1)]
        self.m_size = [lmax + 1]
        for m in range(1, mmax + 1):
            order += [(l, m) for l in range(m, lmax + 1)] + [(l, -m) for l in range(m, lmax + 1)]
            self.m_size.append(lmax - m + 1)
        self.m_primary = order                                           # (l, m) of every reduced coefficient in m-primary order
        self.red_m_primary = [l * l + l + m for l, m in order]           # its index in the full l-primary list
        self.n_red = len(order)


# ---- autograd wrappers -----------------------------------------------------------------------------------------------------------------------------
_LDS_FLOATS = 15 * 1024          # per-call LDS budget of nq_rowop (60 kB)


def _rowmm_ok(r_stride, I, NSS, Cc, transpose):
    """Mirror of csrc/escn.hip: rowmm_plan -- a shared matrix (r_stride 0) with whole 32-channel groups runs on the matrix cores with the WHOLE matrix
    resident in LDS (no splitting of its rows); same budget arithmetic as the C side."""
    if r_stride != 0 or Cc % 32:
        return False
    M, K = (NSS, I) if transpose else (I, NSS)
    MT, Kp = (M + 31) // 32, (K + 1) & ~1
    cs = 128 if Cc % 128 == 0 else (64 if Cc % 64 == 0 else 32)
    while cs > 32 and (MT * (cs // 32) + 3) // 4 > 6:
        cs //= 2
    if (MT * (cs // 32) + 3) // 4 > 6:
        return False
    return 4 * (((MT * 32 * (Kp + 1) + 3) & ~3) + max(Kp, MT * 32) * cs) <= 150 * 1024


def _rowop(R, r_stride, X, x_stride, index, n, I, NSS, Cc, transpose):
    """nq_rowop.  Shared matrices go to the MFMA row operator in one call; per-row matrices use the LDS kernel with the matrix rows split so that
    (matrix chunk + row block) fits its LDS budget: forward chunks write disjoint output rows, transposed chunks accumulate."""
    lib = _lib.load()
    out = _new(n, (NSS if transpose else I) * Cc, like=X)
    idx = None if index is None else _lib.ptr(index)
    if _rowmm_ok(r_stride, I, NSS, Cc, transpose):
        _lib.check(lib.nq_rowop(_lib.ptr(R), r_stride, _lib.ptr(X), x_stride, idx, _lib.ptr(out), (NSS if transpose else I) * Cc, n, I, NSS, Cc, int(transpose), 0, _st()))
        return out
    step = max(1, (_LDS_FLOATS // (NSS + Cc)) if transpose else (_LDS_FLOATS - NSS * Cc) // NSS)
    for a in range(0, I, step):
        rows = min(step, I - a)
        Rp = R.data_ptr() + 4 * a * NSS
        if transpose:
            _lib.check(lib.nq_rowop(Rp, r_stride, X.data_ptr() + 4 * a * Cc, x_stride, idx, _lib.ptr(out), NSS * Cc, n, rows, NSS, Cc, 1, int(a > 0), _st()))
        else:
            _lib.check(lib.nq_rowop(Rp, r_stride, _lib.ptr(X), x_stride, idx, out.data_ptr() + 4 * a * Cc, I * Cc, n, rows, NSS, Cc, 0, 0, _st()))
    return out


def _rowop_blocks(R, r_stride, other, index, seg_side, rows, blocks, n, I, NSS, Cc, transpose):
    """One launch of the row operator with one side in per-block tensors (csrc/escn.hip: nq_rowop_blocks)."""
    k = len(rows)
    seg_rows = (C.c_int32 * k)(*rows)
    seg_ptrs = (C.c_void_p * k)(*[b.data_ptr() for b in blocks])
    _lib.check(_lib.load().nq_rowop_blocks(_lib.ptr(R), r_stride, _lib.ptr(other), other.shape[1], None if index is None else _lib.ptr(index), seg_side, k, seg_rows,
                                           seg_ptrs, n, I, NSS, Cc, int(transpose), 0, _st()))


class _BlocksOutFn(torch.autograd.Function):
    """(out_b) = op(R, X) with the segmented side as the OUTPUT: rotation into the edge frame (rows of R = the m-blocks) and from-grid (columns of R = the
    m-blocks) write the blocks the SO(2) layers consume as separate contiguous tensors, in one pass over X.  Adjoint: _BlocksInFn, other orientation."""

    @staticmethod
    def forward(ctx, X, R, r_stride, I, NSS, Cc, transpose, seg_side, rows, index, inverse, n):
        X = _f32(X)
        ctx.meta = (R, r_stride, I, NSS, Cc, transpose, seg_side, rows, index, inverse, n, X.shape[0])
        outs = [_new(n, r * Cc, like=X) for r in rows]
        _rowop_blocks(R, r_stride, X, index, seg_side, rows, outs, n, I, NSS, Cc, transpose)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        R, r_stride, I, NSS, Cc, transpose, seg_side, rows, index, inverse, n, n_src = ctx.meta
        gs = [_f32(g) for g in gs]
        dX = _new(n, (I if transpose else NSS) * Cc, like=gs[0])
        _rowop_blocks(R, r_stride, dX, None, seg_side, rows, gs, n, I, NSS, Cc, not transpose)
        if index is not None:
            order, ptr = inverse
            dX = _segsum_raw(dX, order, ptr, n_src, dX.shape[1])
        return (dX,) + (None,) * 11


class _BlocksInFn(torch.autograd.Function):
    """out = op(R, (X_b)) with the segmented side as the INPUT (rotate back, to-grid)."""

    @staticmethod
    def forward(ctx, R, r_stride, I, NSS, Cc, transpose, seg_side, rows, n, *Xs):
        Xs = [_f32(x) for x in Xs]
        ctx.meta = (R, r_stride, I, NSS, Cc, transpose, seg_side, rows, n)
        out = _new(n, (NSS if transpose else I) * Cc, like=Xs[0])
        _rowop_blocks(R, r_stride, out, None, seg_side, rows, Xs, n, I, NSS, Cc, transpose)
        return out

    @staticmethod
    def backward(ctx, g):
        R, r_stride, I, NSS, Cc, transpose, seg_side, rows, n = ctx.meta
        g = _f32(g)
        outs = [_new(n, r * Cc, like=g) for r in rows]
        _rowop_blocks(R, r_stride, g, None, seg_side, rows, outs, n, I, NSS, Cc, not transpose)
        return (None,) * 9 + tuple(outs)


def s2_activation_fusable(T, rows, Cc):
    """Mirror of csrc/escn.hip: nq_s2_activation_blocks' requirements (whole groups of 64 channels, <= 32 coefficients, <= 96 grid points)."""
    return Cc % 64 == 0 and sum(rows) <= 32 and T.shape[0] <= 96


class _S2ActBlocksFn(torch.autograd.Function):
    """(y_b) = from_grid(SiLU(to_grid((x_b)))) on per-m-block tensors in ONE kernel (csrc/escn.hip: k_s2act): the [E, grid, C] tensor of the point-wise
    activation (so3.py:301-318; 5x the coefficient tensor) is neither written nor kept for the backward, which recomputes it from the saved coefficient blocks."""

    @staticmethod
    def forward(ctx, T, F, rows, n, Cc, *xs):
        xs = [_f32(x) for x in xs]
        outs = [_new(n, r * Cc, like=xs[0]) for r in rows]
        _S2ActBlocksFn._launch(T, F, rows, n, Cc, xs, None, outs, 0)
        ctx.meta = (T, F, rows, n, Cc)
        ctx.save_for_backward(*xs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        T, F, rows, n, Cc = ctx.meta
        xs = ctx.saved_tensors
        gs = [torch.zeros_like(x) if g is None else _f32(g) for g, x in zip(gs, xs)]
        dxs = [_new(n, r * Cc, like=xs[0]) for r in rows]
        _S2ActBlocksFn._launch(T, F, rows, n, Cc, list(xs), gs, dxs, 1)
        return (None,) * 5 + tuple(dxs)

    @staticmethod
    def _launch(T, F, rows, n, Cc, xs, gs, outs, backward):
        k = len(rows)
        arr = lambda ts: (C.c_void_p * k)(*[t.data_ptr() for t in ts])          # noqa: E731
        _lib.check(_lib.load().nq_s2_activation_blocks(_lib.ptr(T), _lib.ptr(F), T.shape[0], sum(rows), Cc, n, k, (C.c_int32 * k)(*rows), arr(xs),
                                                       None if gs is None else arr(gs), arr(outs), backward, _st()))


class _RowFn(torch.autograd.Function):
    """out[o] = R_o X_r(o) (transpose False) or R_o^T X_r(o) (True); R constant (per-row ``r_stride`` = I * NSS or shared 0).  With ``index`` the rows are
    gathered from node storage and the adjoint is summed back over ``inverse`` = (order, ptr) of the index."""

    @staticmethod
    def forward(ctx, X, R, r_stride, I, NSS, Cc, transpose, index, inverse, n):
        X = _f32(X)
        ctx.meta = (R, r_stride, I, NSS, Cc, transpose, index, inverse, n, X.shape[0])
        return _rowop(R, r_stride, X, X.shape[1], index, n, I, NSS, Cc, transpose)

    @staticmethod
    def backward(ctx, g):
        R, r_stride, I, NSS, Cc, transpose, index, inverse, n, n_src = ctx.meta
        g = _f32(g)
        dX = _rowop(R, r_stride, g, g.shape[1], None, n, I, NSS, Cc, not transpose)
        if index is not None:
            order, ptr = inverse
            dX = _segsum_raw(dX, order, ptr, n_src, dX.shape[1])
        return dX, None, None, None, None, None, None, None, None, None


def _silu(x):
    return _ActFn.apply(x, 0, 1.0)


def _rot_args(K, rows, blocks):
    k = len(rows)
    n_red = sum(rows)
    return (k, (C.c_int32 * k)(*rows), (C.c_void_p * k)(*[b.data_ptr() for b in blocks]), (C.c_int32 * n_red)(*K.red_l_host[:n_red]), n_red)


class _RotateFn(torch.autograd.Function):
    """Node embedding x [n, n_full * C] -> the edge-frame embedding as per-m blocks (SO3_Embedding._rotate + _m_primary): ``sources`` = [(index, (order, ptr)), ...]
    gathers whose rotated rows fill consecutive channel slots of the blocks (one source: eSCN; source and target: EquiformerV2's concatenated message,
    transformer_block.py:210-236).  Uses the degree-block structure of the Wigner rows (nq_es_rotate).  Adjoint: nq_es_rotate_back summed over every node's
    edges through the inverse lists, in list order (no atomics, no per-edge intermediate)."""

    @staticmethod
    def forward(ctx, x, G, K, Cc, sources):
        x = _f32(x)
        o, rows, E = K.order, K.block_rows, G.E
        cs = len(sources) * Cc
        blocks = [_new(E, r * cs, like=x) for r in rows]
        k, rows_c, ptrs, red_l, n_red = _rot_args(K, rows, blocks)
        for slot, (index, _) in enumerate(sources):
            _lib.check(_lib.load().nq_es_rotate(_lib.ptr(G.wigner), o.n_red * o.n_full, _lib.ptr(x), o.n_full * Cc, _lib.ptr(index), k, rows_c, ptrs, cs, slot * Cc, E,
                                                red_l, n_red, o.lmax, Cc, _st()))
        ctx.meta = (G, K, Cc, sources, x.shape[0])
        return tuple(blocks)

    @staticmethod
    def backward(ctx, *gs):
        G, K, Cc, sources, n_src = ctx.meta
        o, rows = K.order, K.block_rows
        gs = [_f32(g) for g in gs]
        cs = len(sources) * Cc
        k, rows_c, ptrs, red_l, n_red = _rot_args(K, rows, gs)
        dX = None
        for slot, (_, (order, ptr)) in enumerate(sources):
            part = _new(n_src, o.n_full * Cc, like=gs[0])
            _lib.check(_lib.load().nq_es_rotate_back(_lib.ptr(G.wigner), o.n_red * o.n_full, k, rows_c, ptrs, cs, slot * Cc, _lib.ptr(ptr), _lib.ptr(order), None,
                                                     _lib.ptr(part), n_src, red_l, n_red, o.lmax, Cc, _st()))
            dX = part if dX is None else lin(dX, part)
        return dX, None, None, None, None


class _RotateBackFn(torch.autograd.Function):
    """Per-m blocks of the edge-frame messages -> rotated back and summed over the in-edges of every target atom: [N, n_full * C] (_rotate_inv + _reduce_edge in
    one pass); ``coef_scale`` [n_full] (or None) multiplies the output coefficients (EquiformerV2's rotate_inv rescale).  ``rows`` may cover only the leading
    blocks (the m = 0 block of the edge-degree embedding)."""

    @staticmethod
    def forward(ctx, G, K, Cc, coef_scale, rows, *blocks):
        blocks = [_f32(b) for b in blocks]
        o = K.order
        out = _new(G.N, o.n_full * Cc, like=blocks[0])
        k, rows_c, ptrs, red_l, n_red = _rot_args(K, rows, blocks)
        _lib.check(_lib.load().nq_es_rotate_back(_lib.ptr(G.wigner), o.n_red * o.n_full, k, rows_c, ptrs, Cc, 0, _lib.ptr(G.ptr), None, _lib.ptr(coef_scale),
                                                 _lib.ptr(out), G.N, red_l, n_red, o.lmax, Cc, _st()))
        ctx.meta = (G, K, Cc, coef_scale, rows)
        return out

    @staticmethod
    def backward(ctx, g):
        G, K, Cc, coef_scale, rows = ctx.meta
        o = K.order
        g = _f32(g)
        if coef_scale is not None:
            gs = torch.empty_like(g)
            _lib.check(_lib.load().nq_eq_scale(_lib.ptr(g), None, None, _lib.ptr(coef_scale), g.shape[0], o.n_full, Cc, _lib.ptr(gs), _st()))
            g = gs
        outs = [_new(G.E, r * Cc, like=g) for r in rows]
        k, rows_c, ptrs, red_l, n_red = _rot_args(K, rows, outs)
        _lib.check(_lib.load().nq_es_rotate(_lib.ptr(G.wigner), o.n_red * o.n_full, _lib.ptr(g), o.n_full * Cc, _lib.ptr(G.dst), k, rows_c, ptrs, Cc, 0, G.E, red_l,
                                            n_red, o.lmax, Cc, _st()))
        return (None,) * 5 + tuple(outs)


class _EmbeddingFn(torch.autograd.Function):
    """W[idx]; the adjoint is a chain of fixed-order segment sums ``levels`` = [(order, ptr, n), ...] (no atomics).  Edge embeddings reduce in two steps
    (edges -> their atom -> the atom's element): a one-step sum over ~10 k edges per element would leave one thread per channel walking the whole list."""

    @staticmethod
    def forward(ctx, W, idx, levels):
        W = _f32(W)
        ctx.levels = levels
        return _gather_raw(W, idx, None, idx.numel())

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        for order, ptr, n in ctx.levels:
            g = _segsum_raw(g, order, ptr, n, g.shape[1])
        return g, None, None


def _inverse_lists(idx, n):
    """(order, ptr) int32: positions sorted (stably) by index value, and the start of every value's run."""
    order = torch.sort(idx.long(), stable=True).indices.to(torch.int32)
    ptr = torch.cat([idx.new_zeros(1, dtype=torch.int64), torch.bincount(idx.long(), minlength=n).cumsum(0)]).to(torch.int32)
    return order, ptr


class _Linear(torch.nn.Module):
    """torch.nn.Linear's parameters (reference initialisation) on the MFMA GEMM; ``act`` fuses SiLU."""

    def __init__(self, n_in, n_out, bias=True):
        super().__init__()
        ref = torch.nn.Linear(n_in, n_out, bias=bias)
        self.weight = ref.weight
        if bias:
            self.bias = ref.bias
        else:
            self.register_parameter("bias", None)

    def forward(self, x, act=False):
        if self.bias is None:
            return _DenseFn.apply(x, self.weight, 1.0 if act else False)          # SiLU in the GEMM epilogue
        return _LinearBiasFn.apply(x, self.weight, self.bias, act)


class GaussianSmearing(torch.nn.Module):
    """escn/smearing.py:14-31."""

    def __init__(self, start, stop, num_gaussians, basis_width_scalar=1.0):
        super().__init__()
        self.num_output = num_gaussians
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (basis_width_scalar * (offset[1] - offset[0])).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, geom):
        out = torch.empty(geom.shape[0], self.num_output, device=geom.device, dtype=torch.float32)
        _lib.check(_lib.load().nq_es_smearing(_lib.ptr(geom), geom.shape[0], self.num_output, _lib.ptr(self.offset), self.coeff, _lib.ptr(out), _st()))
        return out


class EdgeBlock(torch.nn.Module):
    """escn.py:880-932."""

    def __init__(self, edge_channels, distance_expansion, max_num_elements):
        super().__init__()
        self.distance_expansion = distance_expansion
        self.fc1_dist = _Linear(distance_expansion.num_output, edge_channels)
        self.source_embedding = torch.nn.Embedding(max_num_elements, edge_channels)
        self.target_embedding = torch.nn.Embedding(max_num_elements, edge_channels)
        torch.nn.init.uniform_(self.source_embedding.weight.data, -0.001, 0.001)
        torch.nn.init.uniform_(self.target_embedding.weight.data, -0.001, 0.001)
        self.fc1_edge_attr = _Linear(edge_channels, edge_channels)

    def forward(self, x_dist, G):
        x = self.fc1_dist(x_dist)
        x = lin(x, _EmbeddingFn.apply(self.source_embedding.weight, G.z_src, [G.src_inverse + (G.N,), G.z_inverse]))
        x = lin(x, _EmbeddingFn.apply(self.target_embedding.weight, G.z_dst, [G.dst_inverse + (G.N,), G.z_inverse]))
        return self.fc1_edge_attr(_silu(x), act=True)


class SO2Conv(torch.nn.Module):
    """escn.py:807-877."""

    def __init__(self, m, sphere_channels, hidden_channels, edge_channels, lmax, mmax):
        super().__init__()
        assert mmax >= m
        self.m, self.hidden_channels = m, hidden_channels
        num_channels = (lmax - m + 1) * sphere_channels
        self.fc1_dist = _Linear(edge_channels, 2 * hidden_channels)
        self.fc1_r = _Linear(num_channels, hidden_channels, bias=False)
        self.fc2_r = _Linear(hidden_channels, num_channels, bias=False)
        self.fc1_i = _Linear(num_channels, hidden_channels, bias=False)
        self.fc2_i = _Linear(hidden_channels, num_channels, bias=False)

    def forward(self, x_re, x_im, x_edge):
        """x_re / x_im: [E, (lmax - m + 1) * C] (the +m and -m coefficients); returns the two parts of the result."""
        H = self.hidden_channels
        g = self.fc1_dist(x_edge, act=True)
        g_r, g_i = g[:, :H].contiguous(), g[:, H:].contiguous()
        if fused_pairs_available():      # one autograd node: the two output sums and the two input-gradient sums are formed in GEMM epilogues
            return _SO2GatedPairFn.apply(x_re, x_im, g_r, g_i, self.fc1_r.weight, self.fc1_i.weight, self.fc2_r.weight, self.fc2_i.weight)
        r0 = self.fc2_r(_MulFn.apply(self.fc1_r(x_re), g_r))
        r1 = self.fc2_r(_MulFn.apply(self.fc1_r(x_im), g_r))
        i0 = self.fc2_i(_MulFn.apply(self.fc1_i(x_re), g_i))
        i1 = self.fc2_i(_MulFn.apply(self.fc1_i(x_im), g_i))
        return lin(r0, i1, 1.0, -1.0), lin(r1, i0, 1.0, 1.0)                # x_r[:, 0] - x_i[:, 1], x_r[:, 1] + x_i[:, 0]


class SO2Block(torch.nn.Module):
    """escn.py:708-804; the embedding stays in m-primary order throughout (the rotation kernel writes it that way)."""

    def __init__(self, sphere_channels, hidden_channels, edge_channels, lmax, mmax):
        super().__init__()
        self.C = sphere_channels
        n0 = (lmax + 1) * sphere_channels
        self.fc1_dist0 = _Linear(edge_channels, hidden_channels)
        self.fc1_m0 = _Linear(n0, hidden_channels, bias=False)
        self.fc2_m0 = _Linear(hidden_channels, n0, bias=False)
        self.so2_conv = torch.nn.ModuleList([SO2Conv(m, sphere_channels, hidden_channels, edge_channels, lmax, mmax) for m in range(1, mmax + 1)])

    def forward(self, blocks, x_edge):
        """blocks: [m = 0 | +1 | -1 | +2 | -2 ...] each [E, n_m * C] (contiguous, straight from the rotation) -> the same list."""
        out = [self.fc2_m0(_MulFn.apply(self.fc1_m0(blocks[0]), self.fc1_dist0(x_edge, act=True)))]
        for m, conv in enumerate(self.so2_conv, start=1):
            out += list(conv(blocks[2 * m - 1], blocks[2 * m], x_edge))
        return out


class MessageBlock(torch.nn.Module):
    """escn.py:592-705."""

    def __init__(self, sphere_channels, hidden_channels, edge_channels, lmax, mmax, distance_expansion, max_num_elements):
        super().__init__()
        self.edge_block = EdgeBlock(edge_channels, distance_expansion, max_num_elements)
        self.so2_block_source = SO2Block(sphere_channels, hidden_channels, edge_channels, lmax, mmax)
        self.so2_block_target = SO2Block(sphere_channels, hidden_channels, edge_channels, lmax, mmax)

    def forward(self, x, G, K):
        """x: [N, n_full * C] -> messages summed per target atom, [N, n_full * C]."""
        Cc = K.C
        x_edge = self.edge_block(G.x_dist, G)
        o = K.order
        rows = K.block_rows
        xs = _RotateFn.apply(x, G, K, Cc, [(G.src, G.src_inverse)])                        # rotate into the edge frame: one tensor per m-block
        xt = _RotateFn.apply(x, G, K, Cc, [(G.dst, G.dst_inverse)])
        ys = [lin(a, b) for a, b in zip(self.so2_block_source(list(xs), x_edge), self.so2_block_target(list(xt), x_edge))]
        # point-wise SiLU on the (lmax, mmax) grid (so3.py:301-318), grid matrices with their columns in m-primary order
        ng = K.to_grid_red.shape[0]
        if s2_activation_fusable(K.to_grid_red, rows, Cc):
            ys = _S2ActBlocksFn.apply(K.to_grid_red, K.from_grid_red, rows, G.E, Cc, *ys)            # to_grid -> SiLU -> from_grid, the grid stays in registers
        else:
            grid = _BlocksInFn.apply(K.to_grid_red, 0, ng, o.n_red, Cc, False, 1, rows, G.E, *ys)
            ys = _BlocksOutFn.apply(_silu(grid), K.from_grid_red, 0, ng, o.n_red, Cc, True, 1, rows, None, None, G.E)
        return _RotateBackFn.apply(G, K, Cc, None, rows, *ys)                          # rotate back (wigner_inv = transpose) + _reduce_edge over the target's in-edges


class LayerBlock(torch.nn.Module):
    """escn.py:493-589."""

    def __init__(self, layer_idx, sphere_channels, hidden_channels, edge_channels, lmax, mmax, distance_expansion, max_num_elements):
        super().__init__()
        self.message_block = MessageBlock(sphere_channels, hidden_channels, edge_channels, lmax, mmax, distance_expansion, max_num_elements)
        self.fc1_sphere = _Linear(2 * sphere_channels, sphere_channels, bias=False)
        self.fc2_sphere = _Linear(sphere_channels, sphere_channels, bias=False)
        self.fc3_sphere = _Linear(sphere_channels, sphere_channels, bias=False)

    def forward(self, x, G, K):
        msg = self.message_block(x, G, K)
        Cc, nf, T, F = K.C, K.order.n_full, K.to_grid_full, K.from_grid_full
        ng = T.shape[0]
        gm = _RowFn.apply(msg, T, 0, ng, nf, Cc, False, None, None, G.N)                                   # [N, ng * C]
        gx = _RowFn.apply(x, T, 0, ng, nf, Cc, False, None, None, G.N)
        h = torch.cat([gx.view(-1, Cc), gm.view(-1, Cc)], dim=1)                                           # [N * ng, 2C]: cat along the channel (escn.py:580)
        h = self.fc3_sphere(self.fc2_sphere(self.fc1_sphere(h, act=True), act=True)).view(G.N, ng * Cc)
        out = _RowFn.apply(h, F, 0, ng, nf, Cc, True, None, None, G.N)                                     # from_grid
        return out


class EnergyBlock(torch.nn.Module):
    """escn.py:935-967."""

    def __init__(self, num_channels, num_sphere_samples):
        super().__init__()
        self.num_sphere_samples = num_sphere_samples
        self.fc1 = _Linear(num_channels, num_channels)
        self.fc2 = _Linear(num_channels, num_channels)
        self.fc3 = _Linear(num_channels, 1, bias=False)

    def forward(self, x_pt):
        return self.fc3(self.fc2(self.fc1(x_pt, act=True), act=True))                                     # [N * P, 1]; the mean over the points is taken by the caller


class ForceBlock(EnergyBlock):
    """escn.py:970-1003."""


class _Graph:
    pass


class _SphereSumFn(torch.autograd.Function):
    """out = a @ w with float64 accumulation (w: constant quadrature directions [P, 3])."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(w)
        return (a.double() @ w.double()).to(a.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return (g.double() @ w.double().t()).to(g.dtype), None


class eSCN(torch.nn.Module):
    """escn/escn.py:36-490 (constructor arguments :63-84)."""

    def __init__(self, num_targets: int, use_pbc: bool = True, regress_forces: bool = True, otf_graph: bool = False, max_neighbors: int = 40, cutoff: float = 8.0,
                 max_num_elements: int = 90, num_layers: int = 8, lmax_list: List[int] = [6], mmax_list: List[int] = [2], sphere_channels: int = 128,
                 hidden_channels: int = 256, edge_channels: int = 128, use_grid: bool = True, num_sphere_samples: int = 128, distance_function: str = "gaussian",
                 basis_width_scalar: float = 1.0, distance_resolution: float = 0.02, show_timing_info: bool = False) -> None:
        super().__init__()
        for ok, what in ((not use_pbc, "periodic boundary conditions"), (regress_forces, "regress_forces=False"), (len(lmax_list) == 1 and len(mmax_list) == 1,
                         "more than one resolution"), (use_grid, "use_grid=False"), (distance_function == "gaussian", f"distance_function={distance_function}"),
                         (lmax_list[0] <= 6, "lmax > 6"), (num_targets == 1, "num_targets != 1")):
            if not ok:
                raise NotImplementedError(f"eSCN on MI355X: {what} is not built (config/model/escn-oc.yaml is the supported configuration)")
        self.regress_forces, self.use_pbc, self.cutoff, self.otf_graph = regress_forces, use_pbc, cutoff, otf_graph
        self.max_num_elements, self.hidden_channels, self.num_layers = max_num_elements, hidden_channels, num_layers
        self.num_sphere_samples, self.sphere_channels, self.max_neighbors, self.edge_channels = num_sphere_samples, sphere_channels, max_neighbors, edge_channels
        self.lmax_list, self.mmax_list = list(lmax_list), list(mmax_list)
        self.sphere_channels_all = sphere_channels
        lmax, mmax = lmax_list[0], mmax_list[0]
        self.sphere_embedding = torch.nn.Embedding(max_num_elements, sphere_channels)
        self.num_gaussians = int(cutoff / distance_resolution)
        self.distance_expansion = GaussianSmearing(0.0, cutoff, self.num_gaussians, basis_width_scalar)
        self.layer_blocks = torch.nn.ModuleList([LayerBlock(i, sphere_channels, hidden_channels, edge_channels, lmax, mmax, self.distance_expansion, max_num_elements)
                                                 for i in range(num_layers)])
        self.energy_block = EnergyBlock(sphere_channels, num_sphere_samples)
        self.force_block = ForceBlock(sphere_channels, num_sphere_samples)
        pts = sphere_points(num_sphere_samples)
        self.sphere_points = torch.nn.Parameter(pts, requires_grad=False)
        sh = np.concatenate(_sh_e3nn(lmax, pts.double().numpy()), axis=1)                                   # o3.spherical_harmonics(0..lmax, points, normalize=False)
        self.sphharm_weights = torch.nn.ParameterList([torch.nn.Parameter(torch.tensor(sh, dtype=torch.float32), requires_grad=False)])
        # constants of the kernels (not parameters)
        o = CoefficientOrder(lmax, mmax)
        J = j_matrices(lmax)
        Tr, Fr = s2_grids(lmax, mmax)
        Tf, Ff = s2_grids(lmax, lmax)
        f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)                        # noqa: E731
        i32 = lambda a: torch.tensor(a, dtype=torch.int32)                                                # noqa: E731
        self._const = dict(J=f32(np.concatenate([j.reshape(-1) for j in J])), J_off=i32(np.cumsum([0] + [j.size for j in J])[:-1]),
                           red_l=i32([l for l, _ in o.m_primary]), red_row=i32([l + m for l, m in o.m_primary]),
                           to_grid_red=f32(Tr[:, o.red_m_primary]), from_grid_red=f32(Fr[:, o.red_m_primary]), to_grid_full=f32(Tf), from_grid_full=f32(Ff))
        self._order = o
        self._dev_const = None

    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def _constants(self, dev):
        if self._dev_const is None or self._dev_const.device != dev:
            K = _Graph()
            for k, v in self._const.items():
                setattr(K, k, v.to(dev))
            K.order, K.C, K.device = self._order, self.sphere_channels, dev
            o = self._order
            K.block_rows = [o.m_size[0]] + [o.m_size[m] for m in range(1, o.mmax + 1) for _ in (0, 1)]           # rows of every m-block: m = 0, +1, -1, +2, -2, ...
            K.red_l_host = [l for l, _ in o.m_primary]                  # degree of every kept Wigner row (host side of nq_es_rotate)
            self._dev_const = K
        return self._dev_const

    def prepare(self, data, edge_rot_mat=None):
        """Graph, frames, Wigner rows and index lists of a batch (all host reads of the step happen here).  ``data.prepared = net.prepare(data)`` makes
        ``forward(data)`` free of host synchronisation: a training step on that batch can be captured into a HIP graph (trainer.GraphedStep)."""
        G = self.build_graph(data, edge_rot_mat)
        G.geometry_key = _lib.geometry_key(data)        # checked by forward: a prepared batch is tied to its geometry
        return G

    def build_graph(self, data, edge_rot_mat=None):
        """radius graph + frames + Wigner rows (escn.py:313-325)."""
        lib = _lib.load()
        dev = data.pos.device
        pos = data.pos.detach().to(torch.float32).contiguous()
        N = pos.shape[0]
        batch = data.batch
        B = int(batch[-1].item()) + 1
        counts = torch.bincount(batch, minlength=B)
        mol_ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
        atom_mol = batch.to(torch.int32).contiguous()
        i32 = dict(device=dev, dtype=torch.int32)
        deg, ptr = torch.empty(N, **i32), torch.empty(N + 1, **i32)
        e_host = C.c_int32(0)
        _lib.check(lib.nq_es_graph_count(_lib.ptr(pos), _lib.ptr(mol_ptr), _lib.ptr(atom_mol), N, float(self.cutoff), int(self.max_neighbors), _lib.ptr(deg),
                                         _lib.ptr(ptr), C.byref(e_host), _st()))
        E = int(e_host.value)
        if E == 0:
            raise IndexError("batch has no edges within the cutoff")
        G = _Graph()
        G.N, G.B, G.E, G.ptr, G.mol_ptr, G.atom_mol = N, B, E, ptr, mol_ptr, atom_mol
        G.src, G.dst = torch.empty(E, **i32), torch.empty(E, **i32)
        G.geom = torch.empty(E, 4, device=dev, dtype=torch.float32)
        _lib.check(lib.nq_es_graph_fill(_lib.ptr(pos), _lib.ptr(mol_ptr), _lib.ptr(atom_mol), N, float(self.cutoff), int(self.max_neighbors), _lib.ptr(ptr),
                                        _lib.ptr(G.src), _lib.ptr(G.dst), _lib.ptr(G.geom), _st()))
        if edge_rot_mat is None:
            G.rot = torch.empty(E, 3, 3, device=dev, dtype=torch.float32)
            _lib.check(lib.nq_es_frames(_lib.ptr(G.geom), E, _lib.ptr(G.rot), _st()))
        else:
            G.rot = edge_rot_mat.to(device=dev, dtype=torch.float32).contiguous()
        K = self._constants(dev)
        o = K.order
        G.wigner = torch.empty(E, o.n_red * o.n_full, device=dev, dtype=torch.float32)
        scratch = torch.empty(3 * E, device=dev, dtype=torch.float64)          # Euler angles, evaluated in float64 (csrc/escn.hip: k_es_angles)
        _lib.check(lib.nq_es_wigner(_lib.ptr(G.rot), E, _lib.ptr(K.J), _lib.ptr(K.J_off), _lib.ptr(K.red_l), _lib.ptr(K.red_row), o.n_red, o.n_full, o.lmax,
                                    _lib.ptr(scratch), _lib.ptr(G.wigner), _st()))
        # inverse lists for the adjoints of the two gathers: the edges are sorted by target; by source through a stable sort of the source column
        G.dst_inverse = (None, ptr)
        G.src_inverse = _inverse_lists(G.src, N)
        G.z = data.z.to(torch.int32).contiguous()
        lo, hi = (int(v) for v in torch.stack([G.z.min().long(), G.z.max().long()]).tolist())      # one host read: an index outside the table would read out of bounds
        if lo < 0 or hi >= self.max_num_elements:
            raise IndexError(f"atomic numbers {lo}..{hi} outside 0..{self.max_num_elements - 1} (max_num_elements)")
        # atoms without any neighbour are legal (as in the reference): their message is zero
        G.z_src, G.z_dst = G.z[G.src.long()].contiguous(), G.z[G.dst.long()].contiguous()
        T = self.max_num_elements
        G.z_inverse = _inverse_lists(G.z, T) + (T,)
        G.x_dist = self.distance_expansion(G.geom)
        return G

    def forward(self, data, edge_rot_mat=None, return_layers: bool = False):
        _gemnet.weights_epoch_advance()   # bf16 weight copies are re-packed once per forward (parameters may have been updated in place)
        if not data.pos.is_cuda:
            raise RuntimeError("nabladft_amd.eSCN runs on MI355X only: tensors must be on a cuda (HIP) device")
        G = getattr(data, "prepared", None)            # build_graph(data) done ahead: the forward then issues no host synchronisation (HIP-graph capture)
        if G is None:
            G = self.build_graph(data, edge_rot_mat)
        elif G.N != int(data.pos.shape[0]):
            raise ValueError("data.prepared belongs to another batch")
        else:
            _lib.check_prepared(G, data)
        K = self._constants(data.pos.device)
        Cc, nf = self.sphere_channels, K.order.n_full
        emb = _EmbeddingFn.apply(self.sphere_embedding.weight, G.z, [G.z_inverse])                                   # [N, C] -> the l = 0 coefficient (escn.py:333-341)
        x = torch.cat([emb, emb.new_zeros(G.N, (nf - 1) * Cc)], dim=1)
        layers = []
        for i, blk in enumerate(self.layer_blocks):
            out = blk(x, G, K)
            layers.append(out)                                              # what a forward hook on the reference's LayerBlock sees
            x = out if i == 0 else lin(x, out)
        P = self.num_sphere_samples
        x_pt = _RowFn.apply(x, self.sphharm_weights[0], 0, P, nf, Cc, False, None, None, G.N).view(-1, Cc)      # einsum('abc,pb->apc') (escn.py:399-407)
        mean = torch.full((P, 1), 1.0 / P, device=x.device)
        node_energy = _MatmulFn.apply(self.energy_block(x_pt).view(G.N, P), mean)                     # [N, 1]
        energy = _SegSumFn.apply(node_energy, G.mol_ptr, G.atom_mol, G.B).squeeze(1) * 0.001          # escn.py:411-414
        # Force head (escn.py:437-457): a nearly constant scalar field times the unit vectors of the 128 sphere points -- the sum cancels to ~1e-3 of its terms,
        # so this ONE reduction is carried in float64 (a [N,128] x [128,3] product: no cost); the f32 sum's own rounding was the largest contribution to the
        # 4e-5 whole-batch vs single-molecule force difference of round 4
        forces = _SphereSumFn.apply(self.force_block(x_pt).view(G.N, P), (self.sphere_points / P).contiguous())
        if return_layers:
            return energy, forces, layers, G
        return energy, forces
