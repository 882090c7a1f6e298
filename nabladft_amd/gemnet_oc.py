"""GemNet-OC on the HIP kernels of csrc/gemnet_graph.hip + csrc/gemnet.hip and the fp32 MFMA GEMMs -- host-side mirror of the reference's
``nablaDFT.gemnet_oc.GemNetOC`` (gemnet_oc/gemnet_oc.py:36-1340; config/model/gemnet-oc.yaml): same constructor arguments, same parameter / buffer names
(``state_dict`` keys and order are equal, including the aliases of the shared modules), same outputs ``(energy [B], forces [N, 3])``.

What is different from the reference, by design (MI355X first):
  * every graph (a2a, main, a2ee2a, qint) is a CSR by target atom built on the device; the triplet / mixed-triplet / quadruplet index lists of
    interaction_indices.py are never materialised -- the interaction kernels walk "the other in-edges of the same atom" and evaluate the angular bases in
    registers, so the zero-padded ``[E, Kmax, C]`` tiles of layers/efficient.py do not exist either;
  * the main graph is kept in CSR order (the reference: per molecule [source < target][flips]); edge order is not observable in the outputs.
    ``GemNetGraphs.to_reference()`` gives the reference's arrays for the parity tests;
  * all sums run in a fixed order (bitwise reproducible).
Supported configuration: the one config/model/gemnet-oc.yaml uses and its size / cutoff / cap variations -- gaussian radial basis with polynomial
envelope, ``spherical_harmonics`` circular basis, ``legendre_outer`` spherical basis, direct forces, no periodic boundary conditions.  Anything else
raises NotImplementedError.  There is no CPU path: tensors must live on the MI355X and libnablaq.so must load.
"""
import ctypes as C
import math
from types import SimpleNamespace
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _lib
from .painn import build_neighbor_list
from .qhnet import _MatmulFn, _f32

_INV_SQRT2 = 1.0 / math.sqrt(2.0)


def _st():
    return _lib.stream_ptr()


def _set(n, ptr, src, dst, geom):
    s = _lib.GnSet()
    s.n, s.reserved = int(n), 0
    s.ptr, s.src, s.dst, s.geom = ptr.data_ptr(), src.data_ptr(), dst.data_ptr(), geom.data_ptr()
    return s


class GemNetGraphs:
    """The graphs of gemnet_oc.py:892-958 for one batch.  ``main / aea / qint / a2a``: ctypes nq_gn_set views (CSR by target atom); ``t``: the tensors."""

    def to_reference(self):
        """Host copies in the reference's layout (parity tests / debugging): edge_index [2, E] (row 0 source, row 1 target), distance, vector per
        graph; for the main graph in the reference's order (per molecule: the source < target edges by (target, source), then their flips), id_swap, and
        ``main_ref_id`` = reference edge id of every CSR slot."""
        t = {k: v.detach().cpu().numpy() for k, v in self.t.items() if torch.is_tensor(v)}
        out = {}
        for name, src, dst, geom in (("a2a", "col", "dst", "geom"), ("a2ee2a", "a_src", "a_dst", "a_geom"), ("qint", "q_src", "q_dst", "q_geom")):
            g = t[geom]
            sign = -1.0 if name == "a2a" else 1.0          # the a2a set keeps graph.hip's (target -> source) unit vectors
            out[name] = dict(edge_index=np.stack([t[src], t[dst]]).astype(np.int64), distance=g[:, 3].copy(), vector=sign * g[:, :3])
        ptr, low, src, dst, rev = t["ptr_m"], t["lowptr_m"], t["m_src"], t["m_dst"], t["m_rev"]
        mol = t["mol_ptr"]
        E = src.shape[0]
        ref_id = np.zeros(E, dtype=np.int64)
        for b in range(mol.shape[0] - 1):
            a0, a1 = mol[b], mol[b + 1]
            base, half = 2 * low[a0], low[a1] - low[a0]
            for i in range(a0, a1):
                nlow = low[i + 1] - low[i]
                for k in range(nlow):                      # the lower entries come first in a source-ascending row
                    ref_id[ptr[i] + k] = base + (low[i] - low[a0]) + k
        lower = src < dst
        half = np.array([low[mol[b + 1]] - low[mol[b]] for b in range(mol.shape[0] - 1)], dtype=np.int64)
        ref_id[~lower] = ref_id[rev[~lower]] + half[t["atom_mol"][dst[~lower]]]
        order = np.argsort(ref_id)
        g = t["m_geom"]
        out["main"] = dict(edge_index=np.stack([src, dst]).astype(np.int64)[:, order], distance=g[order, 3], vector=g[order, :3])
        out["id_swap"] = ref_id[rev][order]
        out["main_ref_id"] = ref_id
        return out


def build_graphs(pos, batch, z, cutoff, cutoff_qint, cutoff_aeaint, cutoff_aint, max_neighbors, max_neighbors_qint, max_neighbors_aeaint,
                 max_neighbors_aint, ptr=None) -> GemNetGraphs:
    lib = _lib.load()
    nl = build_neighbor_list(pos, batch, z, cutoff_aint, 100000, ptr)          # a2a: radius_graph(r=cutoff_aint) (gemnet_oc.py:1318-1322), the cap checked below
    if nl.E == 0:
        raise IndexError("batch has no edges within the cutoff")
    t = nl.t
    dev = pos.device
    N, E = nl.N, nl.E
    mp = t["mol_ptr"].long()
    per_mol = t["row_ptr"].long()[mp[1:]] - t["row_ptr"].long()[mp[:-1]]
    max_deg, min_mol = (int(v) for v in torch.stack([t["deg"].max().long(), per_mol.min()]).tolist())      # one host read for both checks
    if min_mol == 0:
        raise ValueError("An image has no neighbors")                # gemnet_oc.py:821-825 (a molecule without any pair inside the cutoff)
    if max_deg > max_neighbors_aint:
        raise NotImplementedError(f"an atom has {max_deg} neighbours within cutoff_aint but max_neighbors_aint = {max_neighbors_aint}: the reference's "
                                  "radius_graph truncation by source order is not implemented")
    i32 = dict(device=dev, dtype=torch.int32)
    g = _lib.GnGraphs()
    g.N, g.E, g.k_main, g.k_aea, g.k_qint, g.reserved = N, E, int(max_neighbors), int(max_neighbors_aeaint), int(max_neighbors_qint), 0
    g.cutoff_main, g.cutoff_aea, g.cutoff_qint = float(cutoff), float(cutoff_aeaint), float(cutoff_qint)
    T = {}

    def new(name, n, dtype=torch.int32, width=None):
        T[name] = torch.empty((n,) if width is None else (n, width), device=dev, dtype=dtype)
        setattr(g, name, T[name].data_ptr())
        return T[name]

    for k in ("row_ptr", "col", "rev", "dst", "pos", "geom"):
        setattr(g, k, t[k].data_ptr())
    new("flags", E, torch.uint8)
    for k in ("degm", "lowm", "cnt_a", "cnt_q", "tin_atom"):
        new(k, N)
    for k in ("ptr_m", "lowptr_m", "ptr_a", "ptr_q", "tin_aptr"):
        new(k, N + 1)
    for k in ("mpos", "apos", "qpos", "a_of_rev", "q_of_rev"):
        new(k, E)
    totals = (C.c_int32 * 4)()
    _lib.check(lib.nq_gn_graph_count(C.byref(g), max_deg, totals, _st()))
    Em, Ea, Eq, Tin = (int(v) for v in totals)
    if min(Em, Ea, Eq) == 0:
        raise ValueError("An image has no neighbors")               # gemnet_oc.py:821-825
    for k in ("m_src", "m_dst", "m_rev", "m_slot"):
        new(k, Em)
    new("m_geom", Em, torch.float32, 4)
    for k in ("a_src", "a_dst"):
        new(k, Ea)
    new("a_geom", Ea, torch.float32, 4)
    for k in ("q_src", "q_dst"):
        new(k, Eq)
    new("q_geom", Eq, torch.float32, 4)
    new("tin_ptr", Eq + 1)
    new("tin_main", max(Tin, 1))
    _lib.check(lib.nq_gn_graph_fill(C.byref(g), _st()))
    G = GemNetGraphs()
    G.N, G.B, G.Em, G.Ea, G.Eq, G.Ea2a, G.Tin = N, nl.B, Em, Ea, Eq, E, Tin
    G.KQ = int(max_neighbors_qint) if max_neighbors_qint <= 64 else int(T["cnt_q"].max().item())
    T.update(row_ptr=t["row_ptr"], col=t["col"], dst=t["dst"], rev=t["rev"], geom=t["geom"], mol_ptr=t["mol_ptr"], z=t["z"], atom_mol=t["atom_mol"])
    G.t = T
    G.main = _set(Em, T["ptr_m"], T["m_src"], T["m_dst"], T["m_geom"])
    G.aea = _set(Ea, T["ptr_a"], T["a_src"], T["a_dst"], T["a_geom"])
    G.qint = _set(Eq, T["ptr_q"], T["q_src"], T["q_dst"], T["q_geom"])
    G.a2a = _set(E, T["row_ptr"], T["col"], T["dst"], T["geom"])
    G._c = g
    return G


# ============================================================================================================================================
# autograd wrappers of the C entry points (no CPU path)
# ============================================================================================================================================
GEMM_BYTES = [None]      # bench hook: compulsory HBM bytes of the same products (operands read once, result written once; weights 2 B in the bf16 mode)
GEMM_FLOPS = [None]      # bench hook: set GEMM_FLOPS[0] = 0.0 to accumulate 2*M*N*K of every dense product of the forward pass (backward = 2x that)


def _new(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


_SSILU = 1.0 / 0.6       # ScaledSiLU (layers/base_layers.py:61-71)
_PRECISION = ["f32"]
_PACKED = {}
_EPOCH = [0]


def weights_epoch_advance() -> None:
    """Marks every packed bf16 weight copy stale.  Called at the start of each model forward: between two forwards the optimizer, an EMA swap or a
    load_state_dict may have rewritten the parameters IN PLACE -- FlatParameters points `p.data` at slices of one flat buffer and the fused AdamW /
    `ema.copy_to` write through raw pointers, none of which moves `p._version`, so a version-keyed cache would keep serving the initial weights.
    The backward of the same step re-uses the copies packed in its forward (same epoch)."""
    _EPOCH[0] += 1


def set_gemm_precision(mode: str) -> None:
    """"f32" (default: exact-f32 MFMA, the parity-tested path) or "bf16": the forward and input-gradient products of the Dense layers run on bf16 MFMA with
    fp32 accumulation (operands rounded to bf16, weights packed once per parameter version); weight gradients, all sums over edges / triplets /
    quadruplets and the optimizer stay fp32 -- the mode BASELINE.json names for config/model/gemnet-oc.yaml."""
    if mode not in ("f32", "bf16"):
        raise ValueError(mode)
    _PRECISION[0] = mode
    _PACKED.clear()
    weights_epoch_advance()


def _packed(W):
    """(Wb [N, K], WbT [K, N]) bf16 copies of a weight, packed once per weights epoch (= once per model forward, see weights_epoch_advance)."""
    key = (W.data_ptr(), tuple(W.shape))
    ent = _PACKED.get(key)
    if ent is None or ent[0] != _EPOCH[0]:
        N, K = W.shape
        Wb = torch.empty(N, K, device=W.device, dtype=torch.bfloat16)
        WbT = torch.empty(K, N, device=W.device, dtype=torch.bfloat16)
        _lib.check(_lib.load().nq_bf16_pack(_lib.ptr(W), N, K, _lib.ptr(Wb), _lib.ptr(WbT), _st()))
        ent = (_EPOCH[0], Wb, WbT)
        _PACKED[key] = ent
    return ent[1], ent[2]


def _use_bf16(M, contract):
    return _PRECISION[0] == "bf16" and contract % 32 == 0 and M >= 256


def _gemm_act(x, W, resid, alpha, beta):
    """(pre, out): pre = x W^T, out = alpha * resid + beta * silu(pre) in the GEMM epilogue (one pass over the output tile)."""
    M, K = x.shape
    N = W.shape[0]
    pre, out = _new(M, N, like=x), _new(M, N, like=x)
    if M > 0 and _use_bf16(M, K):
        _lib.check(_lib.load().nq_linear_forward_bf16(_lib.ptr(x), _lib.ptr(_packed(W)[0]), _lib.ptr(pre), _lib.ptr(out), None if resid is None else _lib.ptr(resid),
                                                      float(alpha), float(beta), M, N, K, _st()))
    elif M > 0:
        _lib.check(_lib.load().nq_linear_forward_act(_lib.ptr(x), _lib.ptr(W), _lib.ptr(pre), _lib.ptr(out), None if resid is None else _lib.ptr(resid),
                                                     float(alpha), float(beta), M, N, K, _st()))
    if GEMM_FLOPS[0] is not None:
        GEMM_FLOPS[0] += 2.0 * M * N * K
    if GEMM_BYTES[0] is not None:
        GEMM_BYTES[0] += 4.0 * M * K + 8.0 * M * N + (2.0 if _use_bf16(M, K) else 4.0) * N * K      # x read, pre and out written, W read
    return pre, out


def _ssilu_bwd(pre, g, scale):
    out = torch.empty_like(g)
    _lib.check(_lib.load().nq_gn_ssilu_backward(_lib.ptr(pre), _lib.ptr(g), float(scale), g.numel(), _lib.ptr(out), _st()))
    return out


def _dgrad(g, W, out=None, accumulate=False):
    M, N = g.shape
    K = W.shape[1]
    if out is None:
        out = _new(M, K, like=g)
    if M > 0 and _use_bf16(M, N):
        _lib.check(_lib.load().nq_linear_input_grad_bf16(_lib.ptr(g), _lib.ptr(_packed(W)[1]), _lib.ptr(out), M, N, K, int(accumulate), _st()))
    elif M > 0:
        _lib.check(_lib.load().nq_linear_input_grad(_lib.ptr(g), _lib.ptr(W), _lib.ptr(out), M, N, K, int(accumulate), _st()))
    return out


def _dgrad_epi(g, W, aux, alpha, beta, mode):
    """mode 1: beta * (g W) * silu'(aux); mode 2: alpha * aux + g W -- in the GEMM epilogue."""
    M, N = g.shape
    K = W.shape[1]
    if M > 0 and _use_bf16(M, N):
        # measured: on the short bf16 kernel the 64 scattered aux loads per lane of the fused epilogue cost more (73 us vs 37 us per 20 k x 512 x 512 product)
        # than the separate streaming pass they replace -- keep the passes separate there
        raw = _dgrad(g, W)
        return _ssilu_bwd(aux, raw, beta / _SSILU) if mode == 1 else _lin_raw(aux, raw, alpha, 1.0)
    out = _new(M, K, like=g)
    if M > 0:
        _lib.check(_lib.load().nq_linear_input_grad_epi(_lib.ptr(g), _lib.ptr(W), _lib.ptr(out), M, N, K, _lib.ptr(aux), float(alpha), float(beta), mode, _st()))
    return out


def _wgrad(g, x):
    lib = _lib.load()
    M, N = g.shape
    K = x.shape[1]
    gW = _new(N, K, like=g)
    if _PRECISION[0] == "bf16" and M >= 2048:
        scr = torch.empty(int(lib.nq_weight_grad_bf16_scratch_bytes(M, N, K)), device=g.device, dtype=torch.uint8)
        _lib.check(lib.nq_linear_weight_grad_bf16(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), _st()))
        return gW
    scr = _new(int(lib.nq_weight_grad_scratch_floats(M, N, K)) + 64, like=g)
    _lib.check(lib.nq_linear_weight_grad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), _st()))
    return gW


class _DenseFn(torch.autograd.Function):
    """Dense without bias (layers/base_layers.py:11-58): y = act(x W^T), act = ScaledSiLU (silu / 0.6, fused into the GEMM epilogue) or identity."""

    @staticmethod
    def forward(ctx, x, W, act):
        """act: False / 0 = identity, True = ScaledSiLU (silu / 0.6), a float s = s * silu (1.0: plain SiLU, escn.py)."""
        x, W = _f32(x), _f32(W)
        ctx.act = _SSILU if act is True else float(act)
        if ctx.act:
            pre, y = _gemm_act(x, W, None, 0.0, ctx.act)
            ctx.save_for_backward(x, W, pre)
            return y
        M, K = x.shape
        N = W.shape[0]
        pre = _new(M, N, like=x)
        if M > 0 and _use_bf16(M, K):
            _lib.check(_lib.load().nq_linear_forward_bf16(_lib.ptr(x), _lib.ptr(_packed(W)[0]), _lib.ptr(pre), None, None, 0.0, 0.0, M, N, K, _st()))
        elif M > 0:
            _lib.check(_lib.load().nq_linear_forward(_lib.ptr(x), _lib.ptr(W), None, _lib.ptr(pre), None, M, N, K, _st()))
        if GEMM_FLOPS[0] is not None:
            GEMM_FLOPS[0] += 2.0 * M * N * K
        if GEMM_BYTES[0] is not None:
            GEMM_BYTES[0] += 4.0 * M * K + 4.0 * M * N + (2.0 if _use_bf16(M, K) else 4.0) * N * K
        ctx.save_for_backward(x, W)
        return pre

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        if ctx.act:
            x, W, pre = ctx.saved_tensors
            g = _ssilu_bwd(pre, g, ctx.act / _SSILU)
        else:
            x, W = ctx.saved_tensors
        gx = _dgrad(g, W) if ctx.needs_input_grad[0] else None
        gW = _wgrad(g, x) if ctx.needs_input_grad[1] else None
        return gx, gW, None


def _fwd_f32(x, W, aux=None, alpha=0.0):
    """x W^T, or alpha * aux + x W^T in the epilogue of the same launch (f32-accurate engines only)."""
    M, K = x.shape
    N = W.shape[0]
    out = _new(M, N, like=x)
    if M > 0 and aux is None:
        _lib.check(_lib.load().nq_linear_forward(_lib.ptr(x), _lib.ptr(W), None, _lib.ptr(out), None, M, N, K, _st()))
    elif M > 0:
        _lib.check(_lib.load().nq_linear_forward_res(_lib.ptr(x), _lib.ptr(W), _lib.ptr(aux), float(alpha), _lib.ptr(out), M, N, K, _st()))
    if GEMM_FLOPS[0] is not None:
        GEMM_FLOPS[0] += 2.0 * M * N * K
    if GEMM_BYTES[0] is not None:
        GEMM_BYTES[0] += 4.0 * M * K + (4.0 if aux is None else 8.0) * M * N + 4.0 * N * K
    return out


def fused_pairs_available():
    """The two-term products below exist for the f32-accurate engines; the bf16 mode keeps the composed form (its operands are packed per launch)."""
    return _PRECISION[0] != "bf16"


class _SO2PairFn(torch.autograd.Function):
    """The +-m pair of an SO(2) convolution whose two inputs share the weights (equiformer_v2 so2_ops.py:53-61):
        out_p = x_p W_r^T - x_m W_i^T,   out_m = x_m W_r^T + x_p W_i^T
    as four launches (the second product of each sum takes the first one in its epilogue) instead of four products and two linear combinations; the
    adjoint likewise sums its two contributions per input inside the input-gradient products (no autograd accumulation passes)."""

    @staticmethod
    def forward(ctx, xp, xm, Wr, Wi):
        xp, xm, Wr, Wi = _f32(xp), _f32(xm), _f32(Wr), _f32(Wi)
        out_p = _fwd_f32(xp, Wr, _fwd_f32(xm, Wi), -1.0)
        out_m = _fwd_f32(xm, Wr, _fwd_f32(xp, Wi), 1.0)
        ctx.save_for_backward(xp, xm, Wr, Wi)
        return out_p, out_m

    @staticmethod
    def backward(ctx, gp, gm):
        xp, xm, Wr, Wi = ctx.saved_tensors
        gp, gm = _f32(gp), _f32(gm)
        need = ctx.needs_input_grad
        dxp = _dgrad_epi(gp, Wr, _dgrad(gm, Wi), 1.0, 0.0, 2) if need[0] else None          # g_p W_r + g_m W_i
        dxm = _dgrad_epi(gm, Wr, _dgrad(gp, Wi), -1.0, 0.0, 2) if need[1] else None         # g_m W_r - g_p W_i
        dWr = _wgrad(gp, xp).add_(_wgrad(gm, xm)) if need[2] else None
        dWi = _wgrad(gm, xp).sub_(_wgrad(gp, xm)) if need[3] else None
        return dxp, dxm, dWr, dWi


class _SO2GatedPairFn(torch.autograd.Function):
    """eSCN's SO(2) convolution of one m > 0 (escn.py:858-877) with everything between its inputs and outputs in one autograd node:
        a_r{0,1} = (x_{re,im} W1r^T) * g_r,  a_i{0,1} = (x_{re,im} W1i^T) * g_i,
        out_p = a_r0 W2r^T - a_i1 W2i^T,     out_m = a_r1 W2r^T + a_i0 W2i^T.
    Both output sums and, in the adjoint, both input-gradient sums are formed in GEMM epilogues (alpha * aux + product); the composed form spent two
    linear combinations forward and one scaled copy plus two gradient accumulations of [E, n] tensors backward on them."""

    @staticmethod
    def forward(ctx, x_re, x_im, g_r, g_i, W1r, W1i, W2r, W2i):
        x_re, x_im, g_r, g_i, W1r, W1i, W2r, W2i = (_f32(t) for t in (x_re, x_im, g_r, g_i, W1r, W1i, W2r, W2i))
        p = [_fwd_f32(x_re, W1r), _fwd_f32(x_im, W1r), _fwd_f32(x_re, W1i), _fwd_f32(x_im, W1i)]             # r0, r1, i0, i1
        a = [_mul_raw(p[0], g_r), _mul_raw(p[1], g_r), _mul_raw(p[2], g_i), _mul_raw(p[3], g_i)]
        out_p = _fwd_f32(a[0], W2r, _fwd_f32(a[3], W2i), -1.0)
        out_m = _fwd_f32(a[1], W2r, _fwd_f32(a[2], W2i), 1.0)
        ctx.save_for_backward(x_re, x_im, g_r, g_i, W1r, W1i, W2r, W2i, *p, *a)
        return out_p, out_m

    @staticmethod
    def backward(ctx, gp, gm):
        x_re, x_im, g_r, g_i, W1r, W1i, W2r, W2i, p_r0, p_r1, p_i0, p_i1, a_r0, a_r1, a_i0, a_i1 = ctx.saved_tensors
        gp, gm = _f32(gp), _f32(gm)
        need = ctx.needs_input_grad
        da_r0, da_r1, da_i0, da_i1n = _dgrad(gp, W2r), _dgrad(gm, W2r), _dgrad(gm, W2i), _dgrad(gp, W2i)      # da_i1 = -da_i1n
        dW2r = _wgrad(gp, a_r0).add_(_wgrad(gm, a_r1)) if need[6] else None
        dW2i = _wgrad(gm, a_i0).sub_(_wgrad(gp, a_i1)) if need[7] else None
        dp_r0, dp_r1, dp_i0, dp_i1n = _mul_raw(da_r0, g_r), _mul_raw(da_r1, g_r), _mul_raw(da_i0, g_i), _mul_raw(da_i1n, g_i)
        dg_r = _lin_raw(_mul_raw(da_r0, p_r0), _mul_raw(da_r1, p_r1), 1.0, 1.0) if need[2] else None
        dg_i = _lin_raw(_mul_raw(da_i0, p_i0), _mul_raw(da_i1n, p_i1), 1.0, -1.0) if need[3] else None
        dx_re = _dgrad_epi(dp_r0, W1r, _dgrad(dp_i0, W1i), 1.0, 0.0, 2) if need[0] else None                    # dp_r0 W1r + dp_i0 W1i
        dx_im = _dgrad_epi(dp_r1, W1r, _dgrad(dp_i1n, W1i), -1.0, 0.0, 2) if need[1] else None                  # dp_r1 W1r - dp_i1n W1i
        dW1r = _wgrad(dp_r0, x_re).add_(_wgrad(dp_r1, x_im)) if need[4] else None
        dW1i = _wgrad(dp_i0, x_re).sub_(_wgrad(dp_i1n, x_im)) if need[5] else None
        return dx_re, dx_im, dg_r, dg_i, dW1r, dW1i, dW2r, dW2i


class _ResidualFn(torch.autograd.Function):
    """ResidualLayer (layers/base_layers.py:74-97) with two activated Dense layers: out = (x + ssilu(ssilu(x W1^T) W2^T)) / sqrt(2) -- two GEMM launches
    forward (activation and residual in the epilogues), two elementwise + four GEMM launches backward."""

    @staticmethod
    def forward(ctx, x, W1, W2):
        x, W1, W2 = _f32(x), _f32(W1), _f32(W2)
        pre1, a1 = _gemm_act(x, W1, None, 0.0, _SSILU)
        pre2, out = _gemm_act(a1, W2, x, _INV_SQRT2, _INV_SQRT2 * _SSILU)
        ctx.save_for_backward(x, W1, W2, pre1, a1, pre2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W1, W2, pre1, a1, pre2 = ctx.saved_tensors
        g = _f32(g)
        gp2 = _ssilu_bwd(pre2, g, _INV_SQRT2)
        gW2 = _wgrad(gp2, a1) if ctx.needs_input_grad[2] else None
        gp1 = _dgrad_epi(gp2, W2, pre1, 0.0, _SSILU, 1)                      # (gp2 W2) * d ssilu(pre1)
        gW1 = _wgrad(gp1, x) if ctx.needs_input_grad[1] else None
        gx = _dgrad_epi(gp1, W1, g, _INV_SQRT2, 0.0, 2) if ctx.needs_input_grad[0] else None      # g / sqrt(2) + gp1 W1
        return gx, gW1, gW2


def _mul_raw(a, b):
    out = torch.empty_like(a)
    _lib.check(_lib.load().nq_gn_mul(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(out), _st()))
    return out


def _lin_raw(a, b, alpha, beta):
    out = torch.empty_like(a)
    _lib.check(_lib.load().nq_gn_lincomb(_lib.ptr(a), _lib.ptr(b), float(alpha), float(beta), a.numel(), _lib.ptr(out), _st()))
    return out


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        ctx.save_for_backward(a, b)
        return _mul_raw(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32(g)
        return (_mul_raw(g, b) if ctx.needs_input_grad[0] else None), (_mul_raw(g, a) if ctx.needs_input_grad[1] else None)


class _LinFn(torch.autograd.Function):
    """alpha a + beta b (b optional)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        ctx.ab = (alpha, beta, b is not None)
        return _lin_raw(_f32(a), None if b is None else _f32(b), alpha, beta)

    @staticmethod
    def backward(ctx, g):
        alpha, beta, has_b = ctx.ab
        g = _f32(g)
        # a unit coefficient passes the incoming gradient on unchanged (no copy: nothing downstream writes into its incoming gradient, and autograd only accumulates
        # in place into buffers it owns alone) -- the residual sums and the source + target sums of the SO(2) blocks are all 1 * a + 1 * b
        ga = (g if alpha == 1.0 else _lin_raw(g, None, alpha, 0.0)) if ctx.needs_input_grad[0] else None
        gb = (g if beta == 1.0 else _lin_raw(g, None, beta, 0.0)) if has_b and ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def lin(a, b=None, alpha=1.0, beta=1.0):
    return _LinFn.apply(a, b, alpha, beta)


def _gather_raw(x, idx, y, P):
    Cc = x.shape[1]
    out = _new(P, Cc, like=x)
    _lib.check(_lib.load().nq_gn_gather(_lib.ptr(x), _lib.ptr(idx), None if y is None else _lib.ptr(y), P, Cc, _lib.ptr(out), _st()))
    return out


class _GatherMulFn(torch.autograd.Function):
    """out[p] = x[idx[p]] * y[p] (y optional).  ``scatter(rows)`` is the adjoint of the gather (a fixed-order segment sum over the inverse lists the
    graph stage built: never an atomic scatter)."""

    @staticmethod
    def forward(ctx, x, y, idx, scatter):
        x = _f32(x)
        y = None if y is None else _f32(y)
        ctx.idx, ctx.scatter = idx, scatter
        ctx.save_for_backward(x, y if y is not None else x.new_zeros(0))
        ctx.has_y = y is not None
        return _gather_raw(x, idx, y, idx.numel())

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _f32(g)
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = ctx.scatter(_mul_raw(g, y) if ctx.has_y else g)
        if ctx.has_y and ctx.needs_input_grad[1]:
            gy = _gather_raw(x, ctx.idx, g, ctx.idx.numel())
        return gx, gy, None, None


def _segsum_raw(rows, order, ptr, n, Cc):
    out = _new(n, Cc, like=rows)
    _lib.check(_lib.load().nq_gn_segment_sum(_lib.ptr(rows), None, None if order is None else _lib.ptr(order), _lib.ptr(ptr), n, Cc, _lib.ptr(out), _st()))
    return out


class _SegSumFn(torch.autograd.Function):
    """out[n] = sum of the CONTIGUOUS rows [ptr[n], ptr[n+1]); ``owner[row]`` = n for the adjoint."""

    @staticmethod
    def forward(ctx, rows, ptr, owner, n):
        rows = _f32(rows)
        ctx.owner = owner
        return _segsum_raw(rows, None, ptr, n, rows.shape[1])

    @staticmethod
    def backward(ctx, g):
        return _gather_raw(_f32(g), ctx.owner, None, ctx.owner.numel()), None, None, None


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, G, out_set, in_set, NS, scale):
        x = _f32(x)
        Cc = x.shape[1]
        S = _new(out_set.n, NS * Cc, like=x)
        _lib.check(_lib.load().nq_gn_triplet_forward(C.byref(out_set), C.byref(in_set), _lib.ptr(x), Cc, NS, scale, _lib.ptr(S), _st()))
        ctx.meta = (G, out_set, in_set, NS, scale, Cc)
        return S

    @staticmethod
    def backward(ctx, g):
        G, out_set, in_set, NS, scale, Cc = ctx.meta
        g = _f32(g)
        dx = _new(in_set.n, Cc, like=g)
        _lib.check(_lib.load().nq_gn_triplet_backward(C.byref(out_set), C.byref(in_set), _lib.ptr(g), Cc, NS, scale, _lib.ptr(dx), _st()))
        return dx, None, None, None, None, None


class _QuadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, G, NS, scale):
        x = _f32(x)
        Cc = x.shape[1]
        S = _new(G.Em, NS * NS * Cc, like=x)
        _lib.check(_lib.load().nq_gn_quad_forward(C.byref(G.main), C.byref(G.qint), _lib.ptr(G.t["tin_ptr"]), G.N, _lib.ptr(x), Cc, NS, scale, _lib.ptr(S),
                                                  _st()))
        ctx.meta = (G, NS, scale, Cc)
        return S

    @staticmethod
    def backward(ctx, g):
        G, NS, scale, Cc = ctx.meta
        g = _f32(g)
        dx = _new(max(G.Tin, 1), Cc, like=g)[:G.Tin]
        scr = _new(G.Em * G.KQ * NS * Cc + 64, like=g)
        _lib.check(_lib.load().nq_gn_quad_backward(C.byref(G.main), C.byref(G.qint), _lib.ptr(G.t["tin_ptr"]), G.N, G.Tin, _lib.ptr(g), Cc, NS, G.KQ, scale,
                                                   _lib.ptr(scr), _lib.ptr(dx), _st()))
        return dx, None, None, None


class _CirFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rw, G, I, NS, scale):
        rw = _f32(rw)
        cir = _new(max(G.Tin, 1), I, like=rw)[:G.Tin]
        _lib.check(_lib.load().nq_gn_cir_forward(C.byref(G.main), C.byref(G.qint), _lib.ptr(G.t["tin_ptr"]), G.Tin, _lib.ptr(rw), I, NS, scale, _lib.ptr(cir),
                                                 _st()))
        ctx.meta = (G, I, NS, scale)
        return cir

    @staticmethod
    def backward(ctx, g):
        G, I, NS, scale = ctx.meta
        g = _f32(g)
        drw = _new(G.Eq, I * NS, like=g)
        _lib.check(_lib.load().nq_gn_cir_backward(C.byref(G.main), C.byref(G.qint), _lib.ptr(G.t["tin_ptr"]), _lib.ptr(g), I, NS, scale, _lib.ptr(drw), _st()))
        return drw, None, None, None, None


class _RowMMFn(torch.autograd.Function):
    """out[o, i * C + c] = sum_s R[o, i * NSS + s] S[o, s * C + c]."""

    @staticmethod
    def forward(ctx, R, S, I, NSS, Cc):
        R, S = _f32(R), _f32(S)
        n = R.shape[0]
        out = _new(n, I * Cc, like=R)
        _lib.check(_lib.load().nq_gn_rowmm_forward(_lib.ptr(R), _lib.ptr(S), n, I, NSS, Cc, _lib.ptr(out), _st()))
        ctx.save_for_backward(R, S)
        ctx.meta = (I, NSS, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        R, S = ctx.saved_tensors
        I, NSS, Cc = ctx.meta
        g = _f32(g)
        dR = torch.empty_like(R) if ctx.needs_input_grad[0] else None
        dS = torch.empty_like(S) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().nq_gn_rowmm_backward(_lib.ptr(R), _lib.ptr(S), _lib.ptr(g), R.shape[0], I, NSS, Cc, _lib.ptr(dR), _lib.ptr(dS), _st()))
        return dR, dS, None, None, None


class _PairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rw, x, G):
        rw, x = _f32(rw), _f32(x)
        Rr, Cc = rw.shape[1], x.shape[1]
        out = _new(G.N, Rr * Cc, like=x)
        _lib.check(_lib.load().nq_gn_pair_forward(C.byref(G.a2a), _lib.ptr(rw), _lib.ptr(x), G.N, Rr, Cc, _lib.ptr(out), _st()))
        ctx.save_for_backward(rw, x)
        ctx.G = G
        return out

    @staticmethod
    def backward(ctx, g):
        rw, x = ctx.saved_tensors
        G = ctx.G
        g = _f32(g)
        drw = torch.empty_like(rw) if ctx.needs_input_grad[0] else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().nq_gn_pair_backward(C.byref(G.a2a), _lib.ptr(G.t["rev"]), _lib.ptr(rw), _lib.ptr(x), _lib.ptr(g), G.N, rw.shape[1], x.shape[1],
                                                   _lib.ptr(drw), _lib.ptr(dx), _st()))
        return drw, dx, None


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, m, G):
        h, m = _f32(h), _f32(m)
        A, Em = h.shape[1], m.shape[1]
        cat = _new(G.Em, 2 * A + Em, like=h)
        _lib.check(_lib.load().nq_gn_cat_forward(C.byref(G.main), _lib.ptr(h), _lib.ptr(m), A, Em, _lib.ptr(cat), _st()))
        ctx.meta = (G, A, Em)
        return cat

    @staticmethod
    def backward(ctx, g):
        G, A, Em = ctx.meta
        g = _f32(g)
        dh = _new(G.N, A, like=g)
        _lib.check(_lib.load().nq_gn_cat_backward_h(C.byref(G.main), _lib.ptr(G.t["m_rev"]), _lib.ptr(g), G.N, A, Em, _lib.ptr(dh), _st()))
        return dh, g[:, 2 * A:].contiguous(), None


class _MulSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, r, G):
        m, r = _f32(m), _f32(r)
        out = _new(G.N, m.shape[1], like=m)
        _lib.check(_lib.load().nq_gn_mulsum_forward(C.byref(G.main), _lib.ptr(m), _lib.ptr(r), G.N, m.shape[1], _lib.ptr(out), _st()))
        ctx.save_for_backward(m, r)
        ctx.G = G
        return out

    @staticmethod
    def backward(ctx, g):
        m, r = ctx.saved_tensors
        g = _f32(g)
        dm, dr = torch.empty_like(m), torch.empty_like(r)
        _lib.check(_lib.load().nq_gn_mulsum_backward(C.byref(ctx.G.main), _lib.ptr(m), _lib.ptr(r), _lib.ptr(g), m.shape[1], _lib.ptr(dm), _lib.ptr(dr), _st()))
        return dm, dr, None


class _ForcesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, G, coupled):
        f = _f32(f).reshape(-1)
        out = _new(G.N, 3, like=f)
        _lib.check(_lib.load().nq_gn_forces_forward(C.byref(G.main), _lib.ptr(G.t["m_rev"]), _lib.ptr(f), G.N, int(coupled), _lib.ptr(out), _st()))
        ctx.meta = (G, coupled)
        return out

    @staticmethod
    def backward(ctx, g):
        G, coupled = ctx.meta
        g = _f32(g)
        df = _new(G.Em, 1, like=g)
        _lib.check(_lib.load().nq_gn_forces_backward(C.byref(G.main), _lib.ptr(G.t["m_rev"]), _lib.ptr(g), int(coupled), _lib.ptr(df), _st()))
        return df, None, None


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, zm1, z32):
        W = _f32(W)
        ctx.meta = (z32, W.shape)
        return _gather_raw(W, zm1, None, zm1.numel())

    @staticmethod
    def backward(ctx, g):
        z32, shape = ctx.meta
        g = _f32(g)
        dW = _new(*shape, like=g)
        _lib.check(_lib.load().nq_gn_embed_grad(_lib.ptr(z32), _lib.ptr(g), z32.numel(), shape[0], shape[1], _lib.ptr(dW), _st()))
        return dW, None, None


def _perm(x, rev):
    """x[id_swap]: id_swap is an involution, so the adjoint is the same gather."""
    return _GatherMulFn.apply(x, None, rev, lambda rows: _gather_raw(rows, rev, None, rev.numel()))


# ============================================================================================================================================
# modules (names = the reference's)
# ============================================================================================================================================
def _he_orthogonal_(w):
    """initializers.py:26-45: (semi-)orthogonal, standardised to zero mean / unit variance over the fan-in, scaled by 1 / sqrt(fan_in)."""
    with torch.no_grad():
        torch.nn.init.orthogonal_(w)
        if w.dim() == 3:
            axis, fan_in = (0, 1), w.shape[0] * w.shape[1]
        else:
            axis, fan_in = 1, w.shape[1]
        var, mean = torch.var_mean(w, dim=axis, unbiased=True, keepdim=True)
        w.copy_((w - mean) / (var + 1e-6) ** 0.5 * (1.0 / fan_in) ** 0.5)
    return w


class _Linear(torch.nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = torch.nn.Parameter(_he_orthogonal_(torch.empty(n_out, n_in)))
        self.register_parameter("bias", None)


class Dense(torch.nn.Module):
    """layers/base_layers.py:11-58 (bias is never used by GemNet-OC)."""

    def __init__(self, in_features, out_features, bias=False, activation=None):
        super().__init__()
        if bias:
            raise NotImplementedError("GemNet-OC builds every Dense with bias=False")
        act = activation.lower() if isinstance(activation, str) else activation
        if act not in (None, "silu", "swish"):
            raise NotImplementedError("Activation function not implemented for GemNet (yet).")
        self.linear = _Linear(in_features, out_features)
        self._act = act is not None

    def reset_parameters(self, initializer=_he_orthogonal_):
        initializer(self.linear.weight)

    def forward(self, x):
        return _DenseFn.apply(x, self.linear.weight, self._act)


class ResidualLayer(torch.nn.Module):
    """layers/base_layers.py:74-97."""

    def __init__(self, units, nLayers=2, activation=None):
        super().__init__()
        self.dense_mlp = torch.nn.Sequential(*[Dense(units, units, activation=activation) for _ in range(nLayers)])

    def forward(self, x):
        if len(self.dense_mlp) == 2 and self.dense_mlp[0]._act and self.dense_mlp[1]._act:
            return _ResidualFn.apply(x, self.dense_mlp[0].linear.weight, self.dense_mlp[1].linear.weight)
        return lin(x, self.dense_mlp(x), _INV_SQRT2, _INV_SQRT2)


class ScaleFactor(torch.nn.Module):
    """layers/scale_factor.py:27-154 without the fitting machinery: a frozen scalar, 0 = not fitted = identity.  The value is read once per change
    (the reference calls ``.item()`` on every forward)."""

    def __init__(self):
        super().__init__()
        self.scale_factor = torch.nn.Parameter(torch.tensor(0.0), requires_grad=False)
        self._cache = (None, None)

    def value(self) -> float:
        key = (self.scale_factor._version, self.scale_factor.data_ptr())
        if self._cache[0] != key:
            v = float(self.scale_factor.detach().cpu())
            self._cache = (key, v if v != 0.0 else 1.0)
        return self._cache[1]

    def forward(self, x):
        v = self.value()
        return x if v == 1.0 else lin(x, None, v, 0.0)


class AtomEmbedding(torch.nn.Module):
    """layers/embedding_block.py:20-55."""

    def __init__(self, emb_size, num_elements):
        super().__init__()
        self.embeddings = torch.nn.Embedding(num_elements, emb_size)
        torch.nn.init.uniform_(self.embeddings.weight, a=-math.sqrt(3), b=math.sqrt(3))

    def forward(self, G):
        z = G.t["z"]
        if not getattr(G, "z_checked", False):                              # one host read per batch: an atomic number outside the table would read out of bounds
            lo, hi = (int(v) for v in torch.stack([z.min(), z.max()]).tolist())
            if lo < 1 or hi > self.embeddings.num_embeddings:
                raise IndexError(f"atomic numbers {lo}..{hi} outside 1..{self.embeddings.num_embeddings} (num_elements)")
            G.z_checked = True
        return _EmbedFn.apply(self.embeddings.weight, (z - 1).contiguous(), z)


class EdgeEmbedding(torch.nn.Module):
    """layers/embedding_block.py:58-92."""

    def __init__(self, atom_features, edge_features, out_features, activation=None):
        super().__init__()
        self.dense = Dense(2 * atom_features + edge_features, out_features, activation=activation)

    def forward(self, h, m, G):
        return self.dense(_CatFn.apply(h, m, G))


class _GaussianBasis(torch.nn.Module):
    def __init__(self, num_gaussians):
        super().__init__()
        self.register_buffer("offset", torch.linspace(0.0, 1.0, num_gaussians))


class RadialBasis(torch.nn.Module):
    """layers/radial_basis.py:140-220, gaussian x polynomial envelope only."""

    def __init__(self, num_radial, cutoff, rbf, envelope, scale_basis=False):
        super().__init__()
        if rbf.get("name", "").lower() != "gaussian" or len(rbf) != 1:
            raise NotImplementedError(f"radial basis {rbf}: only the gaussian basis of config/model/gemnet-oc.yaml is built")
        if envelope.get("name", "").lower() != "polynomial":
            raise NotImplementedError(f"envelope {envelope}: only the polynomial envelope is built")
        self.cutoff, self.exponent, self.num_radial = float(cutoff), float(envelope["exponent"]), num_radial
        self.scale_basis = scale_basis
        if scale_basis:
            self.scale_rbf = ScaleFactor()
        self.rbf = _GaussianBasis(num_radial)

    def forward(self, geom):
        n = geom.shape[0]
        out = torch.empty(n, self.num_radial, device=geom.device, dtype=torch.float32)
        scale = self.scale_rbf.value() if self.scale_basis else 1.0
        _lib.check(_lib.load().nq_gn_radial_basis(_lib.ptr(geom), n, self.num_radial, _lib.ptr(self.rbf.offset), self.cutoff, self.exponent, scale, _lib.ptr(out),
                                                  _st()))
        return out


class CircularBasisLayer(torch.nn.Module):
    """layers/spherical_basis.py:15-66; the angular part Y_l0(cos) is evaluated inside the interaction kernels, this module holds its scale."""

    def __init__(self, num_spherical, radial_basis, cbf, scale_basis=False):
        super().__init__()
        if cbf.get("name", "").lower() != "spherical_harmonics":
            raise NotImplementedError(f"circular basis {cbf}: only spherical_harmonics is built")
        self.radial_basis = radial_basis
        self.scale_basis = scale_basis
        if scale_basis:
            self.scale_cbf = ScaleFactor()

    def scale(self):
        return self.scale_cbf.value() if self.scale_basis else 1.0


class SphericalBasisLayer(torch.nn.Module):
    """layers/spherical_basis.py:69-127 ("legendre_outer")."""

    def __init__(self, num_spherical, radial_basis, sbf, scale_basis=False):
        super().__init__()
        if sbf.get("name", "").lower() != "legendre_outer":
            raise NotImplementedError(f"spherical basis {sbf}: only legendre_outer is built")
        self.radial_basis = radial_basis
        self.scale_basis = scale_basis
        if scale_basis:
            self.scale_sbf = ScaleFactor()

    def scale(self):
        return self.scale_sbf.value() if self.scale_basis else 1.0


class BasisEmbedding(torch.nn.Module):
    """layers/efficient.py:14-149: only the product with the radial basis lives here -- ``rad @ W.reshape(R, -1)``, whose flat rows the kernels read as
    [interm][spherical] exactly like the reference's ``reshape(num_edges, -1, sph.shape[-1])`` (:103-104) -- the angular part is in the kernels."""

    def __init__(self, num_radial, emb_size_interm, num_spherical=None):
        super().__init__()
        shape = (emb_size_interm, num_radial) if num_spherical is None else (num_radial, num_spherical, emb_size_interm)
        self.weight = torch.nn.Parameter(_he_orthogonal_(torch.empty(*shape)))
        self.num_spherical = num_spherical

    def forward(self, rad):
        if self.num_spherical is None:
            return _DenseFn.apply(rad, self.weight, False)                       # rad @ W^T
        return _MatmulFn.apply(rad, self.weight.reshape(self.weight.shape[0], -1))


class EfficientInteractionBilinear(torch.nn.Module):
    """layers/efficient.py:152-253: the final Dense over the flattened [interm x in] products."""

    def __init__(self, emb_size_in, emb_size_interm, emb_size_out):
        super().__init__()
        self.emb_size_in, self.emb_size_interm = emb_size_in, emb_size_interm
        self.bilinear = Dense(emb_size_in * emb_size_interm, emb_size_out)


class TripletInteraction(torch.nn.Module):
    """layers/interaction_block.py:478-686.  ``kind``: "e2e" (main -> main), "a2e" (atoms via the a2ee2a graph -> main), "e2a" (main -> atoms)."""

    def __init__(self, emb_size_in, emb_size_out, emb_size_trip_in, emb_size_trip_out, emb_size_rbf, emb_size_cbf, symmetric_mp=True, swap_output=True,
                 activation=None):
        super().__init__()
        self.symmetric_mp, self.swap_output = symmetric_mp, swap_output
        self.dense_ba = Dense(emb_size_in, emb_size_in, activation=activation)
        self.mlp_rbf = Dense(emb_size_rbf, emb_size_in)
        self.scale_rbf = ScaleFactor()
        self.mlp_cbf = EfficientInteractionBilinear(emb_size_trip_in, emb_size_cbf, emb_size_trip_out)
        self.scale_cbf_sum = ScaleFactor()
        self.down_projection = Dense(emb_size_in, emb_size_trip_in, activation=activation)
        self.up_projection_ca = Dense(emb_size_trip_out, emb_size_out, activation=activation)
        if symmetric_mp:
            self.up_projection_ac = Dense(emb_size_trip_out, emb_size_out, activation=activation)

    def forward(self, x, bases, G, kind, NS):
        t = G.t
        x_ba = self.dense_ba(x)
        rad_emb = self.mlp_rbf(bases["rad"])
        if kind == "a2e":                                                     # x_ba[expand_idx] * rad_emb (:640-646)
            aor, rp = t["a_of_rev"], t["row_ptr"]
            x_ba = _GatherMulFn.apply(x_ba, rad_emb, t["a_src"], lambda rows: _segsum_raw(rows, aor, rp, G.N, rows.shape[1]))
        else:
            x_ba = _MulFn.apply(x_ba, rad_emb)
        x_ba = self.down_projection(self.scale_rbf(x_ba))
        out_set, in_set = {"e2e": (G.main, G.main), "a2e": (G.main, G.aea), "e2a": (G.aea, G.main)}[kind]
        Cc, I = self.mlp_cbf.emb_size_in, self.mlp_cbf.emb_size_interm
        S = _TripletFn.apply(x_ba, G, out_set, in_set, NS, bases["cir_scale"])
        X = _RowMMFn.apply(bases["cir"], S, I, NS, Cc)
        if kind == "e2a":                                                     # second aggregation over the a2ee2a edges of each target atom (efficient.py:231-240)
            X = _SegSumFn.apply(X, t["ptr_a"], t["a_dst"], G.N)
        x = self.scale_cbf_sum(self.mlp_cbf.bilinear(X))
        if self.symmetric_mp:
            return lin(self.up_projection_ca(x), _perm(self.up_projection_ac(x), t["m_rev"]), _INV_SQRT2, _INV_SQRT2)
        if self.swap_output:
            x = _perm(x, t["m_rev"])
        return self.up_projection_ca(x)


class QuadrupletInteraction(torch.nn.Module):
    """layers/interaction_block.py:343-475."""

    def __init__(self, emb_size_edge, emb_size_quad_in, emb_size_quad_out, emb_size_rbf, emb_size_cbf, emb_size_sbf, symmetric_mp=True, activation=None):
        super().__init__()
        self.symmetric_mp = symmetric_mp
        self.dense_db = Dense(emb_size_edge, emb_size_edge, activation=activation)
        self.mlp_rbf = Dense(emb_size_rbf, emb_size_edge)
        self.scale_rbf = ScaleFactor()
        self.mlp_cbf = Dense(emb_size_cbf, emb_size_quad_in)
        self.scale_cbf = ScaleFactor()
        self.mlp_sbf = EfficientInteractionBilinear(emb_size_quad_in, emb_size_sbf, emb_size_quad_out)
        self.scale_sbf_sum = ScaleFactor()
        self.down_projection = Dense(emb_size_edge, emb_size_quad_in, activation=activation)
        self.up_projection_ca = Dense(emb_size_quad_out, emb_size_edge, activation=activation)
        if symmetric_mp:
            self.up_projection_ac = Dense(emb_size_quad_out, emb_size_edge, activation=activation)

    def forward(self, m, bases, G, NS):
        t = G.t
        x_db = self.scale_rbf(_MulFn.apply(self.dense_db(m), self.mlp_rbf(bases["rad"])))
        x_db = self.down_projection(x_db)
        Cc, I = self.mlp_sbf.emb_size_in, self.mlp_sbf.emb_size_interm
        lib, main, rp, qor, tp = _lib.load(), G.main, t["row_ptr"], t["q_of_rev"], t["tin_ptr"]

        def scatter(rows):
            out = _new(G.Em, rows.shape[1], like=rows)
            _lib.check(lib.nq_gn_tin_scatter(C.byref(main), _lib.ptr(rp), _lib.ptr(qor), _lib.ptr(tp), _lib.ptr(rows), rows.shape[1], _lib.ptr(out), _st()))
            return out

        x_tin = self.scale_cbf(_GatherMulFn.apply(x_db, self.mlp_cbf(bases["cir"]), t["tin_main"][:G.Tin], scatter))       # x_db[triplet_in.in] * cbf (:579-581)
        S = _QuadFn.apply(x_tin, G, NS, bases["sph_scale"])
        X = _RowMMFn.apply(bases["sph"], S, I, NS * NS, Cc)
        x = self.scale_sbf_sum(self.mlp_sbf.bilinear(X))
        if self.symmetric_mp:
            return lin(self.up_projection_ca(x), _perm(self.up_projection_ac(x), t["m_rev"]), _INV_SQRT2, _INV_SQRT2)
        return self.up_projection_ca(x)


class PairInteraction(torch.nn.Module):
    """layers/interaction_block.py:689-739."""

    def __init__(self, emb_size_atom, emb_size_pair_in, emb_size_pair_out, emb_size_rbf, activation=None):
        super().__init__()
        self.bilinear = Dense(emb_size_rbf * emb_size_pair_in, emb_size_pair_out)
        self.scale_rbf_sum = ScaleFactor()
        self.down_projection = Dense(emb_size_atom, emb_size_pair_in, activation=activation)
        self.up_projection = Dense(emb_size_pair_out, emb_size_atom, activation=activation)

    def forward(self, h, rad_basis, G):
        x = _PairFn.apply(rad_basis, self.down_projection(h), G)
        return self.up_projection(self.scale_rbf_sum(self.bilinear(x)))


class AtomUpdateBlock(torch.nn.Module):
    """layers/atom_update_block.py:15-97."""

    def __init__(self, emb_size_atom, emb_size_edge, emb_size_rbf, nHidden, activation=None):
        super().__init__()
        self.dense_rbf = Dense(emb_size_rbf, emb_size_edge)
        self.scale_sum = ScaleFactor()
        self.layers = self.get_mlp(emb_size_edge, emb_size_atom, nHidden, activation)

    @staticmethod
    def get_mlp(units_in, units, nHidden, activation):
        mlp = [Dense(units_in, units, activation=activation)] if units_in != units else []
        return torch.nn.ModuleList(mlp + [ResidualLayer(units, nLayers=2, activation=activation) for _ in range(nHidden)])

    def forward(self, h, m, basis_rad, G):
        x = self.scale_sum(_MulSumFn.apply(m, self.dense_rbf(basis_rad), G))
        for layer in self.layers:
            x = layer(x)
        return x


class OutputBlock(AtomUpdateBlock):
    """layers/atom_update_block.py:100-172."""

    def __init__(self, emb_size_atom, emb_size_edge, emb_size_rbf, nHidden, nHidden_afteratom, activation=None, direct_forces=True):
        super().__init__(emb_size_atom, emb_size_edge, emb_size_rbf, nHidden, activation)
        self.direct_forces = direct_forces
        self.seq_energy_pre = self.layers
        self.seq_energy2 = self.get_mlp(emb_size_atom, emb_size_atom, nHidden_afteratom, activation) if nHidden_afteratom >= 1 else None
        if direct_forces:
            self.scale_rbf_F = ScaleFactor()
            self.seq_forces = self.get_mlp(emb_size_edge, emb_size_edge, nHidden, activation)
            self.dense_rbf_F = Dense(emb_size_rbf, emb_size_edge)

    def forward(self, h, m, basis_rad, G):
        x_E = self.scale_sum(_MulSumFn.apply(m, self.dense_rbf(basis_rad), G))
        for layer in self.seq_energy_pre:
            x_E = layer(x_E)
        if self.seq_energy2 is not None:
            x_E = lin(x_E, h, _INV_SQRT2, _INV_SQRT2)
            for layer in self.seq_energy2:
                x_E = layer(x_E)
        x_F = None
        if self.direct_forces:
            x_F = m
            for layer in self.seq_forces:
                x_F = layer(x_F)
            x_F = self.scale_rbf_F(_MulFn.apply(x_F, self.dense_rbf_F(basis_rad)))
        return x_E, x_F


class InteractionBlock(torch.nn.Module):
    """layers/interaction_block.py:17-340."""

    def __init__(self, emb_size_atom, emb_size_edge, emb_size_trip_in, emb_size_trip_out, emb_size_quad_in, emb_size_quad_out, emb_size_a2a_in,
                 emb_size_a2a_out, emb_size_rbf, emb_size_cbf, emb_size_sbf, num_before_skip, num_after_skip, num_concat, num_atom, num_atom_emb_layers=0,
                 quad_interaction=False, atom_edge_interaction=False, edge_atom_interaction=False, atom_interaction=False, activation=None):
        super().__init__()
        self.dense_ca = Dense(emb_size_edge, emb_size_edge, activation=activation)
        self.trip_interaction = TripletInteraction(emb_size_edge, emb_size_edge, emb_size_trip_in, emb_size_trip_out, emb_size_rbf, emb_size_cbf, True, True,
                                                   activation)
        self.quad_interaction = (QuadrupletInteraction(emb_size_edge, emb_size_quad_in, emb_size_quad_out, emb_size_rbf, emb_size_cbf, emb_size_sbf, True,
                                                       activation) if quad_interaction else None)
        self.atom_edge_interaction = (TripletInteraction(emb_size_atom, emb_size_edge, emb_size_trip_in, emb_size_trip_out, emb_size_rbf, emb_size_cbf, True,
                                                         True, activation) if atom_edge_interaction else None)
        self.edge_atom_interaction = (TripletInteraction(emb_size_edge, emb_size_atom, emb_size_trip_in, emb_size_trip_out, emb_size_rbf, emb_size_cbf, False,
                                                         False, activation) if edge_atom_interaction else None)
        self.atom_interaction = PairInteraction(emb_size_atom, emb_size_a2a_in, emb_size_a2a_out, emb_size_rbf, activation) if atom_interaction else None
        self.layers_before_skip = torch.nn.ModuleList([ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_before_skip)])
        self.layers_after_skip = torch.nn.ModuleList([ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_after_skip)])
        self.atom_emb_layers = torch.nn.ModuleList([ResidualLayer(emb_size_atom, activation=activation) for _ in range(num_atom_emb_layers)])
        self.atom_update = AtomUpdateBlock(emb_size_atom, emb_size_edge, emb_size_rbf, num_atom, activation)
        self.concat_layer = EdgeEmbedding(emb_size_atom, emb_size_edge, emb_size_edge, activation=activation)
        self.residual_m = torch.nn.ModuleList([ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_concat)])
        self.inv_sqrt_num_eint = 1.0 / math.sqrt(2.0 + bool(quad_interaction) + bool(atom_edge_interaction))
        self.inv_sqrt_num_aint = 1.0 / math.sqrt(1.0 + bool(edge_atom_interaction) + bool(atom_interaction))

    def forward(self, h, m, B, G, NS):
        x = lin(self.dense_ca(m), self.trip_interaction(m, B["e2e"], G, "e2e", NS))
        if self.quad_interaction is not None:
            x = lin(x, self.quad_interaction(m, B["qint"], G, NS))
        if self.atom_edge_interaction is not None:
            x = lin(x, self.atom_edge_interaction(h, B["a2e"], G, "a2e", NS))
        x = lin(x, None, self.inv_sqrt_num_eint, 0.0)
        if self.edge_atom_interaction is not None:
            h_e2a = self.edge_atom_interaction(m, B["e2a"], G, "e2a", NS)
        if self.atom_interaction is not None:
            h_a2a = self.atom_interaction(h, B["a2a_rad"], G)
        if self.edge_atom_interaction is not None:
            h = lin(h, h_e2a)
        if self.atom_interaction is not None:
            h = lin(h, h_a2a)
        h = lin(h, None, self.inv_sqrt_num_aint, 0.0)
        for layer in self.layers_before_skip:
            x = layer(x)
        m = lin(m, x, _INV_SQRT2, _INV_SQRT2)
        for layer in self.layers_after_skip:
            m = layer(m)
        for layer in self.atom_emb_layers:
            h = layer(h)
        h = lin(h, self.atom_update(h, m, B["atom_update"], G), _INV_SQRT2, _INV_SQRT2)
        m2 = self.concat_layer(h, m, G)
        for layer in self.residual_m:
            m2 = layer(m2)
        m = lin(m, m2, _INV_SQRT2, _INV_SQRT2)
        return h, m


class GemNetOC(torch.nn.Module):
    """gemnet_oc/gemnet_oc.py:36-1340 (constructor arguments :162-228)."""

    def __init__(self, num_targets: int, num_spherical: int, num_radial: int, num_blocks: int, emb_size_atom: int, emb_size_edge: int, emb_size_trip_in: int,
                 emb_size_trip_out: int, emb_size_quad_in: int, emb_size_quad_out: int, emb_size_aint_in: int, emb_size_aint_out: int, emb_size_rbf: int,
                 emb_size_cbf: int, emb_size_sbf: int, num_before_skip: int, num_after_skip: int, num_concat: int, num_atom: int,
                 num_output_afteratom: int, num_atom_emb_layers: int = 0, num_global_out_layers: int = 2, regress_forces: bool = True,
                 direct_forces: bool = False, use_pbc: bool = True, scale_backprop_forces: bool = False, cutoff: float = 6.0,
                 cutoff_qint: Optional[float] = None, cutoff_aeaint: Optional[float] = None, cutoff_aint: Optional[float] = None, max_neighbors: int = 50,
                 max_neighbors_qint: Optional[int] = None, max_neighbors_aeaint: Optional[int] = None, max_neighbors_aint: Optional[int] = None,
                 enforce_max_neighbors_strictly: bool = True, rbf: Dict[str, str] = {"name": "gaussian"}, rbf_spherical: Optional[dict] = None,
                 envelope: Dict[str, Union[str, int]] = {"name": "polynomial", "exponent": 5}, cbf: Dict[str, str] = {"name": "spherical_harmonics"},
                 sbf: Dict[str, str] = {"name": "spherical_harmonics"}, extensive: bool = True, forces_coupled: bool = False, output_init: str = "HeOrthogonal",
                 activation: str = "silu", quad_interaction: bool = False, atom_edge_interaction: bool = False, edge_atom_interaction: bool = False,
                 atom_interaction: bool = False, scale_basis: bool = False, num_elements: int = 83, otf_graph: bool = False,
                 scale_file: Optional[str] = None) -> None:
        super().__init__()
        _PACKED.clear()               # packed bf16 weights are keyed by address: a new model may reuse the addresses of a freed one
        weights_epoch_advance()
        for ok, what in ((num_targets == 1, "num_targets != 1"), (not use_pbc, "periodic boundary conditions"), (regress_forces and direct_forces,
                         "forces by back-propagation (direct_forces=False)"), (enforce_max_neighbors_strictly, "enforce_max_neighbors_strictly=False"),
                         (not scale_backprop_forces, "scale_backprop_forces"), (extensive, "extensive=False"), (scale_file is None, "scale_file"),
                         (quad_interaction and atom_edge_interaction and edge_atom_interaction and atom_interaction, "switching interactions off"),
                         (output_init.lower() == "heorthogonal", f"output_init={output_init}"), (num_spherical <= 8 and emb_size_rbf <= 16,
                         "num_spherical > 8 or emb_size_rbf > 16")):
            if not ok:
                raise NotImplementedError(f"GemNetOC on MI355X: {what} is not built (config/model/gemnet-oc.yaml is the supported configuration)")
        self.num_targets, self.num_blocks, self.extensive, self.num_spherical = num_targets, num_blocks, extensive, num_spherical
        self.atom_edge_interaction, self.edge_atom_interaction = atom_edge_interaction, edge_atom_interaction
        self.atom_interaction, self.quad_interaction, self.otf_graph = atom_interaction, quad_interaction, otf_graph
        rbf_spherical = rbf_spherical or rbf
        self.cutoff = cutoff                                                         # set_cutoffs / set_max_neighbors (:375-428)
        self.cutoff_aeaint = cutoff if cutoff_aeaint is None else cutoff_aeaint
        self.cutoff_qint = cutoff if cutoff_qint is None else cutoff_qint
        self.cutoff_aint = max(self.cutoff, self.cutoff_aeaint, self.cutoff_qint) if cutoff_aint is None else cutoff_aint
        assert self.cutoff <= self.cutoff_aint and self.cutoff_aeaint <= self.cutoff_aint and self.cutoff_qint <= self.cutoff_aint
        self.max_neighbors = max_neighbors
        self.max_neighbors_aeaint = max_neighbors if max_neighbors_aeaint is None else max_neighbors_aeaint
        self.max_neighbors_qint = max_neighbors if max_neighbors_qint is None else max_neighbors_qint
        self.max_neighbors_aint = (max(self.max_neighbors, self.max_neighbors_aeaint, self.max_neighbors_qint) if max_neighbors_aint is None
                                   else max_neighbors_aint)
        assert (self.max_neighbors <= self.max_neighbors_aint and self.max_neighbors_aeaint <= self.max_neighbors_aint
                and self.max_neighbors_qint <= self.max_neighbors_aint)
        self.enforce_max_neighbors_strictly, self.use_pbc = enforce_max_neighbors_strictly, use_pbc
        self.direct_forces, self.forces_coupled, self.regress_forces = direct_forces, forces_coupled, regress_forces

        # init_basis_functions (:430-531), in the reference's registration order; radial_basis_spherical is ONE instance shared by three layers
        def rb(c, spec):
            return RadialBasis(num_radial, c, spec, envelope, scale_basis)

        self.radial_basis = rb(self.cutoff, rbf)
        radial_basis_spherical = rb(self.cutoff, rbf_spherical)
        self.cbf_basis_qint = CircularBasisLayer(num_spherical, rb(self.cutoff_qint, rbf_spherical), cbf, scale_basis)
        self.sbf_basis_qint = SphericalBasisLayer(num_spherical, radial_basis_spherical, sbf, scale_basis)
        self.radial_basis_aeaint = rb(self.cutoff_aeaint, rbf)
        self.cbf_basis_aeint = CircularBasisLayer(num_spherical, radial_basis_spherical, cbf, scale_basis)
        self.cbf_basis_eaint = CircularBasisLayer(num_spherical, rb(self.cutoff_aeaint, rbf_spherical), cbf, scale_basis)
        self.radial_basis_aint = rb(self.cutoff_aint, rbf)
        self.cbf_basis_tint = CircularBasisLayer(num_spherical, radial_basis_spherical, cbf, scale_basis)
        # init_shared_basis_layers (:533-622)
        self.mlp_rbf_qint = Dense(num_radial, emb_size_rbf)
        self.mlp_cbf_qint = BasisEmbedding(num_radial, emb_size_cbf, num_spherical)
        self.mlp_sbf_qint = BasisEmbedding(num_radial, emb_size_sbf, num_spherical ** 2)
        self.mlp_rbf_aeint = Dense(num_radial, emb_size_rbf)
        self.mlp_cbf_aeint = BasisEmbedding(num_radial, emb_size_cbf, num_spherical)
        self.mlp_rbf_eaint = Dense(num_radial, emb_size_rbf)
        self.mlp_cbf_eaint = BasisEmbedding(num_radial, emb_size_cbf, num_spherical)
        self.mlp_rbf_aint = BasisEmbedding(num_radial, emb_size_rbf)
        self.mlp_rbf_tint = Dense(num_radial, emb_size_rbf)
        self.mlp_cbf_tint = BasisEmbedding(num_radial, emb_size_cbf, num_spherical)
        self.mlp_rbf_h = Dense(num_radial, emb_size_rbf)
        self.mlp_rbf_out = Dense(num_radial, emb_size_rbf)
        self.shared_parameters = [(self.mlp_rbf_tint.linear.weight, num_blocks), (self.mlp_cbf_tint.weight, num_blocks), (self.mlp_rbf_h.linear.weight, num_blocks),
                                  (self.mlp_rbf_out.linear.weight, num_blocks + 1), (self.mlp_rbf_qint.linear.weight, num_blocks),
                                  (self.mlp_cbf_qint.weight, num_blocks), (self.mlp_sbf_qint.weight, num_blocks), (self.mlp_rbf_aeint.linear.weight, num_blocks),
                                  (self.mlp_cbf_aeint.weight, num_blocks), (self.mlp_rbf_eaint.linear.weight, num_blocks), (self.mlp_cbf_eaint.weight, num_blocks),
                                  (self.mlp_rbf_aint.weight, num_blocks)]
        self.atom_emb = AtomEmbedding(emb_size_atom, num_elements)
        self.edge_emb = EdgeEmbedding(emb_size_atom, num_radial, emb_size_edge, activation=activation)
        self.int_blocks = torch.nn.ModuleList([
            InteractionBlock(emb_size_atom, emb_size_edge, emb_size_trip_in, emb_size_trip_out, emb_size_quad_in, emb_size_quad_out, emb_size_aint_in,
                             emb_size_aint_out, emb_size_rbf, emb_size_cbf, emb_size_sbf, num_before_skip, num_after_skip, num_concat, num_atom,
                             num_atom_emb_layers, quad_interaction, atom_edge_interaction, edge_atom_interaction, atom_interaction, activation)
            for _ in range(num_blocks)])
        self.out_blocks = torch.nn.ModuleList([OutputBlock(emb_size_atom, emb_size_edge, emb_size_rbf, num_atom, num_output_afteratom, activation, direct_forces)
                                               for _ in range(num_blocks + 1)])
        self.out_mlp_E = torch.nn.Sequential(Dense(emb_size_atom * (num_blocks + 1), emb_size_atom, activation=activation),
                                             *[ResidualLayer(emb_size_atom, activation=activation) for _ in range(num_global_out_layers)])
        self.out_energy = Dense(emb_size_atom, num_targets)
        self.out_mlp_F = torch.nn.Sequential(Dense(emb_size_edge * (num_blocks + 1), emb_size_edge, activation=activation),
                                             *[ResidualLayer(emb_size_edge, activation=activation) for _ in range(num_global_out_layers)])
        self.out_forces = Dense(emb_size_edge, num_targets)

    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def get_graphs_and_indices(self, data) -> GemNetGraphs:
        return build_graphs(data.pos, data.batch, data.z, self.cutoff, self.cutoff_qint, self.cutoff_aeaint, self.cutoff_aint, self.max_neighbors,
                            self.max_neighbors_qint, self.max_neighbors_aeaint, self.max_neighbors_aint, getattr(data, "ptr", None))

    def get_bases(self, G: GemNetGraphs):
        """gemnet_oc.py:999-1116 without the per-triplet / per-quadruplet tensors (those are evaluated inside the kernels)."""
        t = G.t
        rad_main = self.radial_basis(t["m_geom"])
        rad_sph = self.cbf_basis_tint.radial_basis(t["m_geom"])                                  # the shared radial_basis_spherical on the main distances
        rad_aea = self.radial_basis_aeaint(t["a_geom"])
        B = {"rad_main_raw": rad_main,
             "atom_update": self.mlp_rbf_h(rad_main), "output": self.mlp_rbf_out(rad_main),
             "qint": {"rad": self.mlp_rbf_qint(rad_main),
                      "cir": _CirFn.apply(self.mlp_cbf_qint(self.cbf_basis_qint.radial_basis(t["q_geom"])), G, self.mlp_cbf_qint.weight.shape[2],
                                          self.num_spherical, self.cbf_basis_qint.scale()),
                      "sph": self.mlp_sbf_qint(rad_sph), "sph_scale": self.sbf_basis_qint.scale()},
             "e2e": {"rad": self.mlp_rbf_tint(rad_main), "cir": self.mlp_cbf_tint(rad_sph), "cir_scale": self.cbf_basis_tint.scale()},
             "a2e": {"rad": self.mlp_rbf_aeint(rad_aea), "cir": self.mlp_cbf_aeint(rad_sph), "cir_scale": self.cbf_basis_aeint.scale()},
             "e2a": {"rad": self.mlp_rbf_eaint(rad_main), "cir": self.mlp_cbf_eaint(self.cbf_basis_eaint.radial_basis(t["a_geom"])),
                     "cir_scale": self.cbf_basis_eaint.scale()},
             "a2a_rad": self.mlp_rbf_aint(self.radial_basis_aint(t["geom"]))}
        return B

    def prepare(self, data):
        """The four graphs, their index lists and the atomic-number range check of a batch (every host read of the step happens here).
        ``data.prepared = net.prepare(data)`` makes ``forward(data)`` free of host synchronisation (trainer.GraphedStep can capture the step)."""
        G = self.get_graphs_and_indices(data)
        with torch.no_grad():
            self.atom_emb(G)                            # runs the range check once (sets G.z_checked)
        G.geometry_key = _lib.geometry_key(data)        # checked by forward: a prepared batch is tied to its geometry
        return G

    def forward(self, data, return_intermediates: bool = False):
        weights_epoch_advance()   # bf16 weight copies are re-packed once per forward (parameters may have been updated in place)
        if not data.pos.is_cuda:
            raise RuntimeError("nabladft_amd.GemNetOC runs on MI355X only: tensors must be on a cuda (HIP) device")
        G = getattr(data, "prepared", None)            # prepare(data) done ahead: the forward then issues no host synchronisation (HIP-graph capture)
        if G is None:
            G = self.get_graphs_and_indices(data)
        elif G.N != int(data.pos.shape[0]):
            raise ValueError("data.prepared belongs to another batch")
        else:
            _lib.check_prepared(G, data)
        B = self.get_bases(G)
        NS = self.num_spherical
        h = self.atom_emb(G)
        m = self.edge_emb(h, B["rad_main_raw"], G)
        inter = {"graphs": G, "bases": B, "atom_emb": h, "edge_emb": m}
        x_E, x_F = self.out_blocks[0](h, m, B["output"], G)
        xs_E, xs_F = [x_E], [x_F]
        inter["out0"] = (x_E, x_F)
        for i in range(self.num_blocks):
            h, m = self.int_blocks[i](h, m, B, G, NS)
            x_E, x_F = self.out_blocks[i + 1](h, m, B["output"], G)
            xs_E.append(x_E)
            xs_F.append(x_F)
            inter[f"int{i}"], inter[f"out{i + 1}"] = (h, m), (x_E, x_F)
        E_t = self.out_energy(self.out_mlp_E(torch.cat(xs_E, dim=-1)))                           # [N, 1]
        F_st = self.out_forces(self.out_mlp_F(torch.cat(xs_F, dim=-1)))                          # [E, 1]
        energy = _SegSumFn.apply(E_t, G.t["mol_ptr"], G.t["atom_mol"], G.B).squeeze(1)           # extensive: sum per molecule (:1209-1211)
        forces = _ForcesFn.apply(F_st, G, self.forces_coupled)
        if return_intermediates:
            return energy, forces, inter
        return energy, forces
