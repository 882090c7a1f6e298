"""EquiformerV2 on the HIP kernels of csrc/equiformer.hip, csrc/escn.hip and the fp32 MFMA GEMMs -- host-side mirror of the reference's
``nablaDFT.equiformer_v2.EquiformerV2_OC20`` (equiformer_v2/equiformer_v2_oc20.py:51-640; config/model/equiformer_v2_oc20.yaml): same constructor arguments,
same module tree (``state_dict`` keys and shapes equal, buffers of the shared helper modules included), same outputs ``(energy [B], forces [N, 3])``.

What runs where
  * radius graph with the neighbour cap, edge vectors, edge frames, Wigner-D rows: the eSCN kernels (csrc/escn.hip); the rows are written in the m-primary order
    of CoefficientMappingModule (so3.py:20-139) and only for |m| <= mmax, so ``_rotate`` + ``_m_primary`` of the reference are one row operator;
  * SO(2) convolutions (so2_ops.py:13-194), radial functions, grid MLPs: MFMA GEMMs; torch.nn.LayerNorm, the equivariant layer norm, attention logits,
    the softmax over in-edges, messages x attention weights, rotate_inv's rescale and drop-path: csrc/equiformer.hip;
  * S2 activations and the FFN grids: the row operator with the constant grid matrices; SO3_LinearV2: the per-degree grouped GEMM (nq_sph_linear_*).
Constants the reference takes from e3nn (ToS2Grid / FromS2Grid, "component" normalisation) are computed by nabladft_amd.escn (own harmonics and quadrature) and
rescaled per degree: PARITY UNPINNED for e3nn's exact grid, like eSCN.  Edge frames use a deterministic helper axis where the reference draws a random vector
(edge_rot_mat.py:15); ``forward(data, edge_rot_mat=...)`` accepts given frames (parity tests).
Only the options of the yaml are built (one resolution, gaussian distance expansion, per-block atom edge embeddings, separable S2 activation, grid MLP,
layer_norm_sh / no attention after S2 / no gate activation / no m-shared radial function, no periodic cells).  No CPU path.
"""
import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .escn import (CoefficientOrder, GaussianSmearing, _BlocksInFn, _BlocksOutFn, _S2ActBlocksFn, s2_activation_fusable, _EmbeddingFn, _RotateBackFn, _RotateFn, _RowFn, _silu, eSCN, j_matrices,
                   s2_grids)
from . import gemnet_oc as _gemnet
from .gemnet_oc import _DenseFn, _MulFn, _SO2PairFn, _SegSumFn, _new, _st, fused_pairs_available, lin
from .phisnet import _SphLinearFn
from .qhnet import _LinearBiasFn, _f32

_AVG_NUM_NODES = 39.65745326960467          # equiformer_v2_oc20.py:47-48
_AVG_DEGREE = 19.16009564536883


class CosineLRLambda:
    """lr_scheduler.py:33-50 (the ``lr_lambda`` of config/model/equiformer_v2_oc20.yaml:47-55): linear warm-up from ``warmup_factor`` to 1 over ``warmup_epochs``
    steps, then half a cosine down to ``lr_min_factor`` at ``epochs``."""

    def __init__(self, scheduler_params) -> None:
        self.warmup_epochs, self.lr_warmup_factor = scheduler_params["warmup_epochs"], scheduler_params["warmup_factor"]
        self.max_epochs, self.lr_min_factor = scheduler_params["epochs"], scheduler_params["lr_min_factor"]

    def __call__(self, current_step: int) -> float:
        if current_step <= self.warmup_epochs:
            a = current_step / float(self.warmup_epochs)
            return self.lr_warmup_factor * (1.0 - a) + a
        if current_step >= self.max_epochs:
            return self.lr_min_factor
        return self.lr_min_factor + 0.5 * (1 - self.lr_min_factor) * (1 + math.cos(math.pi * current_step / self.max_epochs))


# ---- autograd wrappers of csrc/equiformer.hip ----------------------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    """torch.nn.LayerNorm over the last axis of x [rows, W]."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x, weight, bias = _f32(x), _f32(weight), _f32(bias)
        rows, W = x.shape
        y, stats = torch.empty_like(x), _new(rows, 2, like=x)
        _lib.check(_lib.load().nq_eq_layernorm_forward(_lib.ptr(x), W, _lib.ptr(weight), _lib.ptr(bias), rows, W, eps, _lib.ptr(y), W, _lib.ptr(stats), _st()))
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight, stats = ctx.saved_tensors
        g = _f32(g)
        rows, W = x.shape
        gx, gw, gb = torch.empty_like(x), torch.empty_like(weight), torch.empty_like(weight)
        scr = _new(int(lib.nq_eq_layernorm_scratch_floats(rows, W)), like=x)
        _lib.check(lib.nq_eq_layernorm_backward(_lib.ptr(x), W, _lib.ptr(weight), _lib.ptr(g), W, _lib.ptr(stats), rows, W, _lib.ptr(gx), W, _lib.ptr(gw),
                                                _lib.ptr(gb), _lib.ptr(scr), _st()))
        return gx, gw, gb, None


class _NormShFn(torch.autograd.Function):
    """EquivariantLayerNormArraySphericalHarmonics on x [N, (lmax+1)^2 * C]."""

    @staticmethod
    def forward(ctx, x, w0, b0, aw, bw, lmax, Cc, eps):
        x, w0, b0, aw = _f32(x), _f32(w0), _f32(b0), _f32(aw)
        N = x.shape[0]
        y, stats = torch.empty_like(x), _new(N, 3, like=x)
        _lib.check(_lib.load().nq_eq_norm_sh_forward(_lib.ptr(x), _lib.ptr(w0), _lib.ptr(b0), _lib.ptr(aw), _lib.ptr(bw), N, lmax, Cc, eps, _lib.ptr(y),
                                                     _lib.ptr(stats), _st()))
        ctx.save_for_backward(x, w0, aw, bw, stats)
        ctx.meta = (lmax, Cc)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w0, aw, bw, stats = ctx.saved_tensors
        lmax, Cc = ctx.meta
        g = _f32(g)
        N = x.shape[0]
        gx, gw0, gb0, gaw = torch.empty_like(x), torch.empty_like(w0), torch.empty_like(w0), torch.empty_like(aw)
        scr = _new(int(lib.nq_eq_norm_sh_scratch_floats(N, lmax, Cc)), like=x)
        _lib.check(lib.nq_eq_norm_sh_backward(_lib.ptr(x), _lib.ptr(w0), _lib.ptr(aw), _lib.ptr(bw), _lib.ptr(g), _lib.ptr(stats), N, lmax, Cc, _lib.ptr(gx),
                                              _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gaw), _lib.ptr(scr), _st()))
        return gx, gw0, gb0, gaw, None, None, None, None


class _LogitsFn(torch.autograd.Function):
    """z[e, h] = sum_a alpha_dot[h, a] SmoothLeakyReLU(x[e, h, a]); x [E, H * A]."""

    @staticmethod
    def forward(ctx, x, alpha_dot):
        x, alpha_dot = _f32(x), _f32(alpha_dot)
        H, A = alpha_dot.shape
        E = x.shape[0]
        z = _new(E, H, like=x)
        _lib.check(_lib.load().nq_eq_logits_forward(_lib.ptr(x), _lib.ptr(alpha_dot), E, H, A, _lib.ptr(z), _st()))
        ctx.save_for_backward(x, alpha_dot)
        return z

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, alpha_dot = ctx.saved_tensors
        H, A = alpha_dot.shape
        E = x.shape[0]
        g = _f32(g)
        gx, gad = torch.empty_like(x), torch.empty_like(alpha_dot)
        scr = _new(int(lib.nq_eq_logits_scratch_floats(E, H, A)), like=x)
        _lib.check(lib.nq_eq_logits_backward(_lib.ptr(x), _lib.ptr(alpha_dot), _lib.ptr(g), E, H, A, _lib.ptr(gx), _lib.ptr(gad), _lib.ptr(scr), _st()))
        return gx, gad


class _SoftmaxFn(torch.autograd.Function):
    """Softmax of z [E, H] over the in-edges of every target atom (edges sorted by target; ptr [N + 1])."""

    @staticmethod
    def forward(ctx, z, ptr, N):
        z = _f32(z)
        y = torch.zeros_like(z)
        _lib.check(_lib.load().nq_eq_softmax_forward(_lib.ptr(z), _lib.ptr(ptr), N, z.shape[1], _lib.ptr(y), _st()))
        ctx.save_for_backward(y)
        ctx.meta = (ptr, N)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        ptr, N = ctx.meta
        g = _f32(g)
        gz = torch.zeros_like(y)
        _lib.check(_lib.load().nq_eq_softmax_backward(_lib.ptr(y), _lib.ptr(g), _lib.ptr(ptr), N, y.shape[1], _lib.ptr(gz), _st()))
        return gz, None, None


def _head_scale(rows, xs, gs, alpha, H, V, want_alpha):
    k = len(rows)
    E = alpha.shape[0]
    outs = [torch.empty_like(x) for x in xs]
    ga = _new(E, H, like=alpha) if want_alpha else None
    rows_c = (C.c_int32 * k)(*rows)
    xp = (C.c_void_p * k)(*[x.data_ptr() for x in xs])
    gp = None if gs is None else (C.c_void_p * k)(*[g.data_ptr() for g in gs])
    op = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
    _lib.check(_lib.load().nq_eq_head_scale(k, rows_c, xp, gp, _lib.ptr(alpha), E, H, V, op, _lib.ptr(ga), _st()))
    return outs, ga


class _HeadScaleFn(torch.autograd.Function):
    """out_b[e, r, h, v] = x_b[e, r, h, v] * alpha[e, h] for the per-block message tensors x_b [E, rows_b * H * V]."""

    @staticmethod
    def forward(ctx, alpha, rows, H, V, *xs):
        alpha = _f32(alpha)
        xs = [_f32(x) for x in xs]
        ctx.meta = (rows, H, V)
        ctx.save_for_backward(alpha, *xs)
        outs, _ = _head_scale(rows, xs, None, alpha, H, V, False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        alpha, *xs = ctx.saved_tensors
        rows, H, V = ctx.meta
        gs = [_f32(g) for g in gs]
        gx, ga = _head_scale(rows, xs, gs, alpha, H, V, True)
        return (ga, None, None, None) + tuple(gx)


def _scale_raw(x, row_scale, row_index, coef_scale, I, Cc):
    out = torch.empty_like(x)
    _lib.check(_lib.load().nq_eq_scale(_lib.ptr(x), _lib.ptr(row_scale), _lib.ptr(row_index), _lib.ptr(coef_scale), x.shape[0], I, Cc, _lib.ptr(out), _st()))
    return out


class _ScaleFn(torch.autograd.Function):
    """x [N, I * C] times constant per-row (optionally through an index) and per-coefficient factors."""

    @staticmethod
    def forward(ctx, x, row_scale, row_index, coef_scale, I, Cc):
        ctx.meta = (row_scale, row_index, coef_scale, I, Cc)
        return _scale_raw(_f32(x), row_scale, row_index, coef_scale, I, Cc)

    @staticmethod
    def backward(ctx, g):
        row_scale, row_index, coef_scale, I, Cc = ctx.meta
        return _scale_raw(_f32(g), row_scale, row_index, coef_scale, I, Cc), None, None, None, None, None


def _linear(mod, x, act=False):
    """torch.nn.Linear parameters on the MFMA GEMM (SiLU fused when ``act``)."""
    if mod.bias is None:
        return _DenseFn.apply(x, mod.weight, 1.0 if act else False)
    return _LinearBiasFn.apply(x, mod.weight, mod.bias, act)


# ---- helper modules whose buffers are part of the reference's state_dict ----------------------------------------------------------------------------------------
class CoefficientMappingModule(nn.Module):
    """so3.py:20-139: the index buffers (one resolution)."""

    def __init__(self, lmax_list: List[int], mmax_list: List[int]):
        super().__init__()
        self.lmax_list, self.mmax_list = lmax_list, mmax_list
        assert len(lmax_list) == 1
        lmax, mmax = lmax_list[0], mmax_list[0]
        lm = [(l, m) for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]
        n = len(lm)
        to_m = torch.zeros(n, n)
        m_size, off = [], 0
        for m in range(mmax + 1):
            idx_r = [i for i, (l, mm) in enumerate(lm) if mm == m]
            idx_i = [i for i, (l, mm) in enumerate(lm) if mm == -m] if m else []
            for o, i in enumerate(idx_r + idx_i):
                to_m[off + o, i] = 1.0
            off += len(idx_r) + len(idx_i)
            m_size.append(len(idx_r))
        self.register_buffer("l_harmonic", torch.tensor([l for l, _ in lm], dtype=torch.long))
        self.register_buffer("m_harmonic", torch.tensor([abs(m) for _, m in lm], dtype=torch.long))
        self.register_buffer("m_complex", torch.tensor([m for _, m in lm], dtype=torch.long))
        self.register_buffer("res_size", torch.tensor([n], dtype=torch.long))
        self.register_buffer("to_m", to_m)
        self.register_buffer("m_size", torch.tensor(m_size, dtype=torch.long))


class SO3_Rotation(nn.Module):
    """so3.py:312-364: holds the mapping buffers; the Wigner rows themselves are per batch (graph stage)."""

    def __init__(self, lmax: int):
        super().__init__()
        self.lmax = lmax
        self.mapping = CoefficientMappingModule([lmax], [lmax])


def _degree_factors(lmax, fn):
    return np.concatenate([np.full(2 * l + 1, fn(l)) for l in range(lmax + 1)])


def so3_grid_mats(lmax, mmax, normalization="component"):
    """(to_grid_mat, from_grid_mat) of SO3_Grid(lmax, mmax) (so3.py:367-429): [lat, long, kept coefficients] in l-primary order, float64."""
    T, F = s2_grids(lmax, mmax)                                              # integral-normalised harmonics / quadrature, all (l, m)
    nb = 2 * (lmax + 1)
    a = {"integral": lambda l: 1.0, "component": lambda l: math.sqrt(4 * math.pi / ((2 * l + 1) * (lmax + 1))),
         "norm": lambda l: math.sqrt(4 * math.pi / (lmax + 1))}[normalization]
    f = _degree_factors(lmax, a)
    T, F = T * f, F / f
    if lmax != mmax:
        r = _degree_factors(lmax, lambda l: math.sqrt((2 * l + 1) / (2 * mmax + 1)) if l > mmax else 1.0)   # so3.py:395-403,414-422
        T, F = T * r, F * r
    keep = [l * l + l + m for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]
    return T[:, keep].reshape(nb, -1, len(keep)), F[:, keep].reshape(nb, -1, len(keep))


class SO3_Grid(nn.Module):
    def __init__(self, lmax: int, mmax: int, normalization: str = "integral", resolution: Optional[int] = None):
        super().__init__()
        if resolution is not None:
            raise NotImplementedError("EquiformerV2 on MI355X: grid_resolution is not built (the yaml leaves it unset)")
        self.lmax, self.mmax = lmax, mmax
        self.mapping = CoefficientMappingModule([lmax], [lmax])
        T, F = so3_grid_mats(lmax, mmax, normalization)
        self.register_buffer("to_grid_mat", torch.tensor(T, dtype=torch.float32))
        self.register_buffer("from_grid_mat", torch.tensor(F, dtype=torch.float32))


class ModuleListInfo(nn.ModuleList):
    def __init__(self, info_str, modules=None):
        super().__init__(modules)
        self.info_str = str(info_str)

    def __repr__(self):
        return self.info_str


class SO3_LinearV2(nn.Module):
    """so3.py:587-625: one [out, in] matrix per degree (bias on the scalars) -> nq_sph_linear_*."""

    def __init__(self, in_features: int, out_features: int, lmax: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features, self.lmax = in_features, out_features, lmax
        self.weight = nn.Parameter(torch.randn(lmax + 1, out_features, in_features))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.weight, -bound, bound)
        self.bias = nn.Parameter(torch.zeros(out_features))
        self.register_buffer("expand_index", torch.tensor([l for l in range(lmax + 1) for _ in range(2 * l + 1)], dtype=torch.long))

    def forward(self, x):
        """x [N, (lmax+1)^2 * in] -> [N, (lmax+1)^2 * out]."""
        N = x.shape[0]
        ncomp = (self.lmax + 1) ** 2
        y = _SphLinearFn.apply(x.view(N, ncomp, self.in_features), self.bias, self.weight)     # stacked [lmax + 1, out, in]: its gradient comes back as one tensor
        return y.view(N, ncomp * self.out_features)


class RadialFunction(nn.Module):
    """radial_function.py:5-28: Linear -> LayerNorm -> SiLU ... -> Linear; ``net`` keeps the reference's Sequential layout (parameter names)."""

    def __init__(self, channels_list):
        super().__init__()
        mods, n_in = [], channels_list[0]
        for i, n_out in enumerate(channels_list[1:], start=1):
            mods.append(nn.Linear(n_in, n_out, bias=True))
            n_in = n_out
            if i == len(channels_list) - 1:
                break
            mods += [nn.LayerNorm(n_out), nn.SiLU()]
        self.net = nn.Sequential(*mods)

    def hidden(self, x):
        """Everything up to (not including) the last Linear."""
        mods = list(self.net)
        for i in range(0, len(mods) - 1, 3):
            x = _linear(mods[i], x)
            x = _silu(_LayerNormFn.apply(x, mods[i + 1].weight, mods[i + 1].bias, mods[i + 1].eps))
        return x

    def last(self, h, start=0, length=None):
        """Columns [start, start + length) of the last Linear's output: a contiguous row block of its weight (no copy of the output)."""
        fc = self.net[-1]
        length = fc.out_features - start if length is None else length
        return _LinearBiasFn.apply(h, fc.weight.narrow(0, start, length), fc.bias.narrow(0, start, length), False)

    def forward(self, x):
        return self.last(self.hidden(x))


class EquivariantLayerNormArraySphericalHarmonics(nn.Module):
    """layer_norm.py:117-215."""

    def __init__(self, lmax: int, num_channels: int, eps: float = 1e-5, affine: bool = True, normalization: str = "component", std_balance_degrees: bool = True):
        super().__init__()
        if not (affine and normalization == "component" and std_balance_degrees and lmax >= 1):
            raise NotImplementedError("EquiformerV2 on MI355X: layer_norm_sh is built with affine weights, 'component' normalisation and balanced degrees")
        self.lmax, self.num_channels, self.eps = lmax, num_channels, eps
        self.norm_l0 = nn.LayerNorm(num_channels, eps=eps)
        self.affine_weight = nn.Parameter(torch.ones(lmax, num_channels))
        bw = torch.zeros((lmax + 1) ** 2 - 1, 1)
        for l in range(1, lmax + 1):
            bw[l * l - 1:l * l - 1 + 2 * l + 1] = 1.0 / (2 * l + 1)
        self.register_buffer("balance_degree_weight", bw / lmax)

    def forward(self, x):
        return _NormShFn.apply(x, self.norm_l0.weight, self.norm_l0.bias, self.affine_weight, self.balance_degree_weight, self.lmax, self.num_channels, self.eps)


def get_normalization_layer(norm_type: str, lmax: int, num_channels: int, eps: float = 1e-5, affine: bool = True, normalization: str = "component"):
    if norm_type != "layer_norm_sh":
        raise NotImplementedError(f"EquiformerV2 on MI355X: norm_type={norm_type} is not built (the yaml uses layer_norm_sh)")
    return EquivariantLayerNormArraySphericalHarmonics(lmax, num_channels, eps, affine, normalization)


# ---- SO(2) convolution --------------------------------------------------------------------------------------------------------------------------------------------
class SO2_m_Convolution(nn.Module):
    """so2_ops.py:13-62."""

    def __init__(self, m: int, sphere_channels: int, m_output_channels: int, lmax_list: List[int], mmax_list: List[int]):
        super().__init__()
        self.m = m
        n = (lmax_list[0] - m + 1) * sphere_channels
        self.fc = nn.Linear(n, 2 * m_output_channels * (lmax_list[0] - m + 1), bias=False)
        self.fc.weight.data.mul_(1 / math.sqrt(2))

    def forward(self, x_p, x_m):
        """x_p / x_m [E, n]: the +m / -m coefficients -> the two parts of the result (so2_ops.py:53-61)."""
        half = self.fc.out_features // 2
        Wr, Wi = self.fc.weight.narrow(0, 0, half), self.fc.weight.narrow(0, half, half)
        if fused_pairs_available():      # four launches, the sums in the GEMM epilogues (forward and adjoint)
            return _SO2PairFn.apply(x_p, x_m, Wr, Wi)
        out_p = lin(_DenseFn.apply(x_p, Wr, False), _DenseFn.apply(x_m, Wi, False), 1.0, -1.0)          # x_r[:, 0] - x_i[:, 1]
        out_m = lin(_DenseFn.apply(x_m, Wr, False), _DenseFn.apply(x_p, Wi, False), 1.0, 1.0)           # x_r[:, 1] + x_i[:, 0]
        return out_p, out_m


class SO2_Convolution(nn.Module):
    """so2_ops.py:65-194 on the per-m blocks [m = 0 | +1 | -1 | +2 | -2 ...] (each [E, n_m * channels], contiguous)."""

    def __init__(self, sphere_channels: int, m_output_channels: int, lmax_list: List[int], mmax_list: List[int], mappingReduced, internal_weights: bool = True,
                 edge_channels_list: Optional[List[int]] = None, extra_m0_output_channels: Optional[int] = None):
        super().__init__()
        self.sphere_channels, self.m_output_channels = sphere_channels, m_output_channels
        self.lmax_list, self.mmax_list = lmax_list, mmax_list
        self.mappingReduced = mappingReduced
        self.extra_m0_output_channels = extra_m0_output_channels
        lmax, mmax = lmax_list[0], mmax_list[0]
        n0 = (lmax + 1) * sphere_channels
        self.fc_m0 = nn.Linear(n0, m_output_channels * (lmax + 1) + (extra_m0_output_channels or 0))
        self.so2_m_conv = nn.ModuleList([SO2_m_Convolution(m, sphere_channels, m_output_channels, lmax_list, mmax_list) for m in range(1, mmax + 1)])
        self.rad_func = None
        if not internal_weights:
            self.rad_func = RadialFunction(list(edge_channels_list) + [n0 + sum(c.fc.in_features for c in self.so2_m_conv)])

    def forward(self, blocks, x_edge):
        h = self.rad_func.hidden(x_edge) if self.rad_func is not None else None
        x0, off = blocks[0], 0
        if h is not None:
            x0 = _MulFn.apply(x0, self.rad_func.last(h, 0, self.fc_m0.in_features))
            off = self.fc_m0.in_features
        y0 = _linear(self.fc_m0, x0)
        extra = None
        if self.extra_m0_output_channels is not None:
            extra = y0[:, :self.extra_m0_output_channels].contiguous()
            y0 = y0[:, self.extra_m0_output_channels:].contiguous()
        out = [y0]
        for k, conv in enumerate(self.so2_m_conv):
            xp, xm = blocks[2 * k + 1], blocks[2 * k + 2]
            if h is not None:
                r = self.rad_func.last(h, off, conv.fc.in_features)
                xp, xm = _MulFn.apply(xp, r), _MulFn.apply(xm, r)
                off += conv.fc.in_features
            out += list(conv(xp, xm))
        return (out, extra) if self.extra_m0_output_channels is not None else out


# ---- blocks -------------------------------------------------------------------------------------------------------------------------------------------------------
def _edge_scalars(mod, G):
    """cat(distance expansion, source embedding, target embedding) (transformer_block.py:201-208)."""
    s = _EmbeddingFn.apply(mod.source_embedding.weight, G.z_src, [G.src_inverse + (G.N,), G.z_inverse])
    t = _EmbeddingFn.apply(mod.target_embedding.weight, G.z_dst, [G.dst_inverse + (G.N,), G.z_inverse])
    return torch.cat([G.x_dist, s, t], dim=1)


def _atom_edge_embeddings(mod, max_num_elements, channels):
    mod.source_embedding = nn.Embedding(max_num_elements, channels)
    mod.target_embedding = nn.Embedding(max_num_elements, channels)
    nn.init.uniform_(mod.source_embedding.weight.data, -0.001, 0.001)
    nn.init.uniform_(mod.target_embedding.weight.data, -0.001, 0.001)


class SO2EquivariantGraphAttention(nn.Module):
    """transformer_block.py:22-384."""

    def __init__(self, sphere_channels: int, hidden_channels: int, num_heads: int, attn_alpha_channels: int, attn_value_channels: int, output_channels: int,
                 lmax_list: List[int], mmax_list: List[int], SO3_rotation, mappingReduced, SO3_grid, max_num_elements: int, edge_channels_list,
                 use_atom_edge_embedding: bool = True, use_m_share_rad: bool = False, activation="scaled_silu", use_s2_act_attn: bool = False,
                 use_attn_renorm: bool = True, use_gate_act: bool = False, use_sep_s2_act: bool = True, alpha_drop: float = 0.0):
        super().__init__()
        for ok, what in ((use_atom_edge_embedding, "use_atom_edge_embedding=False"), (not use_m_share_rad, "use_m_share_rad"), (not use_s2_act_attn, "use_s2_act_attn"),
                         (use_attn_renorm, "use_attn_renorm=False"), (not use_gate_act, "use_gate_act"), (use_sep_s2_act, "use_sep_s2_act=False")):
            if not ok:
                raise NotImplementedError(f"EquiformerV2 on MI355X: {what} is not built (config/model/equiformer_v2_oc20.yaml is the supported configuration)")
        self.sphere_channels, self.hidden_channels, self.num_heads = sphere_channels, hidden_channels, num_heads
        self.attn_alpha_channels, self.attn_value_channels, self.output_channels = attn_alpha_channels, attn_value_channels, output_channels
        self.lmax_list, self.mmax_list = lmax_list, mmax_list
        self.SO3_rotation, self.mappingReduced, self.SO3_grid = SO3_rotation, mappingReduced, SO3_grid
        self.max_num_elements = max_num_elements
        ecl = list(edge_channels_list)
        _atom_edge_embeddings(self, max_num_elements, ecl[-1])
        ecl[0] = ecl[0] + 2 * ecl[-1]
        self.edge_channels_list = ecl
        extra = num_heads * attn_alpha_channels + hidden_channels
        self.so2_conv_1 = SO2_Convolution(2 * sphere_channels, hidden_channels, lmax_list, mmax_list, mappingReduced, internal_weights=False,
                                          edge_channels_list=ecl, extra_m0_output_channels=extra)
        self.alpha_norm = nn.LayerNorm(attn_alpha_channels)
        self.alpha_dot = nn.Parameter(torch.randn(num_heads, attn_alpha_channels))
        std = 1.0 / math.sqrt(attn_alpha_channels)
        nn.init.uniform_(self.alpha_dot, -std, std)
        self.alpha_drop = alpha_drop
        self.so2_conv_2 = SO2_Convolution(hidden_channels, num_heads * attn_value_channels, lmax_list, mmax_list, mappingReduced, internal_weights=True,
                                          edge_channels_list=None, extra_m0_output_channels=None)
        self.proj = SO3_LinearV2(num_heads * attn_value_channels, output_channels, lmax=lmax_list[0])

    def forward(self, x, G, K):
        """x [N, n_full * C] -> [N, n_full * output_channels]."""
        Cc, Hc, H, A, V = self.sphere_channels, self.hidden_channels, self.num_heads, self.attn_alpha_channels, self.attn_value_channels
        E, o = G.E, K.order
        rows = K.block_rows
        x_edge = _edge_scalars(self, G)
        # source and target embeddings rotated into the edge frame, side by side along the channel (transformer_block.py:210-236): blocks [E, rows, 2C]
        msg = list(_RotateFn.apply(x, G, K, Cc, [(G.src, G.src_inverse), (G.dst, G.dst_inverse)]))
        msg, extra = self.so2_conv_1(msg, x_edge)
        # attention weights (transformer_block.py:343-356)
        xa = extra[:, :H * A].contiguous().view(E * H, A)
        xa = _LayerNormFn.apply(xa, self.alpha_norm.weight, self.alpha_norm.bias, self.alpha_norm.eps).view(E, H * A)
        alpha = _SoftmaxFn.apply(_LogitsFn.apply(xa, self.alpha_dot), G.ptr, G.N)
        if self.training and self.alpha_drop > 0.0:
            keep = (torch.rand(E, H, device=x.device) >= self.alpha_drop).to(torch.float32) / (1.0 - self.alpha_drop)
            alpha = _MulFn.apply(alpha, keep)
        # separable S2 activation (activation.py:155-176): SiLU on the (lmax, mmax) grid, scalars replaced by SiLU(gating scalars)
        gating = _silu(extra[:, H * A:].contiguous())
        ng = K.to_grid_red.shape[0]
        if s2_activation_fusable(K.to_grid_red, rows, Hc):
            msg = list(_S2ActBlocksFn.apply(K.to_grid_red, K.from_grid_red, rows, E, Hc, *msg))      # to_grid -> SiLU -> from_grid in one kernel, grid in registers
        else:
            grid = _BlocksInFn.apply(K.to_grid_red, 0, ng, o.n_red, Hc, False, 1, rows, E, *msg)
            msg = list(_BlocksOutFn.apply(_silu(grid), K.from_grid_red, 0, ng, o.n_red, Hc, True, 1, rows, None, None, E))
        msg[0] = torch.cat([gating, msg[0][:, Hc:]], dim=1)
        msg = self.so2_conv_2(msg, None)
        msg = _HeadScaleFn.apply(alpha, rows, H, V, *msg)
        y = _RotateBackFn.apply(G, K, H * V, K.coef_scale, rows, *msg)       # rotate back with rotate_inv's rescale (so3.py:121-136) + _reduce_edge, one pass
        return self.proj(y)


class FeedForwardNetwork(nn.Module):
    """transformer_block.py:387-507 (grid MLP with separable scalars)."""

    def __init__(self, sphere_channels: int, hidden_channels: int, output_channels: int, lmax_list: List[int], mmax_list: List[int], SO3_grid,
                 activation: str = "scaled_silu", use_gate_act: bool = False, use_grid_mlp: bool = False, use_sep_s2_act: bool = True):
        super().__init__()
        if use_gate_act or not use_grid_mlp or not use_sep_s2_act:
            raise NotImplementedError("EquiformerV2 on MI355X: the feed-forward network is built with use_grid_mlp, use_sep_s2_act and without gate activation")
        self.sphere_channels, self.hidden_channels, self.output_channels = sphere_channels, hidden_channels, output_channels
        self.lmax_list, self.mmax_list, self.SO3_grid = lmax_list, mmax_list, SO3_grid
        self.max_lmax = max(lmax_list)
        self.so3_linear_1 = SO3_LinearV2(sphere_channels, hidden_channels, lmax=self.max_lmax)
        self.scalar_mlp = nn.Sequential(nn.Linear(sphere_channels, hidden_channels, bias=True), nn.SiLU())
        self.grid_mlp = nn.Sequential(nn.Linear(hidden_channels, hidden_channels, bias=False), nn.SiLU(), nn.Linear(hidden_channels, hidden_channels, bias=False),
                                      nn.SiLU(), nn.Linear(hidden_channels, hidden_channels, bias=False))
        self.so3_linear_2 = SO3_LinearV2(hidden_channels, output_channels, lmax=self.max_lmax)

    def forward(self, x, G, K):
        N, nf, Cc, Hf = x.shape[0], K.order.n_full, self.sphere_channels, self.hidden_channels
        gating = _linear(self.scalar_mlp[0], x[:, :Cc].contiguous(), act=True)
        h = self.so3_linear_1(x)
        T, F = K.to_grid_full, K.from_grid_full
        ng = T.shape[0]
        g = _RowFn.apply(h, T, 0, ng, nf, Hf, False, None, None, N).view(N * ng, Hf)
        g = _linear(self.grid_mlp[4], _linear(self.grid_mlp[2], _linear(self.grid_mlp[0], g, act=True), act=True))
        h = _RowFn.apply(g.view(N, ng * Hf), F, 0, ng, nf, Hf, True, None, None, N)
        h = torch.cat([gating, h[:, Hf:]], dim=1)
        return self.so3_linear_2(h)


class GraphDropPath(nn.Module):
    """drop.py:57-71."""

    def __init__(self, drop_prob: float):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x, G, I, Cc):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        d = torch.floor(keep + torch.rand(G.B, device=x.device)) / keep
        return _ScaleFn.apply(x, d.contiguous(), G.atom_mol, None, I, Cc)

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


class TransBlockV2(nn.Module):
    """transformer_block.py:510-630."""

    def __init__(self, sphere_channels: int, attn_hidden_channels: int, num_heads: int, attn_alpha_channels: int, attn_value_channels: int,
                 ffn_hidden_channels: int, output_channels: int, lmax_list: List[int], mmax_list: List[int], SO3_rotation, mappingReduced, SO3_grid,
                 max_num_elements: int, edge_channels_list: List[int], use_atom_edge_embedding: bool = True, use_m_share_rad: bool = False,
                 attn_activation: str = "silu", use_s2_act_attn: bool = False, use_attn_renorm: bool = True, ffn_activation: str = "silu",
                 use_gate_act: bool = False, use_grid_mlp: bool = False, use_sep_s2_act: bool = True, norm_type: str = "rms_norm_sh", alpha_drop: float = 0.0,
                 drop_path_rate: float = 0.0, proj_drop: float = 0.0) -> None:
        super().__init__()
        if proj_drop > 0.0 or sphere_channels != output_channels:
            raise NotImplementedError("EquiformerV2 on MI355X: proj_drop and a channel-changing block are not built")
        max_lmax = max(lmax_list)
        self.norm_1 = get_normalization_layer(norm_type, lmax=max_lmax, num_channels=sphere_channels)
        self.ga = SO2EquivariantGraphAttention(sphere_channels, attn_hidden_channels, num_heads, attn_alpha_channels, attn_value_channels, sphere_channels,
                                               lmax_list, mmax_list, SO3_rotation, mappingReduced, SO3_grid, max_num_elements, edge_channels_list,
                                               use_atom_edge_embedding, use_m_share_rad, attn_activation, use_s2_act_attn, use_attn_renorm, use_gate_act,
                                               use_sep_s2_act, alpha_drop)
        self.drop_path = GraphDropPath(drop_path_rate) if drop_path_rate > 0.0 else None
        self.proj_drop = None
        self.norm_2 = get_normalization_layer(norm_type, lmax=max_lmax, num_channels=sphere_channels)
        self.ffn = FeedForwardNetwork(sphere_channels, ffn_hidden_channels, output_channels, lmax_list, mmax_list, SO3_grid, ffn_activation, use_gate_act,
                                      use_grid_mlp, use_sep_s2_act)
        self.ffn_shortcut = None

    def forward(self, x, G, K, record=None):
        nf, Cc = K.order.n_full, K.C
        h = self.norm_1(x)
        if record is not None:
            record["norm1"] = h
        h = self.ga(h, G, K)
        if record is not None:
            record["ga"] = h
        if self.drop_path is not None:
            h = self.drop_path(h, G, nf, Cc)
        x = lin(h, x)
        h = self.ffn(self.norm_2(x), G, K)
        if self.drop_path is not None:
            h = self.drop_path(h, G, nf, Cc)
        return lin(h, x)


class EdgeDegreeEmbedding(nn.Module):
    """input_block.py:10-117: m = 0 coefficients from the edge scalars, rotated back to the global frame and summed per target atom."""

    def __init__(self, sphere_channels: int, lmax_list: List[int], mmax_list: List[int], SO3_rotation, mappingReduced, max_num_elements: int, edge_channels_list,
                 use_atom_edge_embedding: bool, rescale_factor):
        super().__init__()
        if not use_atom_edge_embedding:
            raise NotImplementedError("EquiformerV2 on MI355X: use_atom_edge_embedding=False is not built")
        self.sphere_channels, self.lmax_list, self.mmax_list = sphere_channels, lmax_list, mmax_list
        self.SO3_rotation, self.mappingReduced = SO3_rotation, mappingReduced
        self.max_num_elements = max_num_elements
        ecl = list(edge_channels_list)
        _atom_edge_embeddings(self, max_num_elements, ecl[-1])
        ecl[0] = ecl[0] + 2 * ecl[-1]
        ecl.append((lmax_list[0] + 1) * sphere_channels)
        self.edge_channels_list = ecl
        self.rad_func = RadialFunction(ecl)
        self.rescale_factor = rescale_factor

    def forward(self, G, K):
        o, Cc = K.order, self.sphere_channels
        h = self.rad_func(_edge_scalars(self, G))                                                       # [E, (lmax + 1) * C]: the m = 0 block
        # the m = 0 rows lead every edge's Wigner block: rotate back + sum over in-edges, scaled by rotate_inv's rescale / rescale_factor
        return _RotateBackFn.apply(G, K, Cc, K.coef_scale_degree, [o.m_size[0]], h)


class _Consts:
    pass


def _drop_device_constants(module, incompatible_keys):
    """load_state_dict post-hook: the kernels' constant tables are rebuilt from the (possibly replaced) SO3_grid buffers on the next forward."""
    module._dev_const = None


class EquiformerV2_OC20(nn.Module):
    """equiformer_v2/equiformer_v2_oc20.py:51-640 (constructor arguments :121-160)."""

    def __init__(self, use_pbc: bool = True, regress_forces: bool = True, otf_graph: bool = True, max_neighbors: int = 500, max_radius: float = 5.0,
                 max_num_elements: int = 90, num_layers: int = 12, sphere_channels: int = 128, attn_hidden_channels: int = 128, num_heads: int = 8,
                 attn_alpha_channels: int = 32, attn_value_channels: int = 16, ffn_hidden_channels: int = 512, norm_type: str = "rms_norm_sh",
                 lmax_list: List[int] = [6], mmax_list: List[int] = [2], grid_resolution: Optional[int] = None, num_sphere_samples: int = 128,
                 edge_channels: int = 128, use_atom_edge_embedding: bool = True, share_atom_edge_embedding: bool = False, use_m_share_rad: bool = False,
                 distance_function: str = "gaussian", num_distance_basis: int = 512, attn_activation: str = "scaled_silu", use_s2_act_attn: bool = False,
                 use_attn_renorm: bool = True, ffn_activation: str = "scaled_silu", use_gate_act: bool = False, use_grid_mlp: bool = False,
                 use_sep_s2_act: bool = True, alpha_drop: float = 0.1, drop_path_rate: float = 0.05, proj_drop: float = 0.0, weight_init: str = "normal",
                 enforce_max_neighbors_strictly: bool = True, avg_num_nodes: Optional[float] = None, avg_degree: Optional[float] = None,
                 use_energy_lin_ref: Optional[bool] = False, load_energy_lin_ref: Optional[bool] = False):
        super().__init__()
        for ok, what in ((not use_pbc, "periodic boundary conditions"), (regress_forces, "regress_forces=False"), (len(lmax_list) == 1 and len(mmax_list) == 1,
                         "more than one resolution"), (distance_function == "gaussian", f"distance_function={distance_function}"), (1 <= lmax_list[0] <= 6, "lmax > 6"),
                         (grid_resolution is None, "grid_resolution"), (not share_atom_edge_embedding, "share_atom_edge_embedding"),
                         (not (use_energy_lin_ref or load_energy_lin_ref), "energy linear references"), (weight_init in ("normal", "uniform"), f"weight_init={weight_init}")):
            if not ok:
                raise NotImplementedError(f"EquiformerV2 on MI355X: {what} is not built (config/model/equiformer_v2_oc20.yaml is the supported configuration)")
        self.use_pbc, self.regress_forces, self.otf_graph = use_pbc, regress_forces, otf_graph
        self.max_neighbors, self.max_radius, self.cutoff, self.max_num_elements = max_neighbors, max_radius, max_radius, max_num_elements
        self.num_layers, self.sphere_channels, self.attn_hidden_channels, self.num_heads = num_layers, sphere_channels, attn_hidden_channels, num_heads
        self.attn_alpha_channels, self.attn_value_channels, self.ffn_hidden_channels = attn_alpha_channels, attn_value_channels, ffn_hidden_channels
        self.norm_type, self.lmax_list, self.mmax_list = norm_type, list(lmax_list), list(mmax_list)
        self.num_sphere_samples, self.edge_channels = num_sphere_samples, edge_channels
        self.alpha_drop, self.drop_path_rate, self.proj_drop, self.weight_init = alpha_drop, drop_path_rate, proj_drop, weight_init
        self.avg_num_nodes, self.avg_degree = avg_num_nodes or _AVG_NUM_NODES, avg_degree or _AVG_DEGREE
        self.sphere_channels_all = sphere_channels
        lmax, mmax = lmax_list[0], mmax_list[0]
        self.sphere_embedding = nn.Embedding(max_num_elements, sphere_channels)
        self.distance_expansion = GaussianSmearing(0.0, self.cutoff, 600, 2.0)
        self.edge_channels_list = [int(self.distance_expansion.num_output)] + [edge_channels] * 2
        self.SO3_rotation = nn.ModuleList([SO3_Rotation(lmax)])
        self.mappingReduced = CoefficientMappingModule(self.lmax_list, self.mmax_list)
        self.SO3_grid = ModuleListInfo(f"({lmax}, {lmax})")
        for lval in range(lmax + 1):
            self.SO3_grid.append(nn.ModuleList([SO3_Grid(lval, m, normalization="component") for m in range(lmax + 1)]))
        self.edge_degree_embedding = EdgeDegreeEmbedding(sphere_channels, self.lmax_list, self.mmax_list, self.SO3_rotation, self.mappingReduced, max_num_elements,
                                                         self.edge_channels_list, use_atom_edge_embedding, rescale_factor=self.avg_degree)
        self.blocks = nn.ModuleList([
            TransBlockV2(sphere_channels, attn_hidden_channels, num_heads, attn_alpha_channels, attn_value_channels, ffn_hidden_channels, sphere_channels,
                         self.lmax_list, self.mmax_list, self.SO3_rotation, self.mappingReduced, self.SO3_grid, max_num_elements, self.edge_channels_list,
                         use_atom_edge_embedding, use_m_share_rad, attn_activation, use_s2_act_attn, use_attn_renorm, ffn_activation, use_gate_act, use_grid_mlp,
                         use_sep_s2_act, norm_type, alpha_drop, drop_path_rate, proj_drop) for _ in range(num_layers)])
        self.norm = get_normalization_layer(norm_type, lmax=lmax, num_channels=sphere_channels)
        self.energy_block = FeedForwardNetwork(sphere_channels, ffn_hidden_channels, 1, self.lmax_list, self.mmax_list, self.SO3_grid, ffn_activation, use_gate_act,
                                               use_grid_mlp, use_sep_s2_act)
        self.force_block = SO2EquivariantGraphAttention(sphere_channels, attn_hidden_channels, num_heads, attn_alpha_channels, attn_value_channels, 1, self.lmax_list,
                                                        self.mmax_list, self.SO3_rotation, self.mappingReduced, self.SO3_grid, max_num_elements,
                                                        self.edge_channels_list, use_atom_edge_embedding, use_m_share_rad, attn_activation, use_s2_act_attn,
                                                        use_attn_renorm, use_gate_act, use_sep_s2_act, alpha_drop=0.0)
        self.apply(self._init_weights)
        self.apply(self._uniform_init_rad_func_linear_weights)
        # constants of the kernels (not parameters): Wigner J matrices, coefficient orders, the grid matrices with their columns in m-primary order
        o = CoefficientOrder(lmax, mmax)
        J = j_matrices(lmax)
        Tr, Fr = (a.reshape(-1, a.shape[-1]) for a in so3_grid_mats(lmax, mmax))                    # columns: kept coefficients, l-primary
        Tf, Ff = (a.reshape(-1, a.shape[-1]) for a in so3_grid_mats(lmax, lmax))
        pos = {full: k for k, full in enumerate(o.red_l_primary)}
        perm = [pos[full] for full in o.red_m_primary]                                                # l-primary position of every m-primary coefficient
        scale = _degree_factors(lmax, lambda l: math.sqrt((2 * l + 1) / (2 * mmax + 1)) if l > mmax else 1.0)
        f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32)                        # noqa: E731
        i32 = lambda a: torch.tensor(a, dtype=torch.int32)                                                # noqa: E731
        self._const = dict(J=f32(np.concatenate([j.reshape(-1) for j in J])), J_off=i32(np.cumsum([0] + [j.size for j in J])[:-1]),
                           red_l=i32([l for l, _ in o.m_primary]), red_row=i32([l + m for l, m in o.m_primary]),
                           to_grid_red=f32(Tr[:, perm]), from_grid_red=f32(Fr[:, perm]), to_grid_full=f32(Tf), from_grid_full=f32(Ff),
                           coef_scale=f32(scale), coef_scale_degree=f32(scale / self.avg_degree))
        self._order = o
        self._grid_perm = perm
        self._dev_const = None
        self.register_load_state_dict_post_hook(_drop_device_constants)   # module-level function: a lambda here would make the model unpicklable

    # ---- reference initialisation (equiformer_v2_oc20.py:588-612) ----
    def _init_weights(self, m):
        if isinstance(m, (nn.Linear, SO3_LinearV2)):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
            if self.weight_init == "normal":
                nn.init.normal_(m.weight, 0, 1 / math.sqrt(m.in_features))
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _uniform_init_rad_func_linear_weights(self, m):
        if isinstance(m, RadialFunction):
            m.apply(self._uniform_init_linear_weights)

    def _uniform_init_linear_weights(self, m):
        if isinstance(m, nn.Linear):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
            std = 1 / math.sqrt(m.in_features)
            nn.init.uniform_(m.weight, -std, std)

    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def no_weight_decay(self):
        """equiformer_v2_oc20.py:614-640: biases, layer-norm parameters and the embeddings' scalars stay out of weight decay."""
        out = set()
        for mname, m in self.named_modules():
            if isinstance(m, (nn.Linear, SO3_LinearV2, nn.LayerNorm, EquivariantLayerNormArraySphericalHarmonics)):
                for pname, _ in m.named_parameters(recurse=False):
                    if isinstance(m, (nn.Linear, SO3_LinearV2)) and "weight" in pname:
                        continue
                    out.add(f"{mname}.{pname}")
        return out

    def _constants(self, dev):
        """The kernel constants on ``dev``.  The S2-grid matrices are taken from the model's own ``SO3_grid`` buffers (reordered to the kernels' m-primary columns):
        freshly constructed they are this package's matrices, after ``load_state_dict`` of a reference checkpoint they are the matrices e3nn gave the reference, so a
        loaded model evaluates the grids the checkpoint was trained with (the one place where the restated e3nn arithmetic could differ in the last digits)."""
        fresh = self._dev_const is None or self._dev_const.device != dev
        K = eSCN._constants(self, dev)
        if fresh:
            lmax, mmax = self.lmax_list[0], self.mmax_list[0]
            red, full = self.SO3_grid[lmax][mmax], self.SO3_grid[lmax][lmax]
            perm = torch.tensor(self._grid_perm, dtype=torch.long)
            flat = lambda m: m.detach().to(device=dev, dtype=torch.float32).reshape(-1, m.shape[-1])                    # noqa: E731
            K.to_grid_red = flat(red.to_grid_mat)[:, perm.to(dev)].contiguous()
            K.from_grid_red = flat(red.from_grid_mat)[:, perm.to(dev)].contiguous()
            K.to_grid_full, K.from_grid_full = flat(full.to_grid_mat).contiguous(), flat(full.from_grid_mat).contiguous()
        return K

    prepare = eSCN.prepare                              # data.prepared = net.prepare(data): forward without host synchronisation (trainer.GraphedStep)
    build_graph = eSCN.build_graph                      # radius graph + frames + Wigner rows + the inverse lists of the gathers (escn.py / equiformer: same stage)

    def forward(self, data, edge_rot_mat=None, return_intermediates: bool = False):
        _gemnet.weights_epoch_advance()   # bf16 weight copies are re-packed once per forward (parameters may have been updated in place)
        if not data.pos.is_cuda:
            raise RuntimeError("nabladft_amd.EquiformerV2_OC20 runs on MI355X only: tensors must be on a cuda (HIP) device")
        G = getattr(data, "prepared", None)            # build_graph(data) done ahead: the forward then issues no host synchronisation (HIP-graph capture)
        if G is None:
            G = self.build_graph(data, edge_rot_mat)
        elif G.N != int(data.pos.shape[0]):
            raise ValueError("data.prepared belongs to another batch")
        else:
            _lib.check_prepared(G, data)
        K = self._constants(data.pos.device)
        Cc, nf = self.sphere_channels, K.order.n_full
        emb = _EmbeddingFn.apply(self.sphere_embedding.weight, G.z, [G.z_inverse])                     # the l = 0 coefficient (equiformer_v2_oc20.py:517-530)
        x = torch.cat([emb, emb.new_zeros(G.N, (nf - 1) * Cc)], dim=1)
        x = lin(x, self.edge_degree_embedding(G, K))
        rec = {"embed": x} if return_intermediates else None
        for i, blk in enumerate(self.blocks):
            x = blk(x, G, K, rec if i == 0 else None)
            if rec is not None:
                rec[f"block{i}"] = x
        x = self.norm(x)
        node_energy = self.energy_block(x, G, K)[:, :1].contiguous()                                   # [N, 1]: the scalar coefficient of the single output channel
        energy = _SegSumFn.apply(node_energy, G.mol_ptr, G.atom_mol, G.B).squeeze(1) / self.avg_num_nodes
        forces = self.force_block(x, G, K)[:, 1:4].contiguous()                                        # the l = 1 coefficients (equiformer_v2_oc20.py:573-575)
        if return_intermediates:
            return energy, forces, rec, G
        return energy, forces
