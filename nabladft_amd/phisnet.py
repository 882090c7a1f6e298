"""PhiSNet building blocks on MI355X: mirrors of
  Swish, ShiftedSoftplus       phisnet/nn/modules/swish.py:10-24, shifted_softplus.py:14-32
  SphericalLinear              spherical_linear.py:10-59
  ResidualBlock / ResidualStack  residual_block.py:12-64, residual_stack.py:10-43
  InteractionBlock             interaction_block.py:13-150
  ModularBlock                 modular_block.py:11-80
with the reference constructors, attribute / parameter names (state_dict compatible) and list-of-orders tensors
``xs[l]: [1, N, 2l+1, F]``.  The arithmetic runs in HIP kernels: Clebsch-Gordan mixing (csrc/so3.hip), fp32 MFMA GEMMs for every
Linear (csrc/gemm.hip), feature-wise activations, the neighbour gather and the fixed-order segment sum over pairs (csrc/geobasis.hip:
deterministic, unlike ``index_add`` on a GPU).  torch is used for autograd plumbing and residual adds.  GPU only.
``idx_i`` may be a ``PairIndex`` (built once per batch) instead of a tensor.
"""
from typing import List

import os

import torch
from torch import nn

from . import _lib
import ctypes as C

from .so3 import PackedList, PairMixing, SelfMixing, _LinearFn, _pack, _require_gpu, _unpack


class PairIndex:
    """Index tables of one batch of pairs for the deterministic gather / segment-sum kernels: ``idx_i`` must be sorted (PhiSNet's fill_idx and
    QHNet's full graph are); the reverse of the neighbour gather needs the pairs grouped by ``idx_j`` (a stable argsort, once per batch)."""

    def __init__(self, idx_i: torch.Tensor, idx_j: torch.Tensor, num_atoms: int):
        self.idx_i, self.idx_j, self.N = idx_i.long().contiguous(), idx_j.long().contiguous(), int(num_atoms)
        if self.idx_i.numel() > 1 and bool((self.idx_i[1:] < self.idx_i[:-1]).any()):
            raise ValueError("pairs must be sorted by the centre atom idx_i")
        cnt_i = torch.bincount(self.idx_i, minlength=self.N)
        self.ptr_i = torch.cat([cnt_i.new_zeros(1), cnt_i.cumsum(0)]).contiguous()
        self.order_j = torch.argsort(self.idx_j, stable=True).contiguous()
        cnt_j = torch.bincount(self.idx_j, minlength=self.N)
        self.ptr_j = torch.cat([cnt_j.new_zeros(1), cnt_j.cumsum(0)]).contiguous()


class _GatherFn(torch.autograd.Function):
    """x [1, N, m, F] -> [1, P, m, F] rows of idx_j; reverse = segment sum over the pairs of each j."""

    @staticmethod
    def forward(ctx, x, pidx):
        lib = _lib.load()
        x2 = x.to(torch.float32).contiguous()
        Cw = x2.shape[-2] * x2.shape[-1]
        P = pidx.idx_j.numel()
        out = torch.empty(1, P, x2.shape[-2], x2.shape[-1], device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_gather_rows(_lib.ptr(x2), _lib.ptr(pidx.idx_j), P, Cw, _lib.ptr(out), _lib.stream_ptr()))
        ctx.pidx, ctx.shape = pidx, x2.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pidx = ctx.pidx
        g = g.to(torch.float32).contiguous()
        Cw = ctx.shape[-2] * ctx.shape[-1]
        out = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        _lib.check(lib.nq_segment_sum(_lib.ptr(g), _lib.ptr(pidx.order_j), _lib.ptr(pidx.ptr_j), None, pidx.N, Cw, _lib.ptr(out), _lib.stream_ptr()))
        return out, None


class _SegmentAddFn(torch.autograd.Function):
    """base [1, N, m, F] + sum over the pair rows of every centre atom (index_add over sorted idx_i); reverse = identity + gather by idx_i."""

    @staticmethod
    def forward(ctx, base, rows, pidx):
        lib = _lib.load()
        b2, r2 = base.to(torch.float32).contiguous(), rows.to(torch.float32).contiguous()
        Cw = b2.shape[-2] * b2.shape[-1]
        out = torch.empty_like(b2)
        _lib.check(lib.nq_segment_sum(_lib.ptr(r2), None, _lib.ptr(pidx.ptr_i), _lib.ptr(b2), pidx.N, Cw, _lib.ptr(out), _lib.stream_ptr()))
        ctx.pidx = pidx
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pidx = ctx.pidx
        g = g.to(torch.float32).contiguous()
        Cw = g.shape[-2] * g.shape[-1]
        P = pidx.idx_i.numel()
        grows = torch.empty(1, P, g.shape[-2], g.shape[-1], device=g.device, dtype=torch.float32)
        _lib.check(lib.nq_gather_rows(_lib.ptr(g), _lib.ptr(pidx.idx_i), P, Cw, _lib.ptr(grows), _lib.stream_ptr()))
        return g, grows, None


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha, beta, kind):
        lib = _lib.load()
        F = alpha.shape[0]
        x2 = x.to(torch.float32).contiguous()
        a, b = alpha.detach().to(torch.float32).contiguous(), beta.detach().to(torch.float32).contiguous()
        y = torch.empty_like(x2)
        _lib.check(lib.nq_feature_act(_lib.ptr(x2), _lib.ptr(a), _lib.ptr(b), x2.numel() // F, F, kind, _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x2, a, b)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x2, a, b = ctx.saved_tensors
        F = a.shape[0]
        g = g.to(torch.float32).contiguous()
        gx, ga, gb = torch.empty_like(x2), torch.empty_like(x2), torch.empty_like(x2)
        _lib.check(lib.nq_feature_act_backward(_lib.ptr(x2), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g), x2.numel() // F, F, ctx.kind, _lib.ptr(gx), _lib.ptr(ga),
                                               _lib.ptr(gb), _lib.stream_ptr()))
        return gx, ga.view(-1, F).sum(0), gb.view(-1, F).sum(0), None


class _PackedAct0Fn(torch.autograd.Function):
    """Activation of the scalar component of a packed irreps tensor [rows, ncomp, F], copy of the rest (one launch)."""

    @staticmethod
    def forward(ctx, x, alpha, beta, kind):
        lib = _lib.load()
        x2 = x.to(torch.float32).contiguous()
        rows, ncomp, F = x2.shape
        a, b = alpha.detach().to(torch.float32).contiguous(), beta.detach().to(torch.float32).contiguous()
        y = torch.empty_like(x2)
        _lib.check(lib.nq_packed_act0(_lib.ptr(x2), _lib.ptr(a), _lib.ptr(b), rows, ncomp, F, kind, _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x2, a, b)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x2, a, b = ctx.saved_tensors
        rows, ncomp, F = x2.shape
        g = g.to(torch.float32).contiguous()
        gx = torch.empty_like(x2)
        ga, gb = torch.empty(rows, F, device=g.device, dtype=torch.float32), torch.empty(rows, F, device=g.device, dtype=torch.float32)
        _lib.check(lib.nq_packed_act0_backward(_lib.ptr(x2), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g), rows, ncomp, F, ctx.kind, _lib.ptr(gx), _lib.ptr(ga),
                                               _lib.ptr(gb), _lib.stream_ptr()))
        return gx, ga.sum(0), gb.sum(0), None


def _act0(xs, act):
    """``ys = list(xs); ys[0] = act(ys[0])`` of the reference blocks; on a packed list it is one kernel and stays packed."""
    if isinstance(xs, PackedList) and xs.packed is not None and xs.packed.is_cuda:
        return _unpack(_PackedAct0Fn.apply(xs.packed, act.alpha, act.beta, act._kind), xs.order, xs.lead, xs.F)
    ys = list(xs)
    ys[0] = act(ys[0])
    return ys


def _add_lists(xs, ys):
    px, py = isinstance(xs, PackedList) and xs.packed is not None, isinstance(ys, PackedList) and ys.packed is not None
    if px and py and xs.order == ys.order:
        return _unpack(xs.packed + ys.packed, xs.order, xs.lead, xs.F)
    if (px or py) and len(xs) == len(ys):            # one side packed: pack the other (one concatenate) instead of order-wise adds
        p, q = (xs, ys) if px else (ys, xs)
        return _unpack(p.packed + _pack(q, p.order, p.F)[0], p.order, p.lead, p.F)
    return [x + y for x, y in zip(xs, ys)]


def _gather_list(xs, pidx, order, F):
    """Neighbour gather of every order: one launch on the packed rows."""
    x, lead = _pack(xs, order, F)
    n = x.shape[0]
    g = _GatherFn.apply(x.view(1, n, x.shape[1], F), pidx)
    return _unpack(g.view(g.shape[1], x.shape[1], F), order, (*lead[:-1], g.shape[1]), F)


def _segment_add_list(base, rows, pidx, order, F):
    """base + sum over the pair rows of every centre atom, all orders in one launch."""
    b, lead = _pack(base, order, F)
    r, _ = _pack(rows, order, F)
    out = _SegmentAddFn.apply(b.view(1, *b.shape), r.view(1, *r.shape), pidx)
    return _unpack(out.view(b.shape), order, lead, F)


class _SphLinearFn(torch.autograd.Function):
    """All per-order Linear layers of a SphericalLinear on the packed tensor: one launch forward, one for the input gradient.  ``weights`` is either one
    [Fout, Fin] matrix per order, or ONE stacked tensor [order + 1, Fout, Fin] (e3nn-style ``o3.Linear`` / ``SO3_LinearV2`` parameters): the stacked form
    gets its gradient back as one tensor -- with per-order views autograd re-assembles it from order + 1 zero-filled full-size pieces."""

    @staticmethod
    def forward(ctx, x, bias, *weights):
        lib = _lib.load()
        x2 = x.to(torch.float32).contiguous()
        rows, ncomp, Fin = x2.shape
        stacked = len(weights) == 1 and weights[0].dim() == 3
        if stacked:
            Wst = weights[0].detach().to(torch.float32).contiguous()
            ws = list(Wst.unbind(0))                        # contiguous slices of one buffer
        else:
            ws = [w.detach().to(torch.float32).contiguous() for w in weights]
        order = len(ws) - 1
        Fout = ws[0].shape[0]
        y = torch.empty(rows, ncomp, Fout, device=x.device, dtype=torch.float32)
        wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        b = bias.detach().to(torch.float32).contiguous() if bias is not None else None
        _lib.check(lib.nq_sph_linear_forward(_lib.ptr(x2), wp, _lib.ptr(b), _lib.ptr(y), rows, order, Fin, Fout, _lib.stream_ptr()))
        if stacked:
            ctx.save_for_backward(x2, Wst)
        else:
            ctx.save_for_backward(x2, *ws)
        ctx.has_bias, ctx.stacked = bias is not None, stacked
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x2, *ws = ctx.saved_tensors
        if ctx.stacked:
            Wst = ws[0]
            ws = list(Wst.unbind(0))
        rows, ncomp, Fin = x2.shape
        order, Fout = len(ws) - 1, ws[0].shape[0]
        g = g.to(torch.float32).contiguous()
        wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        gx = torch.empty_like(x2)
        _lib.check(lib.nq_sph_linear_input_grad(_lib.ptr(g), wp, _lib.ptr(gx), rows, order, Fin, Fout, _lib.stream_ptr()))
        if ctx.stacked:
            gst = torch.empty_like(Wst)
            gws = list(gst.unbind(0))
        else:
            gws = [torch.empty_like(w) for w in ws]
        gwp = (C.c_void_p * len(gws))(*[w.data_ptr() for w in gws])
        gb = torch.empty(Fout, device=g.device, dtype=torch.float32) if ctx.has_bias else None
        scr = torch.empty(int(lib.nq_sph_weight_grad_scratch_floats(rows, order, Fin, Fout)) + 64, device=g.device, dtype=torch.float32)
        _lib.check(lib.nq_sph_linear_weight_grad(_lib.ptr(g), _lib.ptr(x2), gwp, _lib.ptr(gb), rows, order, Fin, Fout, _lib.ptr(scr), _lib.stream_ptr()))
        if ctx.stacked:
            return gx, gb, gst
        return (gx, gb, *gws)


class _Activation(nn.Module):
    _kind = 0

    def __init__(self, num_features, initial_alpha=1.0, initial_beta=1.702):
        super().__init__()
        self.num_features, self.initial_alpha, self.initial_beta = num_features, initial_alpha, initial_beta
        self.register_parameter("alpha", nn.Parameter(torch.Tensor(num_features)))
        self.register_parameter("beta", nn.Parameter(torch.Tensor(num_features)))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.constant_(self.alpha, self.initial_alpha)
        nn.init.constant_(self.beta, self.initial_beta)

    def forward(self, x):
        _require_gpu(x)
        return _ActFn.apply(x, self.alpha, self.beta, self._kind)


class Swish(_Activation):
    _kind = 0


class ShiftedSoftplus(_Activation):
    _kind = 1

    def __init__(self, num_features, initial_alpha=1.0, initial_beta=1.0):
        super().__init__(num_features, initial_alpha, initial_beta)


def _linear(x, lin: nn.Linear):
    """nn.Linear on the last axis through the engine GEMM (K = 1: the outer product it degenerates to)."""
    lead = x.shape[:-1]
    if lin.in_features == 1:
        y = x * lin.weight.view(*(1,) * len(lead), -1)
    else:
        y = _LinearFn.apply(x.reshape(-1, lin.in_features), lin.weight).view(*lead, lin.out_features)
    return y if lin.bias is None else y + lin.bias


class SphericalLinear(nn.Module):
    def __init__(self, order_in, num_in, order_out, num_out, clebsch_gordan=None, mix_orders=True, bias=True, zero_init=False):
        super().__init__()
        self.order_in, self.num_in, self.order_out, self.num_out = order_in, num_in, order_out, num_out
        self.bias, self.mix_orders, self.zero_init = bias, mix_orders, zero_init
        if mix_orders:
            assert clebsch_gordan is not None
            self.mixing = SelfMixing(order_in, order_out, num_in, clebsch_gordan)
        else:
            assert order_in == order_out
        self.linear = nn.ModuleList([nn.Linear(num_in, num_out, bias=(bias and L == 0)) for L in range(order_out + 1)])
        self.reset_parameters()

    def reset_parameters(self):
        for L in range(self.order_out + 1):
            (nn.init.zeros_ if self.zero_init else nn.init.orthogonal_)(self.linear[L].weight)
        if self.bias:
            nn.init.zeros_(self.linear[0].bias)

    def forward(self, xs):
        ys = self.mixing(xs) if self.mix_orders else xs
        if self.num_in > 1 and ys[0].is_cuda and len(ys) == self.order_out + 1:
            x, lead = _pack(ys, self.order_out, self.num_in)
            y = _SphLinearFn.apply(x, self.linear[0].bias, *[lin.weight for lin in self.linear])
            return _unpack(y, self.order_out, lead, self.num_out)
        return [_linear(y, lin) for y, lin in zip(ys, self.linear)]


class ResidualBlock(nn.Module):
    def __init__(self, order, num_features, clebsch_gordan=None, mix_orders=True, activation="swish"):
        super().__init__()
        self.order, self.num_features, self.mix_orders = order, num_features, mix_orders
        act = {"swish": Swish, "ssp": ShiftedSoftplus}.get(activation)
        if act is None:
            raise ValueError(f"Unsupported activation function: {activation}")
        self.activation_pre, self.activation_post = act(num_features), act(num_features)
        self.linear1 = SphericalLinear(order, num_features, order, num_features, clebsch_gordan, mix_orders)
        self.linear2 = SphericalLinear(order, num_features, order, num_features, clebsch_gordan, mix_orders, zero_init=True)

    def forward(self, xs):
        if xs[0].is_cuda and not (isinstance(xs, PackedList) and xs.packed is not None):
            xs = _unpack(*_pack(xs, self.order, self.num_features)[:1], self.order, xs[0].shape[:-2], self.num_features)   # pack once per stack
        ys = self.linear1(_act0(xs, self.activation_pre))
        ys = self.linear2(_act0(ys, self.activation_post))
        return _add_lists(xs, ys)


class ResidualStack(nn.Module):
    def __init__(self, num_blocks, order, num_features, clebsch_gordan=None, mix_orders=True, activation="swish"):
        super().__init__()
        self.num_blocks, self.order, self.num_features = num_blocks, order, num_features
        self.stack = nn.ModuleList([ResidualBlock(order, num_features, clebsch_gordan, mix_orders, activation) for _ in range(num_blocks)])

    def forward(self, xs):
        for block in self.stack:
            xs = block(xs)
        if isinstance(xs, PackedList) and xs.packed is not None:
            return _unpack(xs.packed, xs.order, xs.lead, xs.F)      # a fresh list object, as the reference's ``list(xs)``
        return list(xs)


class InteractionBlock(nn.Module):
    def __init__(self, order, num_features, num_basis_functions, num_residual_pre_vi, num_residual_pre_vj, num_residual_post_v, clebsch_gordan=None,
                 mix_orders=True, activation="swish"):
        super().__init__()
        self.order, self.num_features, self.num_basis_functions = order, num_features, num_basis_functions
        self.num_residual_pre_vi, self.num_residual_pre_vj, self.num_residual_post_v = num_residual_pre_vi, num_residual_pre_vj, num_residual_post_v
        act = {"swish": Swish, "ssp": ShiftedSoftplus}.get(activation)
        if act is None:
            raise ValueError(f"Unsupported activation function: {activation}")
        self.activation_i, self.activation_j, self.activation_v = act(num_features), act(num_features), act(num_features)
        self.angular_fn1 = SphericalLinear(order, 1, order, num_features, clebsch_gordan, mix_orders=False)
        self.angular_fn2 = SphericalLinear(order, 1, order, num_features, clebsch_gordan, mix_orders=False)
        self.radial_fn = nn.ModuleList([nn.Linear(num_basis_functions, num_features, bias=False) for _ in range(order + 1)])
        self.mixing = PairMixing(order, order, order, num_basis_functions, num_features, clebsch_gordan)
        self.linear_i = SphericalLinear(order, num_features, order, num_features, clebsch_gordan, mix_orders)
        self.linear_j = SphericalLinear(order, num_features, order, num_features, clebsch_gordan, mix_orders)
        self.linear_v = SphericalLinear(order, num_features, order, num_features, clebsch_gordan, mix_orders)
        self.residual_pre_vi = ResidualStack(num_residual_pre_vi, order, num_features, clebsch_gordan, mix_orders, activation)
        self.residual_pre_vj = ResidualStack(num_residual_pre_vj, order, num_features, clebsch_gordan, mix_orders, activation)
        self.residual_post_v = ResidualStack(num_residual_post_v, order, num_features, clebsch_gordan, mix_orders, activation)
        self.reset_parameters()

    def reset_parameters(self):
        for L in range(self.order + 1):
            nn.init.orthogonal_(self.radial_fn[L].weight)

    def forward(self, xs, rbf, sph, idx_i, idx_j):
        _require_gpu(rbf)
        yi = self.residual_pre_vi(xs)
        yi = _act0(yi, self.activation_i)
        yi = self.linear_i(yi)
        yj = self.residual_pre_vj(xs)
        yj = _act0(yj, self.activation_j)
        yj = self.linear_j(yj)
        pidx = idx_i if isinstance(idx_i, PairIndex) else PairIndex(idx_i, idx_j, xs[0].shape[1])
        yj = _gather_list(yj, pidx, self.order, self.num_features)                    # neighbour gather (interaction_block.py:135-137)
        vs = self.mixing(yj, self.angular_fn1(sph), rbf)
        a = self.angular_fn2(sph)
        extra = [_linear(rbf, self.radial_fn[L]) * a[L] * yj[0] for L in range(self.order + 1)]
        vs = _segment_add_list(yi, _add_lists(vs, extra), pidx, self.order, self.num_features)    # index_add over the centre atoms (:139-142)
        vs = self.residual_post_v(vs)
        vs = _act0(vs, self.activation_v)
        vs = self.linear_v(vs)
        return _add_lists(xs, vs)


class ModularBlock(nn.Module):
    def __init__(self, order, num_features, num_basis_functions, num_residual_pre_x, num_residual_post_x, num_residual_pre_vi, num_residual_pre_vj,
                 num_residual_post_v, num_residual_output, clebsch_gordan=None, mix_orders=True, activation="swish"):
        super().__init__()
        self.order, self.num_features, self.num_basis_functions = order, num_features, num_basis_functions
        self.interaction = InteractionBlock(order, num_features, num_basis_functions, num_residual_pre_vi, num_residual_pre_vj, num_residual_post_v,
                                            clebsch_gordan, mix_orders, activation)
        self.residual_pre_x = ResidualStack(num_residual_pre_x, order, num_features, clebsch_gordan, mix_orders, activation)
        self.residual_post_x = ResidualStack(num_residual_post_x, order, num_features, clebsch_gordan, mix_orders, activation)
        self.residual_out = ResidualStack(num_residual_output, order, num_features, clebsch_gordan, mix_orders, activation)

    def forward(self, xs, rbf, sph, idx_i, idx_j):
        xs = self.residual_pre_x(xs)
        xs = self.interaction(xs, rbf, sph, idx_i, idx_j)
        xs = self.residual_post_x(xs)
        ys = self.residual_out(xs)
        return xs, ys


# ---- embeddings (phisnet/nn/modules/embedding.py:13-34, spherical_embedding.py:11-27) ------------------------------------------------------------
def electron_configuration_table(zmax: int = 87) -> torch.Tensor:
    """[zmax, 16] ground-state electron configurations scaled to [0, 1]: column 0 = Z / 86, then the occupations of 1s 2s 2p 3s 3p 3d 4s 4p 4d 4f 5s 5p
    5d 6s 6p divided by the shell capacities (aufbau filling with the usual d/f exceptions).  It is the DEFAULT of the ``electron_config``
    buffer only: the buffer is part of the state_dict, so a reference checkpoint brings the reference's own table."""
    fill = [("1s", 2), ("2s", 2), ("2p", 6), ("3s", 2), ("3p", 6), ("4s", 2), ("3d", 10), ("4p", 6), ("5s", 2), ("4d", 10), ("5p", 6), ("6s", 2), ("4f", 14),
            ("5d", 10), ("6p", 6)]                                                               # Madelung filling order
    order = sorted(fill, key=lambda t: (int(t[0][0]), "spdf".index(t[0][1])))                    # column order: by shell, then subshell
    exceptions = {24: {"4s": 1, "3d": 5}, 29: {"4s": 1, "3d": 10}, 41: {"5s": 1, "4d": 4}, 42: {"5s": 1, "4d": 5}, 44: {"5s": 1, "4d": 7}, 45: {"5s": 1, "4d": 8},
                  46: {"5s": 0, "4d": 10}, 47: {"5s": 1, "4d": 10}, 57: {"4f": 0, "5d": 1}, 58: {"4f": 1, "5d": 1}, 64: {"4f": 7, "5d": 1}, 78: {"6s": 1, "5d": 9},
                  79: {"6s": 1, "5d": 10}}
    table = torch.zeros(zmax, 16)
    for Z in range(1, zmax):
        left, occ = Z, {}
        for name, cap in fill:
            occ[name] = min(cap, left)
            left -= occ[name]
        occ.update(exceptions.get(Z, {}))
        table[Z, 0] = Z / 86.0
        for c, (name, cap) in enumerate(order):
            table[Z, 1 + c] = occ[name] / cap
    return table


class Embedding(nn.Module):
    def __init__(self, num_features, Zmax=87, electron_config=None):
        super().__init__()
        self.num_features, self.Zmax = num_features, Zmax
        ec = electron_configuration_table(Zmax) if electron_config is None else torch.as_tensor(electron_config, dtype=torch.float32)
        self.register_buffer("electron_config", ec)
        self.register_parameter("element_embedding", nn.Parameter(torch.Tensor(Zmax, num_features)))
        self.config_linear = nn.Linear(ec.size(1), num_features, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.element_embedding, -3 ** 0.5, 3 ** 0.5)
        nn.init.orthogonal_(self.config_linear.weight)

    def forward(self, Z):
        table = self.element_embedding + _linear(self.electron_config, self.config_linear)          # [Zmax, F]
        return table.index_select(0, Z.reshape(-1)).view(*Z.shape, self.num_features)


class SphericalEmbedding(nn.Module):
    def __init__(self, order, num_features, Zmax=87, electron_config=None):
        super().__init__()
        self.order, self.num_features, self.Zmax = order, num_features, Zmax
        self.embedding = Embedding(num_features, Zmax, electron_config)

    def forward(self, Z):
        x0 = self.embedding(Z).unsqueeze(-2)                                                          # [..., 1, F]
        return [x0] + [x0.new_zeros(*x0.shape[:-2], 2 * L + 1, self.num_features) for L in range(1, self.order + 1)]


class _SegmentMeanFn(torch.autograd.Function):
    """rows [R, C] grouped in consecutive segments (ptr [B+1]) -> per-segment mean [B, C] (fixed summation order)."""

    @staticmethod
    def forward(ctx, rows, ptr):
        lib = _lib.load()
        r2 = rows.to(torch.float32).contiguous()
        B, Cw = ptr.numel() - 1, r2.shape[-1]
        out = torch.empty(B, Cw, device=rows.device, dtype=torch.float32)
        _lib.check(lib.nq_segment_sum(_lib.ptr(r2), None, _lib.ptr(ptr), None, B, Cw, _lib.ptr(out), _lib.stream_ptr()))
        cnt = (ptr[1:] - ptr[:-1]).to(torch.float32).clamp_(min=1).view(-1, 1)
        ctx.save_for_backward(ptr, cnt)
        return out / cnt

    @staticmethod
    def backward(ctx, g):
        ptr, cnt = ctx.saved_tensors
        seg = torch.repeat_interleave(torch.arange(ptr.numel() - 1, device=g.device), ptr[1:] - ptr[:-1])
        return (g / cnt).index_select(0, seg), None


class EnergyLayer(nn.Module):
    """phisnet/nn/modules/energy_layer.py:9-52: per-molecule means of activated scalar atom / pair features -> one energy per molecule."""

    def __init__(self, num_in, num_out, activation, zero_init=False):
        super().__init__()
        self.num_in, self.num_out, self.zero_init = num_in, num_out, zero_init
        self.linear_diagonal, self.linear_offdiagonal = nn.Linear(num_in, num_out), nn.Linear(num_in, num_out)
        self.linear_out = nn.Linear(2 * num_out, 1)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        init = nn.init.zeros_ if self.zero_init else nn.init.orthogonal_
        for lin in (self.linear_diagonal, self.linear_offdiagonal, self.linear_out):
            init(lin.weight)
            nn.init.zeros_(lin.bias)

    def forward(self, fii, fij, sizes, pair_sizes):
        dev = fii[0].device
        feats = []
        for f, lin, sz in ((fii, self.linear_diagonal, sizes), (fij, self.linear_offdiagonal, pair_sizes)):
            h = self.activation(_linear(f[0].reshape(-1, self.num_in), lin))
            sz = torch.as_tensor([int(v) for v in sz], dtype=torch.long)
            ptr = torch.cat([sz.new_zeros(1), sz.cumsum(0)]).to(dev)
            feats.append(_SegmentMeanFn.apply(h, ptr))
        full = torch.cat(feats, dim=1)
        return full @ self.linear_out.weight.t() + self.linear_out.bias                      # [B, 2*num_out] x [2*num_out, 1]: a dot product per molecule


def inferred_pair_of_pairs(n: int):
    """(idx_pi, idx_pj) of one molecule with n atoms over its n(n-1) ordered pairs (i-major, neighbours ascending): pair p = (i, j) receives the
    rbf-weighted neighbour features of every pair q = (i, k), k not in {i, j}.  INFERRED: the reference reads this table
    from ``modules/pindex_dict.npy``, which is not part of the reference tree; a model can pass its own table instead."""
    i = torch.arange(n).repeat_interleave(n - 1)
    r = torch.arange(n - 1).repeat(n)
    j = r + (r >= i).long()
    p = torch.arange(n * (n - 1))
    pi = p.repeat_interleave(n - 2) if n > 2 else p.new_zeros(0)
    # for pair p = (i, j): all k != i, j in ascending order -> pair index (i, k)
    if n > 2:
        k_all = torch.arange(n).repeat(n * (n - 1), 1)
        keep = (k_all != i[:, None]) & (k_all != j[:, None])
        k = k_all[keep].view(-1)
        ii = i.repeat_interleave(n - 2)
        pj = ii * (n - 1) + k - (k > ii).long()
    else:
        pj = p.new_zeros(0)
    return pi, pj


class NeuralNetwork(nn.Module):
    """PhiSNet (phisnet/nn/neural_network.py:31-995) on the HIP ops of this package: same constructor keywords, sub-module names and
    ``forward(atoms_batch) -> dict`` contract for the Hamiltonian / overlap prediction path.  Differences, all explicit:
      * ``clebsch_gordan``: the CG provider (the reference builds ``ClebschGordan()`` from its data file); default = tensors computed from scratch
        (nabladft_amd/cg.py, canonical signs -- checkpoints trained with the reference's table need the reference's provider);
      * ``pindex``: {molecule size: (idx_pi, idx_pj)} (the reference loads ``modules/pindex_dict.npy``, missing from its tree); default = the
        inferred table ``inferred_pair_of_pairs`` -- UNVERIFIED;
      * matrices are returned packed (``*_packed``, differentiable) and dense ([1, Norb, Norb], detached copy as in the reference's layout);
      * forces (``predict_energy`` + ``calculate_forces``, neural_network.py:737, :981-984): ``-dE/dR`` through the adjoints of the geometry bases
        (nq_sph_harm_backward, nq_bernstein_rbf_grad_r_dev), FIRST ORDER only -- set ``create_graph = False`` (inference / evaluation; the reference's
        default True keeps the second-order graph for a force loss, which the HIP ops do not provide: raises).  exp-bernstein basis only."""

    def __init__(self, max_orbitals=None, order=None, num_features=None, num_basis_functions=None, num_modules=None, num_residual_pre_x=None,
                 num_residual_post_x=None, num_residual_pre_vi=None, num_residual_pre_vj=None, num_residual_post_v=None, num_residual_output=None,
                 num_residual_pc=None, num_residual_pn=None, num_residual_ii=None, num_residual_ij=None, num_residual_full_ii=None,
                 num_residual_full_ij=None, num_residual_core_ii=None, num_residual_core_ij=None, num_residual_over_ij=None, basis_functions=None,
                 cutoff=None, activation=None, load_from=None, Zmax=87, num_energy_features=64, fallback_args=None, clebsch_gordan=None, pindex=None,
                 electron_config=None):
        super().__init__()
        from . import cg as _cg
        from .hamiltonian import IrrepsAssembler, compute_matrix_irreps
        from .so3 import RADIAL_BASES
        saved_state = None
        if load_from is not None:   # neural_network.py:97-140: hyper-parameters come from the file ('args' namespace of the training script, or the flat dict `save` writes)
            from argparse import Namespace
            # tensors + the training script's argparse.Namespace only: no arbitrary pickle execution for a checkpoint path a user passes
            # (NQ_UNSAFE_LOAD=1 restores the full unpickler for files that carry other Python objects)
            if os.environ.get("NQ_UNSAFE_LOAD") == "1":
                saved_state = torch.load(load_from, map_location="cpu", weights_only=False)
            else:
                with torch.serialization.safe_globals([Namespace]):
                    saved_state = torch.load(load_from, map_location="cpu", weights_only=True)
            try:
                args = saved_state["args"]
            except KeyError:
                args = Namespace(**saved_state)
            if isinstance(args, dict):
                args = Namespace(**args)
            max_orbitals = args.max_orbitals if max_orbitals is None else max_orbitals
            order, num_features, num_basis_functions, num_modules = args.order, args.num_features, args.num_basis_functions, args.num_modules
            num_residual_pre_x, num_residual_post_x = args.num_residual_pre_x, args.num_residual_post_x
            num_residual_pre_vi, num_residual_pre_vj, num_residual_post_v = args.num_residual_pre_vi, args.num_residual_pre_vj, args.num_residual_post_v
            num_residual_output, num_residual_pc, num_residual_pn = args.num_residual_output, args.num_residual_pc, args.num_residual_pn
            num_residual_ii, num_residual_ij = args.num_residual_ii, args.num_residual_ij
            num_residual_full_ii, num_residual_full_ij = args.num_residual_full_ii, args.num_residual_full_ij
            num_residual_core_ii, num_residual_core_ij, num_residual_over_ij = args.num_residual_core_ii, args.num_residual_core_ij, args.num_residual_over_ij
            basis_functions, cutoff, activation = args.basis_functions, args.cutoff, args.activation
        self._hp = dict(num_residual_pre_x=num_residual_pre_x, num_residual_post_x=num_residual_post_x, num_residual_pre_vi=num_residual_pre_vi,
                        num_residual_pre_vj=num_residual_pre_vj, num_residual_post_v=num_residual_post_v, num_residual_output=num_residual_output,
                        num_residual_pc=num_residual_pc, num_residual_pn=num_residual_pn, num_residual_ii=num_residual_ii, num_residual_ij=num_residual_ij,
                        num_residual_full_ii=num_residual_full_ii, num_residual_full_ij=num_residual_full_ij, num_residual_core_ii=num_residual_core_ii,
                        num_residual_core_ij=num_residual_core_ij, num_residual_over_ij=num_residual_over_ij, basis_functions=basis_functions)
        if basis_functions not in ("exp-gaussian", "exp-bernstein", "gaussian", "bernstein"):         # the four choices of neural_network.py:210-221
            raise ValueError(f"basis function type: {basis_functions} is not supported")
        self.calculate_full_hamiltonian = self.calculate_core_hamiltonian = self.calculate_overlap_matrix = True
        self.calculate_energy = self.predict_energy = self.calculate_forces = False
        self.create_graph = True                       # neural_network.py:93; forces need False here (first-order adjoints only)
        self.max_orbitals, self.order, self.num_features, self.num_basis_functions = max_orbitals, order, num_features, num_basis_functions
        self.num_modules, self.cutoff, self.activation, self.Zmax = num_modules, cutoff, activation, Zmax
        order_max = max(l for orbs in max_orbitals for _, l in orbs)
        if order < order_max:
            raise ValueError(f"An orbital with L={order_max} was found, but the neural network was initialized with L={order}")
        self.clebsch_gordan = clebsch_gordan if clebsch_gordan is not None else (lambda a, b, c: torch.tensor(_cg.canonical(a, b, c), dtype=torch.float32))
        cgp = self.clebsch_gordan
        F, K = num_features, num_basis_functions
        act = {"swish": Swish, "ssp": ShiftedSoftplus}[activation]
        self.embedding = SphericalEmbedding(order, F, Zmax, electron_config)
        self.radial_basis_functions = RADIAL_BASES[basis_functions](K, cutoff)
        self.module = nn.ModuleList([ModularBlock(order, F, K, num_residual_pre_x, num_residual_post_x, num_residual_pre_vi, num_residual_pre_vj,
                                                  num_residual_post_v, num_residual_output, cgp, True, activation) for _ in range(num_modules)])
        self.angular_fn = SphericalLinear(order, 1, order, F, cgp, mix_orders=False)
        self.mix_s = PairMixing(order, order, order, K, F, cgp)
        self.mix_ij = PairMixing(order, order, order, K, F, cgp)
        self.radial_ii = nn.ModuleList([nn.Linear(K, F, bias=False) for _ in range(order + 1)])
        self.radial_ij = nn.ModuleList([nn.Linear(K, F, bias=False) for _ in range(order + 1)])
        mk = lambda n: ResidualStack(n, order, F, cgp, True, activation)
        self.residual_pc, self.residual_pn, self.residual_ii, self.residual_ij = mk(num_residual_pc), mk(num_residual_pn), mk(num_residual_ii), mk(num_residual_ij)
        self.residual_full_ii, self.residual_full_ij = mk(num_residual_full_ii), mk(num_residual_full_ij)
        self.residual_core_ii, self.residual_core_ij = mk(num_residual_core_ii), mk(num_residual_core_ij)
        self.residual_over_ij = mk(num_residual_over_ij)
        self.activation_full_ii, self.activation_full_ij, self.activation_core_ii = act(F), act(F), act(F)
        self.activation_core_ij, self.activation_over_ij = act(F), act(F)
        self.activation_energy = act(num_energy_features)
        self.num_energy_features = num_energy_features
        # index dictionaries for collecting irreps (neural_network.py:368-417)
        number_L, self.irreps_ii = [0] * (2 * order_max + 1), {}
        for orbs in max_orbitals:
            self.irreps_ii, number_L = compute_matrix_irreps(orbs, orbs, self.irreps_ii, number_L)
        n_ii = max(number_L)
        out = lambda n: SphericalLinear(order, F, 2 * order_max, n, cgp, zero_init=True)
        self.output_full_ii, self.output_core_ii, self.output_over_ii = out(n_ii), out(n_ii), out(n_ii)
        for L in range(self.output_over_ii.order_out + 1):
            self.output_over_ii.linear[L].weight.requires_grad = False                       # diagonal overlap blocks are constant (:402-403)
        number_L, self.irreps_ij = [0] * (2 * order_max + 1), {}
        for i, oi in enumerate(max_orbitals):
            for j, oj in enumerate(max_orbitals):
                if i == j:
                    continue
                self.irreps_ij, number_L = compute_matrix_irreps(oi, oj, self.irreps_ij, number_L)
        n_ij = max(number_L) if self.irreps_ij else 1
        self.output_full_ij, self.output_core_ij, self.output_over_ij = out(n_ij), out(n_ij), out(n_ij)
        for L in range(order + 1):                                                           # reset_parameters (:460-463)
            nn.init.orthogonal_(self.radial_ii[L].weight)
            nn.init.orthogonal_(self.radial_ij[L].weight)
        self.energy_predictor = EnergyLayer(F, num_energy_features, zero_init=False, activation=self.activation_energy)
        self._n_out = (n_ii, n_ij)
        self._order_out = 2 * order_max
        self._pindex = pindex
        a2o = {}
        for orbs in max_orbitals:
            a2o.setdefault(int(orbs[0][0]), tuple((int(zz), int(l)) for zz, l in orbs))
        self._assembler = IrrepsAssembler(a2o, self.irreps_ii, self.irreps_ij, cgp)
        if saved_state is not None:
            # neural_network.py:445-449: non-strict load; the reference constructs its EnergyLayer AFTER this load (:453), so the energy predictor keeps its fresh
            # initialisation there -- reproduced by leaving its keys out
            try:
                sd = saved_state["model_state_dict"]
            except KeyError:
                sd = saved_state["state_dict"]
            self.load_state_dict({k: v for k, v in sd.items() if not k.startswith("energy_predictor.")}, strict=False)

    def save(self, PATH):
        """The model and all hyper-parameters in the flat layout of neural_network.py:470-503 (a file `load_from` accepts, here and in the reference)."""
        torch.save(dict(state_dict=self.state_dict(), max_orbitals=self.max_orbitals, order=self.order, num_features=self.num_features,
                        num_basis_functions=self.num_basis_functions, num_modules=self.num_modules, cutoff=self.cutoff, activation=self.activation, Zmax=self.Zmax,
                        num_energy_features=self.num_energy_features, **self._hp), PATH)

    def get_number_of_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    # ---- helpers ------------------------------------------------------------------------------------------------------------------------
    def fill_idx(self, molecule_size, device):
        """All ordered atom pairs of every molecule (i-major) and the pair-of-pairs index (neural_network.py:515-561)."""
        from .hamiltonian import full_pair_index
        sizes = torch.as_tensor(molecule_size).long().cpu()
        ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
        ei = full_pair_index(ptr)
        self.idx_i, self.idx_j = ei[1].to(device), ei[0].to(device)
        pis, pjs, off = [], [], 0
        for n in sizes.tolist():
            pi, pj = (self._pindex[n] if self._pindex is not None else inferred_pair_of_pairs(n))
            pis.append(torch.as_tensor(pi, dtype=torch.long) + off), pjs.append(torch.as_tensor(pj, dtype=torch.long) + off)
            off += n * (n - 1)
        self.idx_pi, self.idx_pj = torch.cat(pis).to(device), torch.cat(pjs).to(device)
        return ptr.to(device)

    def _heads(self, f, res, act, out):
        g = res(f)
        g = _act0(g, act)
        return out(g)

    def _pack(self, fs, n_out):
        x = fs.packed if isinstance(fs, PackedList) and fs.packed is not None else torch.cat([f[0] for f in fs], dim=-2)   # [rows, (Lout+1)^2, n_out]
        pad = max(self._n_out) - n_out
        return torch.nn.functional.pad(x, (0, pad)) if pad else x

    def prepare(self, atoms_batch):
        """Everything of a batch that depends only on its composition (molecule sizes, atomic numbers): pair lists, the pair-of-pairs index, the sorted
        index structures of the gather / segment-sum kernels and the matrix-assembly tables.  Building it reads sizes on the host; pass the result back as
        ``atoms_batch["prepared"]`` and ``forward`` issues no host synchronisation at all (a training step on a repeating composition can then be captured
        into a HIP graph, trainer.GraphedStep)."""
        from types import SimpleNamespace
        R = atoms_batch["positions"].view(1, -1, 3)
        _require_gpu(R)
        Z = atoms_batch["atomic_numbers"].view(1, -1).long()
        ptr = self.fill_idx(atoms_batch["molecule_size"], R.device)
        idx_i, idx_j = self.idx_i, self.idx_j
        N, P = Z.shape[1], idx_i.numel()
        pidx = PairIndex(idx_i, idx_j, N)
        # pair-of-pairs index as a second sorted pair structure: rows = pairs, "neighbours" = the pairs listed in idx_pj
        ppidx = PairIndex(self.idx_pi, self.idx_pj, P) if self.idx_pi.numel() else None
        sizes = [int(v) for v in torch.as_tensor(atoms_batch["molecule_size"]).tolist()]
        return SimpleNamespace(ptr=ptr, idx_i=idx_i, idx_j=idx_j, idx_pi=self.idx_pi, idx_pj=self.idx_pj, pidx=pidx, ppidx=ppidx, swap=_SwapIndex(pidx),
                               plan=self._assembler.plan(Z[0], ptr, idx_i, idx_j), sizes=sizes, checked=False)

    def forward(self, atoms_batch):
        if self.predict_energy and self.calculate_forces and not torch.is_grad_enabled():
            with torch.enable_grad():                   # forces are a derivative: the graph is needed even under torch.no_grad()
                return self.forward(atoms_batch)
        R = atoms_batch["positions"]
        R = R.view(1, -1, 3)
        _require_gpu(R)
        Z = atoms_batch["atomic_numbers"].view(1, -1).long()
        prep = atoms_batch.get("prepared") if isinstance(atoms_batch, dict) else None
        if prep is None:
            prep = self.prepare(atoms_batch)
        self.idx_i, self.idx_j, self.idx_pi, self.idx_pj = prep.idx_i, prep.idx_j, prep.idx_pi, prep.idx_pj
        ptr, idx_i, idx_j, pidx, ppidx = prep.ptr, prep.idx_i, prep.idx_j, prep.pidx, prep.ppidx
        N, P = Z.shape[1], idx_i.numel()
        want_forces = bool(self.predict_energy and self.calculate_forces)
        grad_was_enabled = torch.is_grad_enabled()
        if want_forces:
            if self.create_graph:
                raise NotImplementedError("nabladft_amd.phisnet.NeuralNetwork: forces with create_graph=True (a force LOSS: second-order derivatives) are not "
                                          "built; set net.create_graph = False for inference / evaluation forces")
            R = R.detach().clone().requires_grad_(True)          # neural_network.py:737 (R.requires_grad = True), without touching the caller's tensor
        rij = R[0].index_select(0, idx_j) - R[0].index_select(0, idx_i)
        dij = rij.norm(dim=-1, keepdim=True)
        uij = rij / dij
        from .so3 import spherical_harmonics
        rbf = self.radial_basis_functions(dij).view(1, P, 1, self.num_basis_functions)
        sph = [s.view(1, P, -1, 1) for s in spherical_harmonics(self.order, uij)]
        xs = self.embedding(Z)
        swap = prep.swap
        gather_i = lambda ts: _gather_list(ts, swap, self.order, self.num_features)      # all orders of the centre atoms, one launch
        gather_j = lambda ts: _gather_list(ts, pidx, self.order, self.num_features)
        results = {}
        if self.calculate_overlap_matrix:
            fii_over = self.output_over_ii(xs)
            a = self.angular_fn(sph)
            si = gather_i(xs)
            sj = [_GatherFn.apply(xs[0], pidx)] + [a[L] for L in range(1, self.order + 1)]
            fij_over = self._heads(self.mix_s(si, sj, rbf), self.residual_over_ij, self.activation_over_ij, self.output_over_ij)
        fs = [torch.zeros_like(x) for x in xs]
        for module in self.module:
            xs, ys = module(xs, rbf, sph, pidx, idx_j)
            fs = _add_lists(fs, ys)
        fpc, fpn = self.residual_pc(fs), self.residual_pn(fs)
        fpn_g = gather_j(fpn)
        fpn_j_ii = [_linear(rbf, self.radial_ii[L]) * fpn_g[L] for L in range(self.order + 1)]
        fii = self.residual_ii(_segment_add_list(fpc, fpn_j_ii, pidx, self.order, self.num_features))
        fij = self.mix_ij(gather_i(fpc), gather_j(fpc), rbf)
        if ppidx is not None:
            fpn_j = [_linear(rbf, self.radial_ij[L]) * fpn_g[L] for L in range(self.order + 1)]
            fij = _segment_add_list(fij, _gather_list(fpn_j, ppidx, self.order, self.num_features), ppidx, self.order, self.num_features)
        fij = self.residual_ij(fij)
        asm = self._assembler
        plan = prep.plan
        n_ii, n_ij = self._n_out
        if self.calculate_full_hamiltonian:
            f1 = self._heads(fii, self.residual_full_ii, self.activation_full_ii, self.output_full_ii)
            f2 = self._heads(fij, self.residual_full_ij, self.activation_full_ij, self.output_full_ij)
            results["full_hamiltonian_packed"] = asm.assemble(plan, self._pack(f1, n_ii), self._pack(f2, n_ij), symmetrize=True)
        if self.calculate_core_hamiltonian:
            f1 = self._heads(fii, self.residual_core_ii, self.activation_core_ii, self.output_core_ii)
            f2 = self._heads(fij, self.residual_core_ij, self.activation_core_ij, self.output_core_ij)
            results["core_hamiltonian_packed"] = asm.assemble(plan, self._pack(f1, n_ii), self._pack(f2, n_ij), symmetrize=True)
        if self.calculate_overlap_matrix:
            results["overlap_matrix_packed"] = asm.assemble(plan, self._pack(fii_over, n_ii), self._pack(fij_over, n_ij), symmetrize=True, unit_diagonal=True)
        if not prep.checked:                      # error flag of the assembly kernels: read once per prepared composition (a host synchronisation)
            asm.check(plan)
            prep.checked = True
        for k in ("full_hamiltonian", "core_hamiltonian", "overlap_matrix"):
            if k + "_packed" in results:
                results[k] = asm.to_dense(plan, results[k + "_packed"].detach()).unsqueeze(0)
        norb, eye = plan.m_total, None
        for k in ("full_hamiltonian", "core_hamiltonian", "overlap_matrix"):                # a disabled matrix is the identity (:935-966)
            if k not in results:
                eye = torch.eye(norb, device=R.device, dtype=R.dtype).unsqueeze(0) if eye is None else eye
                results[k] = eye
        if self.predict_energy:
            sizes = prep.sizes
            results["energy"] = self.energy_predictor(fii, fij, sizes, [n * (n - 1) for n in sizes])
        else:
            results["energy"] = torch.zeros(1, 1, device=R.device, dtype=R.dtype)
        if want_forces:                                            # neural_network.py:981-984
            with torch.enable_grad():
                results["forces"] = -torch.autograd.grad(torch.sum(results["energy"]), R, create_graph=False, retain_graph=grad_was_enabled)[0]   # keep the graph only for a caller who will backpropagate through the other outputs
        else:
            results["forces"] = torch.zeros_like(R)
        results["orbital_energies"] = torch.zeros(1, norb, device=R.device, dtype=R.dtype)
        results["orbital_coefficients"] = torch.zeros(1, norb, norb, device=R.device, dtype=R.dtype)
        results["plan"] = plan
        return results


class _SwapIndex:
    """View of a PairIndex whose gather reads the centre atoms (idx_i) instead of the neighbours."""

    def __init__(self, pidx: PairIndex):
        self.idx_j, self.N = pidx.idx_i, pidx.N
        self.order_j = torch.arange(pidx.idx_i.numel(), device=pidx.idx_i.device)        # idx_i is sorted: the pairs of atom n are contiguous
        self.ptr_j = pidx.ptr_i
