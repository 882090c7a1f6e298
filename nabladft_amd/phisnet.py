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

import torch
from torch import nn

from . import _lib
from .so3 import PairMixing, SelfMixing, _LinearFn, _require_gpu


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
        ys = list(xs)
        ys[0] = self.activation_pre(ys[0])
        ys = self.linear1(ys)
        ys[0] = self.activation_post(ys[0])
        ys = self.linear2(ys)
        return [x + y for x, y in zip(xs, ys)]


class ResidualStack(nn.Module):
    def __init__(self, num_blocks, order, num_features, clebsch_gordan=None, mix_orders=True, activation="swish"):
        super().__init__()
        self.num_blocks, self.order, self.num_features = num_blocks, order, num_features
        self.stack = nn.ModuleList([ResidualBlock(order, num_features, clebsch_gordan, mix_orders, activation) for _ in range(num_blocks)])

    def forward(self, xs):
        for block in self.stack:
            xs = block(xs)
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
        yi[0] = self.activation_i(yi[0])
        yi = self.linear_i(yi)
        yj = self.residual_pre_vj(xs)
        yj[0] = self.activation_j(yj[0])
        yj = self.linear_j(yj)
        pidx = idx_i if isinstance(idx_i, PairIndex) else PairIndex(idx_i, idx_j, xs[0].shape[1])
        yj = [_GatherFn.apply(y, pidx) for y in yj]                                   # neighbour gather (interaction_block.py:135-137)
        vs = self.mixing(yj, self.angular_fn1(sph), rbf)
        a = self.angular_fn2(sph)
        vs = [_SegmentAddFn.apply(yi[L], vs[L] + _linear(rbf, self.radial_fn[L]) * a[L] * yj[0], pidx) for L in range(self.order + 1)]
        vs = self.residual_post_v(vs)
        vs[0] = self.activation_v(vs[0])
        vs = self.linear_v(vs)
        return [x + v for x, v in zip(xs, vs)]


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
