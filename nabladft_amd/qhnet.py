"""QHNet on MI355X (BASELINE.json configs[3] = config/qhnet.yaml).

Mirrors, with the reference's constructor arguments, sub-module / parameter names (state_dict compatible) and forward contract:
  QHNet                     /root/reference/nablaDFT/qhnet/qhnet.py:24-342      (forward :186-252, build_graph :254-291)
  ConvNetLayer / ConvLayer  /root/reference/nablaDFT/qhnet/layers.py:297-343, :150-274
  NormGate, InnerProduct    layers.py:123-147, :277-294
  PairNetLayer              layers.py:346-492
  SelfNetLayer              layers.py:495-582
  Expansion                 layers.py:585-682
  get_feasible_irrep        layers.py:44-83  (path selection and QHNet's own path weights, incl. the shadowed loop variable at :73)
and the e3nn 0.5.1 modules they are built from (third-party; semantics restated from the published behaviour, SURVEY.md Appendix A,
PARITY UNPINNED): o3.Linear -> O3Linear, nn.FullyConnectedNet -> FullyConnectedNet, o3.TensorProduct / o3.Norm / ElementwiseTensorProduct
-> the HIP kernels of csrc/qhnet.hip and csrc/so3.hip.

Device layout: irreps features are [rows, 25, C] (component l*l + m + l, channel fastest), never e3nn's [mul, 2l+1]; parameters keep e3nn's
shapes so checkpoints load.  All arithmetic runs in HIP kernels behind the C ABI (include/nablaq.h): MFMA GEMMs for every dense map, the
Clebsch-Gordan contractions with compile-time coefficients, CSR gathers instead of scatter (no atomics, reproducible).  torch is used for
device memory, autograd plumbing and parameter-sized reshapes only.  GPU only: no CPU fallback.

Reference quirks kept: ConvLayer builds its edge invariants from x[dst] twice (layers.py:240-246); PairNetLayer.linear_node_pair is the
second assignment (layers.py:440); Expansion.weights is an unused parameter (layers.py:594-595); SelfNet / PairNet treat the 1o / 3o parts
of the node features as even irreps (qhnet.py:56-58).
"""
import ctypes as C
import copy
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib, cg
from .hamiltonian import BlockAssembler, full_pair_index, transpose_index
from .painn import build_neighbor_list
from .phisnet import _SphLinearFn
from .so3 import ExponentialBernsteinRadialBasisFunctions, _LinearFn, _MixFn

LMAX = 4
NCOMP = (LMAX + 1) ** 2


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("nabladft_amd.qhnet runs on MI355X only (no CPU fallback): move the model and the batch to cuda")


# ---- e3nn normalisation constants -----------------------------------------------------------------------------------------------------------
_MOM = {}


def normalize2mom_constant(kind: str) -> float:
    """e3nn.math.normalize2mom (used by e3nn.nn.FullyConnectedNet): cst = E[f(z)^2]^(-1/2), estimated by e3nn with 10^6 float64 normal
    samples from a CPU generator seeded with 0 -- reproduced literally because the Monte-Carlo value, not the exact integral, is what
    trained checkpoints contain."""
    if kind not in _MOM:
        gen = torch.Generator(device="cpu").manual_seed(0)
        z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
        f = {"ssp": lambda t: torch.nn.functional.softplus(t) - math.log(2.0), "silu": torch.nn.functional.silu}[kind]
        _MOM[kind] = float(f(z).pow(2).mean().pow(-0.5))
    return _MOM[kind]


def conv_paths(first_layer: bool):
    """Instructions of ConvLayer.tp_node in e3nn order (get_feasible_irrep, layers.py:48-56): (l1, l2, lo) with even l1 + l2 + lo (node
    irreps and spherical harmonics both have parity (-1)^l), l1 = 0 only for the first layer."""
    return [p for p in cg.ALL_PATHS if (p[0] + p[1] + p[2]) % 2 == 0 and (p[0] == 0 or not first_layer)]


def path_constants(paths):
    """sign(e3nn 3j vs kernel tensor) * e3nn coefficient sqrt(alpha): alpha = dim(lo) * path_weight / #(paths into lo) with QHNet's
    path_weight = sqrt(dim(lo) / #paths) (layers.py:60-77: the comprehension variable shadows ``ins``, so x = #paths * fan, fan = 1)."""
    cnt = {}
    for (_, _, lo) in paths:
        cnt[lo] = cnt.get(lo, 0) + 1
    out = []
    for (l1, l2, lo) in paths:
        d = 2 * lo + 1
        out.append(cg.e3nn_sign(l1, l2, lo) * math.sqrt(d / cnt[lo] * math.sqrt(d / len(paths))))
    return out


def _path_index(paths):
    arr = (C.c_int8 * len(cg.ALL_PATHS))(*([-1] * len(cg.ALL_PATHS)))
    for i, p in enumerate(paths):
        arr[cg.PATH_ID[p]] = i
    return arr


# ---- graph ------------------------------------------------------------------------------------------------------------------------------------
class _Csr:
    """CSR by owner atom (reference: src = row 1 of edge_index), neighbours (dst = row 0) ascending; int32 device arrays."""

    def __init__(self, nl):
        self.N, self.R = nl.N, nl.E
        self.row_ptr, self.col, self.own, self.rev, self.geom = nl.t["row_ptr"], nl.t["col"], nl.t["dst"], nl.t["rev"], nl.t["geom"]
        self._nl = nl

    @property
    def edge_index(self):
        return torch.stack([self.col.long(), self.own.long()])


class _Graphs:
    pass


# ---- autograd functions over the C ABI ---------------------------------------------------------------------------------------------------------
def _f32(t):
    return t.to(torch.float32).contiguous()


class _MatmulFn(torch.autograd.Function):
    """y = x @ W, W [in, out] (e3nn's FullyConnectedNet layout) on the fp32 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, W):
        lib = _lib.load()
        x, W = _f32(x), _f32(W)
        M, K = x.shape
        N = W.shape[1]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(x), _lib.ptr(W), _lib.ptr(y), M, K, N, 0, _lib.stream_ptr()))
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        g = _f32(g)
        M, K = x.shape
        N = W.shape[1]
        gx = torch.empty_like(x)
        _lib.check(lib.nq_linear_forward(_lib.ptr(g), _lib.ptr(W), None, _lib.ptr(gx), None, M, K, N, _lib.stream_ptr()))
        gW = torch.empty_like(W)
        scr = torch.empty(int(lib.nq_weight_grad_scratch_floats(M, K, N)) + 64, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_weight_grad(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gW), M, K, N, _lib.ptr(scr), _lib.stream_ptr()))
        return gx, gW


class _LinearBiasFn(torch.autograd.Function):
    """torch.nn.Linear (+ optional SiLU) through nq_linear_forward's fused bias / SiLU epilogue."""

    @staticmethod
    def forward(ctx, x, W, b, silu):
        lib = _lib.load()
        x, W, b = _f32(x), _f32(W), _f32(b)
        M, K = x.shape
        N = W.shape[0]
        pre = torch.empty(M, N, device=x.device, dtype=torch.float32)
        post = torch.empty_like(pre) if silu else None
        _lib.check(lib.nq_linear_forward(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(pre), _lib.ptr(post), M, N, K, _lib.stream_ptr()))
        ctx.save_for_backward(x, W, pre if silu else x.new_zeros(0))
        ctx.silu = silu
        return post if silu else pre

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W, pre = ctx.saved_tensors
        g = _f32(g)
        M, K = x.shape
        N = W.shape[0]
        if ctx.silu:
            gp = torch.empty_like(g)
            _lib.check(lib.nq_qh_act(_lib.ptr(pre), _lib.ptr(g), 0, 1.0, g.numel(), _lib.ptr(gp), _lib.stream_ptr()))
            g = gp
        gx = torch.empty_like(x)
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, 0, _lib.stream_ptr()))
        gW = torch.empty_like(W)
        gb = torch.empty(N, device=g.device, dtype=torch.float32)
        scr = torch.empty(int(lib.nq_weight_grad_scratch_floats(M, N, K)) + 64, device=x.device, dtype=torch.float32)
        # weight and bias gradient in one launch: the column sums of g are taken from the operand registers of the contraction (fixed order)
        _lib.check(lib.nq_linear_weight_grad_bias(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), _lib.ptr(gb), M, N, K, _lib.ptr(scr), _lib.stream_ptr()))
        return gx, gW, gb, None


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, cst):
        lib = _lib.load()
        x = _f32(x)
        y = torch.empty_like(x)
        _lib.check(lib.nq_qh_act(_lib.ptr(x), None, kind, cst, x.numel(), _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x)
        ctx.meta = (kind, cst)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        g = _f32(g)
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_act(_lib.ptr(x), _lib.ptr(g), ctx.meta[0], ctx.meta[1], x.numel(), _lib.ptr(gx), _lib.stream_ptr()))
        return gx, None, None


class _InvFn(torch.autograd.Function):
    """Edge / pair invariants s0 (InnerProduct + the concatenations of layers.py:236-258, :466-476)."""

    @staticmethod
    def forward(ctx, x, csr, second_from_owner):
        lib = _lib.load()
        x = _f32(x)
        N, ncomp, Cc = x.shape
        lmax = int(round(math.sqrt(ncomp))) - 1
        s0 = torch.empty(csr.R, (2 + lmax) * Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_invariants_forward(_lib.ptr(x), N, ncomp, Cc, _lib.ptr(csr.own), _lib.ptr(csr.col), csr.R, int(second_from_owner), _lib.ptr(s0),
                                                _lib.stream_ptr()))
        ctx.save_for_backward(x)
        ctx.meta = (csr, second_from_owner)
        return s0

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        csr, sfo = ctx.meta
        N, ncomp, Cc = x.shape
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_invariants_backward(_lib.ptr(x), _lib.ptr(_f32(g)), N, ncomp, Cc, _lib.ptr(csr.row_ptr), _lib.ptr(csr.col), _lib.ptr(csr.rev), int(sfo),
                                                 _lib.ptr(gx), _lib.stream_ptr()))
        return gx, None, None


class _ConvFn(torch.autograd.Function):
    """ConvLayer.tp_node + scatter (+ self connection): out [N, 25, C].  Messages are computed per edge (one thread per edge and channel) and
    summed per receiving atom along the reverse slots of its CSR row."""

    @staticmethod
    def forward(ctx, x, w1, w2, csr, sh, path_set, add_self):
        lib = _lib.load()
        x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
        N, n1, Cc = x.shape
        msg = torch.empty(csr.R, NCOMP, Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_tp_forward(_lib.ptr(x), n1, _lib.ptr(csr.own), _lib.ptr(sh), None, _lib.ptr(w1), _lib.ptr(w2), csr.R, Cc, path_set, _lib.ptr(msg),
                                        _lib.stream_ptr()))
        out = torch.empty(N, NCOMP, Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_pair_reduce(None, _lib.ptr(msg), _lib.ptr(x) if add_self else None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), N, NCOMP * Cc,
                                         _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(x, w1, w2)
        ctx.meta = (csr, sh, path_set, add_self)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w1, w2 = ctx.saved_tensors
        csr, sh, path_set, add_self = ctx.meta
        N, n1, Cc = x.shape
        g = _f32(g)
        grow = torch.empty(csr.R, n1, Cc, device=x.device, dtype=torch.float32)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.nq_qh_tp_backward(_lib.ptr(x), n1, _lib.ptr(csr.own), _lib.ptr(sh), None, _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(g), _lib.ptr(csr.col), csr.R, Cc,
                                         path_set, _lib.ptr(grow), None, _lib.ptr(gw1), _lib.ptr(gw2), _lib.stream_ptr()))
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_pair_reduce(_lib.ptr(grow), None, _lib.ptr(g) if add_self else None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), N, n1 * Cc, _lib.ptr(gx),
                                         _lib.stream_ptr()))
        return gx, gw1, gw2, None, None, None, None


class _PairMixFn(torch.autograd.Function):
    """PairNetLayer.tp_node_pair: y[r] = TP_uuu(x[src(r)], x[dst(r)], w1[r] * w2[r]) over the full pair list."""

    @staticmethod
    def forward(ctx, x, w1, w2, csr):
        lib = _lib.load()
        x, w1, w2 = _f32(x), _f32(w1), _f32(w2)
        N, _, Cc = x.shape
        y = torch.empty(csr.R, NCOMP, Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_tp_forward(_lib.ptr(x), NCOMP, _lib.ptr(csr.own), None, _lib.ptr(csr.col), _lib.ptr(w1), _lib.ptr(w2), csr.R, Cc, 0, _lib.ptr(y),
                                        _lib.stream_ptr()))
        ctx.save_for_backward(x, w1, w2)
        ctx.csr = csr
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w1, w2 = ctx.saved_tensors
        csr = ctx.csr
        N, _, Cc = x.shape
        g = _f32(g)
        g1 = torch.empty(csr.R, NCOMP, Cc, device=x.device, dtype=torch.float32)
        g2 = torch.empty_like(g1)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.nq_qh_tp_backward(_lib.ptr(x), NCOMP, _lib.ptr(csr.own), None, _lib.ptr(csr.col), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(g), None, csr.R, Cc, 0,
                                         _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(gw1), _lib.ptr(gw2), _lib.stream_ptr()))
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_pair_reduce(_lib.ptr(g1), _lib.ptr(g2), None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), N, NCOMP * Cc, _lib.ptr(gx), _lib.stream_ptr()))
        return gx, gw1, gw2, None


PAIR_GENERATOR_FUSION = os.environ.get("NQ_QH_GEN", "0") == "1"


def set_pair_generator_fusion(on: bool):
    """PairNetLayer: generate the per-pair path weights inside the forward tensor-product kernel (csrc/qhgen.hip) instead of materialising the two
    [pairs, 65 * C] factors.  Default per the measurement in profiles/r06_qhnet_generator_fusion.txt."""
    global PAIR_GENERATOR_FUSION
    PAIR_GENERATOR_FUSION = bool(on)


class _PairMixGenFn(torch.autograd.Function):
    """PairNetLayer.tp_node_pair with weight = (h1 @ W1) * (h2 @ W2^T + b2) generated in the kernel (layers.py:476-481), forward and reverse: the two
    [pairs, 65 C] factors never exist; their adjoints do (they are the operands of the generators' weight-gradient and input-gradient products)."""

    @staticmethod
    def forward(ctx, x, h1, W1, h2, W2, b2, csr):
        lib = _lib.load()
        x, h1, W1, h2, W2, b2 = _f32(x), _f32(h1), _f32(W1), _f32(h2), _f32(W2), _f32(b2)
        N, _, Cc = x.shape
        K = h1.shape[1]
        assert h2.shape[1] == K and W1.shape == (K, W2.shape[0]) and W2.shape[1] == K
        nfl = int(lib.nq_qh_gen_fragment_floats(Cc, K))
        frag = torch.empty(2 * nfl, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_gen_presplit(_lib.ptr(W1), None, K, Cc, 0, _lib.ptr(frag), _lib.stream_ptr()))
        _lib.check(lib.nq_qh_gen_presplit(_lib.ptr(W2), None, K, Cc, 1, _lib.ptr(frag[nfl:]), _lib.stream_ptr()))
        y = torch.empty(csr.R, NCOMP, Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_tp_forward_gen(_lib.ptr(x), _lib.ptr(csr.own), _lib.ptr(csr.col), _lib.ptr(h1), _lib.ptr(h2), _lib.ptr(frag), _lib.ptr(frag[nfl:]),
                                            _lib.ptr(b2), csr.R, Cc, K, _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x, h1, W1, h2, W2, b2, frag)
        ctx.csr = csr
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, h1, W1, h2, W2, b2, frag = ctx.saved_tensors
        csr = ctx.csr
        N, _, Cc = x.shape
        R, K, ncol = csr.R, h1.shape[1], W1.shape[1]
        nfl = frag.numel() // 2
        g = _f32(g)
        g1 = torch.empty(R, NCOMP, Cc, device=x.device, dtype=torch.float32)
        g2 = torch.empty_like(g1)
        gw1 = torch.empty(R, ncol, device=x.device, dtype=torch.float32)
        gw2 = torch.empty_like(gw1)
        _lib.check(lib.nq_qh_tp_backward_gen(_lib.ptr(x), _lib.ptr(csr.own), _lib.ptr(csr.col), _lib.ptr(h1), _lib.ptr(h2), _lib.ptr(frag), _lib.ptr(frag[nfl:]), _lib.ptr(b2),
                                             _lib.ptr(g), R, Cc, K, _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(gw1), _lib.ptr(gw2), _lib.stream_ptr()))
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_pair_reduce(_lib.ptr(g1), _lib.ptr(g2), None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), N, NCOMP * Cc, _lib.ptr(gx), _lib.stream_ptr()))
        # the generators' own adjoints: w1 = h1 @ W1 (as _MatmulFn), w2 = h2 @ W2^T + b2 (as _LinearBiasFn)
        gh1 = torch.empty_like(h1)
        _lib.check(lib.nq_linear_forward(_lib.ptr(gw1), _lib.ptr(W1), None, _lib.ptr(gh1), None, R, K, ncol, _lib.stream_ptr()))
        gW1 = torch.empty_like(W1)
        scr = torch.empty(int(max(lib.nq_weight_grad_scratch_floats(R, K, ncol), lib.nq_weight_grad_scratch_floats(R, ncol, K))) + 64, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_weight_grad(_lib.ptr(h1), _lib.ptr(gw1), _lib.ptr(gW1), R, K, ncol, _lib.ptr(scr), _lib.stream_ptr()))
        gh2 = torch.empty_like(h2)
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(gw2), _lib.ptr(W2), _lib.ptr(gh2), R, ncol, K, 0, _lib.stream_ptr()))
        gW2 = torch.empty_like(W2)
        gb2 = torch.empty(ncol, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_weight_grad_bias(_lib.ptr(gw2), _lib.ptr(h2), _lib.ptr(gW2), _lib.ptr(gb2), R, ncol, K, _lib.ptr(scr), _lib.stream_ptr()))
        return gx, gh1, gW1, gh2, gW2, gb2, None


class _PairGatherAddFn(torch.autograd.Function):
    """out[r] = a[dst(r)] + b[src(r)] over the pair list; the reverse sums over each atom's own row / reverse slots (fixed order, no atomics)."""

    @staticmethod
    def forward(ctx, a, b, csr):
        ctx.csr = csr
        return a[csr.col.long()] + b[csr.own.long()]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        csr = ctx.csr
        g = _f32(g)
        W = g.shape[1]
        ga = torch.empty(csr.N, W, device=g.device, dtype=torch.float32)
        gb = torch.empty_like(ga)
        _lib.check(lib.nq_qh_pair_reduce(None, _lib.ptr(g), None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), csr.N, W, _lib.ptr(ga), _lib.stream_ptr()))
        _lib.check(lib.nq_qh_pair_reduce(_lib.ptr(g), None, None, _lib.ptr(csr.row_ptr), _lib.ptr(csr.rev), csr.N, W, _lib.ptr(gb), _lib.stream_ptr()))
        return ga, gb, None


class _NormCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32(x)
        rows, ncomp, Cc = x.shape
        lmax = int(round(math.sqrt(ncomp))) - 1
        out = torch.empty(rows, (lmax + 1) * Cc, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_normcat(_lib.ptr(x), None, rows, Cc, lmax, _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        rows, ncomp, Cc = x.shape
        lmax = int(round(math.sqrt(ncomp))) - 1
        gx = torch.empty_like(x)
        _lib.check(lib.nq_qh_normcat(_lib.ptr(x), _lib.ptr(_f32(g)), rows, Cc, lmax, _lib.ptr(gx), _lib.stream_ptr()))
        return gx


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gates):
        lib = _lib.load()
        x, gates = _f32(x), _f32(gates)
        rows, ncomp, Cc = x.shape
        lmax = int(round(math.sqrt(ncomp))) - 1
        y = torch.empty_like(x)
        _lib.check(lib.nq_qh_gate(_lib.ptr(x), _lib.ptr(gates), None, rows, Cc, lmax, _lib.ptr(y), None, None, _lib.stream_ptr()))
        ctx.save_for_backward(x, gates)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, gates = ctx.saved_tensors
        rows, ncomp, Cc = x.shape
        lmax = int(round(math.sqrt(ncomp))) - 1
        gx, gg = torch.empty_like(x), torch.empty_like(gates)
        _lib.check(lib.nq_qh_gate(_lib.ptr(x), _lib.ptr(gates), _lib.ptr(_f32(g)), rows, Cc, lmax, None, _lib.ptr(gx), _lib.ptr(gg), _lib.stream_ptr()))
        return gx, gg


class _ExpansionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, spec):
        lib = _lib.load()
        x, W, b = _f32(x), _f32(W), _f32(b)
        shells, w3j, S = spec
        R, _, Cb = x.shape
        out = torch.empty(R, S, S, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_expansion_forward(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), R, Cb, shells, W.shape[1], b.shape[1], _lib.ptr(w3j), _lib.ptr(out),
                                               _lib.stream_ptr()))
        ctx.save_for_backward(x, W)
        ctx.meta = (spec, b.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        (shells, w3j, S), nb = ctx.meta
        R, _, Cb = x.shape
        gx, gW = torch.empty_like(x), torch.empty_like(W)
        gb = torch.empty(R, nb, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_qh_expansion_backward(_lib.ptr(x), _lib.ptr(W), _lib.ptr(_f32(g)), R, Cb, shells, W.shape[1], nb, _lib.ptr(w3j), _lib.ptr(gx), _lib.ptr(gW),
                                                _lib.ptr(gb), _lib.stream_ptr()))
        return gx, gW, gb, None


class _ToDenseFn(torch.autograd.Function):
    """packed diagonal blocks -> the block_diag matrix the reference returns; reverse = gather of the diagonal blocks."""

    @staticmethod
    def forward(ctx, packed, asm, plan):
        ctx.meta = (asm, plan)
        return asm.to_dense(plan, packed.detach())

    @staticmethod
    def backward(ctx, g):
        asm, plan = ctx.meta
        return asm.from_dense(plan, g), None, None


class _ScaledTransposeFn(torch.autograd.Function):
    """e3nn's flat ``o3.Linear`` weight [(lmax + 1) * c_in * c_out] -> the engine's stacked [lmax + 1, c_out, c_in] * scale: ONE elementwise launch each way
    (written as ``(w.view(..).transpose(1, 2) * scale).contiguous()`` + per-order views it was 2 launches forward and order + 3 backward, per layer)."""

    @staticmethod
    def forward(ctx, weight, n, c_in, c_out, scale):
        W = torch.empty(n, c_out, c_in, device=weight.device, dtype=torch.float32)
        torch.mul(weight.detach().view(n, c_in, c_out).transpose(1, 2), scale, out=W)
        ctx.meta = (n, c_in, c_out, scale)
        return W

    @staticmethod
    def backward(ctx, g):
        n, c_in, c_out, scale = ctx.meta
        gw = torch.empty(n * c_in * c_out, device=g.device, dtype=torch.float32)
        torch.mul(g.transpose(1, 2), scale, out=gw.view(n, c_in, c_out))
        return gw, None, None, None, None


# ---- e3nn-shaped parameter containers ------------------------------------------------------------------------------------------------------------
class _TensorProductShell(nn.Module):
    """Holds what an e3nn.o3.TensorProduct contributes to a state_dict: ``weight`` (a parameter when internal, else an empty buffer) and the
    ``output_mask`` buffer.  The arithmetic is in the HIP kernels."""

    def __init__(self, out_dim, weight_numel=0):
        super().__init__()
        if weight_numel:
            self.weight = nn.Parameter(torch.randn(weight_numel))
        else:
            self.register_buffer("weight", torch.Tensor())
        self.register_buffer("output_mask", torch.ones(out_dim))
        self.weight_numel = weight_numel


class _NormShell(nn.Module):          # e3nn.o3.Norm = a TensorProduct wrapped as ``tp``
    def __init__(self, num_mul):
        super().__init__()
        self.tp = _TensorProductShell(num_mul)


class InnerProduct(_NormShell):        # layers.py:277-294 (the arithmetic is fused into nq_qh_invariants_*)
    pass


class O3Linear(nn.Module):
    """e3nn.o3.Linear between ``c_in x (0 + .. + lmax)`` and ``c_out x (0 + .. + lmax)``: per l a matrix W_l [c_in, c_out] (flattened, in
    l order) applied as x_l W_l / sqrt(c_in); bias on the l = 0 output only (``biases=True``).  ``lmax_in = 0`` feeds scalars only."""

    def __init__(self, c_in, c_out, lmax=LMAX, biases=False):
        super().__init__()
        self.c_in, self.c_out, self.lmax = c_in, c_out, lmax
        self.weight = nn.Parameter(torch.randn((lmax + 1) * c_in * c_out))
        if biases:
            self.bias = nn.Parameter(torch.zeros(c_out))
        else:
            self.register_buffer("bias", torch.Tensor())
        self.register_buffer("output_mask", torch.ones((lmax + 1) ** 2 * c_out))
        self.has_bias = biases

    def forward(self, x):
        W = _ScaledTransposeFn.apply(self.weight, self.lmax + 1, self.c_in, self.c_out, 1.0 / math.sqrt(self.c_in))
        return _SphLinearFn.apply(x, self.bias if self.has_bias else None, W)


class _FCLayer(nn.Module):
    def __init__(self, h_in, h_out):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(h_in, h_out))


class FullyConnectedNet(nn.Module):
    """e3nn.nn.FullyConnectedNet([h0, h1, h2], act): act(x W0 / sqrt(h0)) * cst, then W1 / sqrt(h1); no biases, no activation on the output."""

    def __init__(self, hs, act="ssp"):
        super().__init__()
        assert len(hs) == 3
        self.hs = list(hs)
        self.layer0, self.layer1 = _FCLayer(hs[0], hs[1]), _FCLayer(hs[1], hs[2])
        self.kind = {"silu": 0, "ssp": 1}[act]
        self.cst = normalize2mom_constant(act)
        self._cs, self._cs_key = None, None

    def forward(self, x, col_scale=None):
        h, W1 = self.hidden_and_weight(x, col_scale)
        return _MatmulFn.apply(h, W1)

    def hidden_and_weight(self, x, col_scale=None):
        """(h, W1) with forward(x) = h @ W1: the hidden activations [rows, h1] and the effective last weight matrix [h1, h2] (path constants and
        1 / sqrt(h1) folded in) -- what the generator-fused tensor product (csrc/qhgen.hip) takes instead of the materialised product."""
        h = _MatmulFn.apply(x, self.layer0.weight * (1.0 / math.sqrt(self.hs[0])))
        h = _ActFn.apply(h, self.kind, self.cst)
        c1 = 1.0 / math.sqrt(self.hs[1])
        if col_scale is not None:
            # the path constants are a buffer of the caller: their product with 1 / sqrt(h1) is taken once.  The cached tensor must be an ordinary one: a product
            # taken under torch.inference_mode() (Lightning's sanity validation runs first) is an inference tensor that a later TRAINING forward cannot save
            # for backward, so the cache is rebuilt when it is one and grad mode is on
            key = (col_scale.data_ptr(), col_scale._version, str(col_scale.device))
            stale = self._cs is not None and self._cs.is_inference() and not torch.is_inference_mode_enabled()
            if self._cs_key != key or stale:
                self._cs, self._cs_key = (col_scale.detach() * c1), key
            W1 = self.layer1.weight * self._cs
        else:
            W1 = self.layer1.weight * c1
        return h, W1


def _mlp(seq: nn.Sequential, x):
    """nn.Sequential(Linear, SiLU, Linear) on the MFMA GEMMs."""
    h = _LinearBiasFn.apply(x, seq[0].weight, seq[0].bias, True)
    return _LinearBiasFn.apply(h, seq[2].weight, seq[2].bias, False)


class NormGate(nn.Module):
    """layers.py:123-147 for ``c x (0 + .. + 4)``: gates = MLP([x_0 | ||x_l||]); scalars are replaced by their gates, l > 0 multiplied."""

    def __init__(self, c, lmax=LMAX):
        super().__init__()
        num_mul = (lmax + 1) * c
        self.norm = _NormShell(num_mul)
        self.mul = _TensorProductShell(((lmax + 1) ** 2 - 1) * c)
        self.fc = nn.Sequential(nn.Linear(num_mul, num_mul), nn.SiLU(), nn.Linear(num_mul, num_mul))
        self.num_mul, self.num_mul_wo_0 = num_mul, num_mul - c

    def forward(self, x):
        gates = _mlp(self.fc, _NormCatFn.apply(x))
        return _GateFn.apply(x, gates)


class ConvLayer(nn.Module):
    def __init__(self, c_in, c, first_layer, edge_attr_dim, invariant_neurons=32, use_norm_gate=True):
        super().__init__()
        self.c, self.first, self.use_norm_gate = c, first_layer, use_norm_gate
        self._paths = conv_paths(first_layer)
        self._pidx = _path_index(self._paths)
        wn = len(self._paths) * c
        self.tp_node = _TensorProductShell(NCOMP * c)
        self.fc_node = FullyConnectedNet([edge_attr_dim, invariant_neurons, wn], "ssp")
        num_mul = c if first_layer else (LMAX + 1) * c
        self.layer_l0 = FullyConnectedNet([num_mul + c, invariant_neurons, wn], "ssp")
        self.linear_out = O3Linear(c, c, biases=True)
        if use_norm_gate:
            self.norm_gate = NormGate(c)
            self.linear_node = O3Linear(c, c, biases=True)
            self.linear_node_pre = O3Linear(c, c, biases=True)
        self.inner_product = InnerProduct(num_mul)
        self.register_buffer("_pc", torch.tensor(path_constants(self._paths), dtype=torch.float32).repeat_interleave(c), persistent=False)

    def forward(self, g, x):
        if self.use_norm_gate:
            pre_x = self.linear_node_pre(x)
            s0 = _InvFn.apply(pre_x, g.conv, False)
            x = self.linear_node(self.norm_gate(x))
        else:
            s0 = _InvFn.apply(x, g.conv, False)
        w1 = self.fc_node(g.edge_attr, self._pc)
        w2 = self.layer_l0(s0)
        out = _ConvFn.apply(x, w1, w2, g.conv, g.edge_sh, 2 if self.first else 1, not self.first)
        return self.linear_out(out)


class ConvNetLayer(nn.Module):
    def __init__(self, c, first_layer, edge_attr_dim, resnet=True, use_norm_gate=True):
        super().__init__()
        self.resnet = resnet and not first_layer
        self.conv = ConvLayer(c, c, first_layer, edge_attr_dim, 32, use_norm_gate)

    def forward(self, g, x):
        y = self.conv(g, x)
        return x + y if self.resnet else y


class SelfNetLayer(nn.Module):
    def __init__(self, c, resnet=True):
        super().__init__()
        self.c, self.resnet = c, resnet
        self._paths = list(cg.ALL_PATHS)
        self._pidx = _path_index(self._paths)
        self.linear_node_1 = O3Linear(c, c, biases=True)
        self.linear_node_2 = O3Linear(c, c, biases=True)
        self.tp = _TensorProductShell(NCOMP * c, len(self._paths) * c)
        self.norm_gate, self.norm_gate_1, self.norm_gate_2 = NormGate(c), NormGate(c), NormGate(c)
        self.linear_node_3 = O3Linear(c, c, biases=True)
        self.register_buffer("_pc", torch.tensor(path_constants(self._paths), dtype=torch.float32).view(-1, 1), persistent=False)

    def forward(self, g, x, old_fii):
        old_x = x
        xl = self.linear_node_1(self.norm_gate_1(x))
        xr = self.linear_node_2(self.norm_gate_2(x))
        coeff = (self.tp.weight.view(len(self._paths), self.c) * self._pc).contiguous()
        x = _MixFn.apply(xl, xr, coeff, None, (LMAX, LMAX, LMAX, self._pidx, False, 0, False))
        if self.resnet:
            x = x + old_x
        x = self.linear_node_3(self.norm_gate(x))
        if self.resnet and old_fii is not None:
            x = old_fii + x
        return x


class PairNetLayer(nn.Module):
    def __init__(self, c, edge_attr_dim, invariant_neurons, resnet=True):
        super().__init__()
        self.c, self.resnet = c, resnet
        self._paths = list(cg.ALL_PATHS)
        self._pidx = _path_index(self._paths)
        wn = len(self._paths) * c
        self.linear_node_pair = O3Linear(c, c, biases=True)       # registered first: the reference assigns it twice (layers.py:391, :440)
        self.linear_node_pair_n = O3Linear(c, c, biases=True)
        self.linear_node_pair_inner = O3Linear(c, c, biases=True)
        self.tp_node_pair = _TensorProductShell(NCOMP * c)
        self.fc_node_pair = FullyConnectedNet([edge_attr_dim, invariant_neurons, wn], "ssp")
        self.norm_gate = NormGate(c)
        self.inner_product = InnerProduct((LMAX + 1) * c)
        self.norm = _NormShell((LMAX + 1) * c)
        self.norm_gate_pre = NormGate(c)
        self.fc = nn.Sequential(nn.Linear(c + (LMAX + 1) * c, c), nn.SiLU(), nn.Linear(c, wn))
        self.register_buffer("_pc", torch.tensor(path_constants(self._paths), dtype=torch.float32).repeat_interleave(c), persistent=False)

    def forward(self, g, node_attr, node_pair_attr=None):
        a0 = self.linear_node_pair_inner(node_attr)
        s0 = _InvFn.apply(a0, g.full, True)
        xn = self.linear_node_pair_n(self.norm_gate_pre(node_attr))
        if PAIR_GENERATOR_FUSION and self.c % 16 == 0 and self.fc_node_pair.hs[1] == self.fc[0].out_features and self.fc[0].out_features in (32, 64, 128):
            h1, W1 = self.fc_node_pair.hidden_and_weight(g.full_edge_attr, self._pc)
            h2 = _LinearBiasFn.apply(s0, self.fc[0].weight, self.fc[0].bias, True)
            node_pair = _PairMixGenFn.apply(xn, h1, W1, h2, self.fc[2].weight, self.fc[2].bias, g.full)
        else:
            w1 = self.fc_node_pair(g.full_edge_attr, self._pc)
            w2 = _mlp(self.fc, s0)
            node_pair = _PairMixFn.apply(xn, w1, w2, g.full)
        node_pair = self.linear_node_pair(self.norm_gate(node_pair))
        if self.resnet and node_pair_attr is not None:
            node_pair = node_pair + node_pair_attr
        return node_pair


class Expansion(nn.Module):
    """layers.py:585-682 for ``cb x (0e + .. + 4e) -> (n_s x 0e + n_p x 1e + n_d x 2e)^2``; per-row path weights and biases come from the
    caller (QHNet's fc_ii / fc_ij heads); ``weights`` is the reference's unused parameter."""

    def __init__(self, cb, n_s, n_p, n_d):
        super().__init__()
        self.cb, self.counts = cb, (n_s, n_p, n_d)
        self.S = n_s + 3 * n_p + 5 * n_d
        ins = [(li, l1, l2) for li in range(LMAX + 1) for l1 in range(3) for l2 in range(3) if abs(l1 - l2) <= li <= l1 + l2]
        self.instructions = ins
        self.num_path_weight = sum(cb * self.counts[l1] * self.counts[l2] for _, l1, l2 in ins)
        self.num_bias = sum(self.counts[l1] * self.counts[l2] for li, l1, l2 in ins if li == 0)
        self.num_weights = self.num_path_weight + self.num_bias
        self.weights = nn.Parameter(torch.rand(self.num_weights))
        w3j = torch.zeros(len(ins), 5, 5, 9, dtype=torch.float32)
        for k, (li, l1, l2) in enumerate(ins):
            w3j[k, :2 * l1 + 1, :2 * l2 + 1, :2 * li + 1] = torch.tensor(cg.wigner_3j_e3nn(l1, l2, li), dtype=torch.float32)   # o3.wigner_3j(ins[1], ins[2], ins[0]), layers.py:617
        self.register_buffer("_w3j", w3j, persistent=False)
        self._shells = (C.c_int32 * 3)(n_s, n_p, n_d)

    def forward(self, x_in, weights=None, bias_weights=None):
        if weights is None or bias_weights is None:
            raise NotImplementedError("Expansion is built for per-row weights and biases (the only way QHNet calls it, qhnet.py:222-232)")
        return _ExpansionFn.apply(x_in, weights, bias_weights, (self._shells, self._w3j, self.S))


# ---- the network ---------------------------------------------------------------------------------------------------------------------------------
class QHNet(nn.Module):
    """Same constructor as the reference (qhnet.py:30-41; config/model/qhnet.yaml:7-22).  ``forward(data)`` takes a PyG-style batch
    (``pos, z, batch, ptr``) on the GPU and returns the block-diagonal Hamiltonian [sum M_b, sum M_b] (or the blocks with
    ``keep_blocks=True``); ``forward(data, packed=True)`` returns the diagonal blocks packed molecule after molecule (what the loss kernel
    consumes -- the dense matrix of a 64-molecule batch would be 2.8 GB of zeros)."""

    def __init__(self, in_node_features=1, sh_lmax=4, hidden_size=128, bottle_hidden_size=32, num_gnn_layers=5, max_radius=12, num_nodes=10,
                 radius_embed_dim=32, orbitals: Optional[Dict[int, List[int]]] = None):
        super().__init__()
        if sh_lmax != LMAX:
            raise NotImplementedError("nabladft_amd.qhnet: sh_lmax = 4 (the hidden irreps of the reference are fixed at l <= 4, qhnet.py:53-58)")
        if orbitals is None:
            raise ValueError("QHNet needs the `orbitals` table (config/model/qhnet.yaml:14-22)")
        self.order, self.hs, self.hbs = sh_lmax, hidden_size, bottle_hidden_size
        self.radius_embed_dim, self.max_radius, self.num_gnn_layers = radius_embed_dim, max_radius, num_gnn_layers
        self.node_embedding = nn.Embedding(num_nodes, self.hs)
        self.distance_expansion = ExponentialBernsteinRadialBasisFunctions(self.radius_embed_dim, self.max_radius)
        self.num_fc_layer = 1
        self.orbitals = {int(k): list(v) for k, v in orbitals.items()}
        self._asm = BlockAssembler(self.orbitals)
        self.orbital_mask = {k: torch.tensor(v) for k, v in self._asm.masks.items()}
        max_s, max_p, max_d = self._asm.s_max, self._asm.p_max, self._asm.d_max
        self.e3_gnn_layer, self.e3_gnn_node_pair_layer, self.e3_gnn_node_layer = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.udpate_layer = nn.ModuleList()
        self.start_layer = 2
        for i in range(self.num_gnn_layers):
            self.e3_gnn_layer.append(ConvNetLayer(self.hs, i == 0, self.radius_embed_dim, resnet=True, use_norm_gate=i != 0))
            if i > self.start_layer:
                self.e3_gnn_node_layer.append(SelfNetLayer(self.hs, resnet=True))
                self.e3_gnn_node_pair_layer.append(PairNetLayer(self.hs, self.radius_embed_dim, self.hs, resnet=True))
        self.expand_ii, self.expand_ij = nn.ModuleDict(), nn.ModuleDict()
        self.fc_ii, self.fc_ij, self.fc_ii_bias, self.fc_ij_bias = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        for name in ["hamiltonian"]:
            self.expand_ii[name] = Expansion(self.hbs, max_s, max_p, max_d)
            self.fc_ii[name] = nn.Sequential(nn.Linear(self.hs, self.hs), nn.SiLU(), nn.Linear(self.hs, self.expand_ii[name].num_path_weight))
            self.fc_ii_bias[name] = nn.Sequential(nn.Linear(self.hs, self.hs), nn.SiLU(), nn.Linear(self.hs, self.expand_ii[name].num_bias))
            self.expand_ij[name] = Expansion(self.hbs, max_s, max_p, max_d)
            self.fc_ij[name] = nn.Sequential(nn.Linear(self.hs * 2, self.hs), nn.SiLU(), nn.Linear(self.hs, self.expand_ij[name].num_path_weight))
            self.fc_ij_bias[name] = nn.Sequential(nn.Linear(self.hs * 2, self.hs), nn.SiLU(), nn.Linear(self.hs, self.expand_ij[name].num_bias))
        self.output_ii = O3Linear(self.hs, self.hbs)
        self.output_ij = O3Linear(self.hs, self.hbs)

    def set(self):
        for key in self.orbital_mask.keys():
            self.orbital_mask[key] = self.orbital_mask[key].to(self.device)

    def get_number_of_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    @property
    def device(self):
        return next(self.parameters()).device

    # -- graph ---------------------------------------------------------------------------------------------------------------------------------
    def _csr(self, data, max_radius):
        ptr = getattr(data, "ptr", None)
        nl = build_neighbor_list(data.pos, data.batch, None, float(max_radius), 1 << 30, ptr=ptr)
        return _Csr(nl)

    def _edge_features(self, csr):
        lib = _lib.load()
        d = csr.geom[:, 3].contiguous()
        rbf = self.distance_expansion(d.unsqueeze(-1))
        u = csr.geom[:, :3].contiguous()
        sh = torch.empty(csr.R, NCOMP, device=u.device, dtype=torch.float32)
        _lib.check(lib.nq_sph_harm(_lib.ptr(u), csr.R, LMAX, _lib.ptr(sh), _lib.stream_ptr()))
        return rbf, sh

    def build_graph(self, data, max_radius, edge_index=None):
        """Same return tuple as the reference (qhnet.py:254-291): (z, edge_index [2, E] with row 0 = dst, row 1 = src, rbf [E, K], edge_sh [E, 25],
        transpose index).  The transpose index is only meaningful on the full graph, as in the reference."""
        _require_gpu(data.pos)
        csr = self._csr(data, max_radius)
        rbf, sh = self._edge_features(csr)
        ptr = data.ptr.to(data.pos.device)
        return data.z.squeeze(), csr.edge_index, rbf, sh, transpose_index(ptr)

    def _graphs(self, data):
        """The two neighbour structures (index arrays + geometry; no learnable parameter involved)."""
        g = _Graphs()
        g.conv = self._csr(data, self.max_radius)
        g.full = self._csr(data, 10000)
        return g

    def _features(self, g):
        """Radial basis (learnable exponent: part of every step's autograd graph, never cached across steps) and harmonics of both edge sets."""
        g.edge_attr, g.edge_sh = self._edge_features(g.conv)
        g.full_edge_attr, g.full_edge_sh = self._edge_features(g.full)

    # -- forward -------------------------------------------------------------------------------------------------------------------------------
    def prepare(self, data):
        """Everything of a batch that is built with host reads: the two neighbour structures (edge counts come back to the host to size the per-edge tensors)
        and the matrix-assembly tables.  Pass the result as ``data.prepared`` and ``forward`` issues no host synchronisation at all -- a training step on
        that batch can then be captured into a HIP graph (trainer.GraphedStep; the edge sets depend on the positions, so a prepared batch is tied to its
        geometry, not only to its composition)."""
        _require_gpu(data.pos)
        g = self._graphs(data)
        z = data.z.squeeze().long()
        g.plan = self._asm.plan(z, data.ptr.to(z.device), g.full.edge_index)
        g.ptr = data.ptr.to(data.pos.device)
        g.transpose = transpose_index(g.ptr)
        g.n_atoms = int(data.pos.shape[0])
        g.geometry_key = _lib.geometry_key(data)        # checked by forward: the edge sets depend on the positions
        return g

    def forward(self, data, keep_blocks=False, packed=False):
        _require_gpu(data.pos)
        g = getattr(data, "prepared", None)
        if g is None:
            g = self.prepare(data)
        elif g.n_atoms != int(data.pos.shape[0]):
            raise ValueError("data.prepared belongs to another batch")
        else:
            _lib.check_prepared(g, data)
        g = copy.copy(g)                                # per-call view: the autograd-tracked edge features below never land on the cached prepare() result
        self._features(g)
        z = data.z.squeeze().long()
        node_attr = self.node_embedding(z)
        # The reference leaves these on the batch (qhnet.py:189-208) and its layers read them back; here the layers read `g`, and the copies left on the batch
        # are DETACHED: a tracked tensor parked on `data` keeps the previous step's autograd graph -- and with it the AccumulateGrad nodes of the embedding and
        # of the radial exponent, which belong to the stream that first created them -- alive into the next step.  Captured into a HIP graph, that step then
        # makes the legacy default stream wait on the capture stream and hipStreamEndCapture crashes (rounds 3-5: scripts/debug_qhnet_capture.py).
        data.node_attr, data.edge_index, data.edge_attr, data.edge_sh = node_attr.detach(), g.conv.edge_index, g.edge_attr.detach(), g.edge_sh
        data.full_edge_index, data.full_edge_attr, data.full_edge_sh = g.full.edge_index, g.full_edge_attr.detach(), g.full_edge_sh
        x = node_attr.view(-1, 1, self.hs)
        fii = fij = None
        for layer_idx, layer in enumerate(self.e3_gnn_layer):
            x = layer(g, x)
            if layer_idx > self.start_layer:
                k = layer_idx - self.start_layer - 1
                fii = self.e3_gnn_node_layer[k](g, x, fii)
                fij = self.e3_gnn_node_pair_layer[k](g, x, fij)
        fii, fij = self.output_ii(fii), self.output_ij(fij)
        name = "hamiltonian"
        diag = self.expand_ii[name](fii, _mlp(self.fc_ii[name], node_attr), _mlp(self.fc_ii_bias[name], node_attr))
        nondiag = self.expand_ij[name](fij, self._pair_head(self.fc_ij[name], node_attr, g.full), self._pair_head(self.fc_ij_bias[name], node_attr, g.full))
        if keep_blocks:
            t = g.transpose
            return {"hamiltonian_diagonal_blocks": diag + diag.transpose(-1, -2),
                    "hamiltonian_non_diagonal_blocks": nondiag + nondiag[t].transpose(-1, -2)}
        plan = g.plan
        H = self._asm.assemble(plan, diag, nondiag, symmetrize=True)          # build_final_matrix + H + H^T (qhnet.py:234-237)
        self.last_plan = plan
        if packed:
            return H
        return _ToDenseFn.apply(H, self._asm, plan)

    def _pair_head(self, seq, node_attr, csr):
        """seq(cat([node_attr[dst], node_attr[src]])) (qhnet.py:226-232) with the first Linear applied per atom before the gather."""
        W0 = seq[0].weight
        c = self.hs
        a = _LinearFn.apply(node_attr, W0[:, :c].contiguous())
        b = _LinearFn.apply(node_attr, W0[:, c:].contiguous())
        pre = _PairGatherAddFn.apply(a, b + seq[0].bias, csr)
        h = _ActFn.apply(pre, 0, 1.0)
        return _LinearBiasFn.apply(h, seq[2].weight, seq[2].bias, False)

    def build_final_matrix(self, data, diagonal_matrix, non_diagonal_matrix):
        return self._asm.build_final_matrix(data, diagonal_matrix, non_diagonal_matrix)
