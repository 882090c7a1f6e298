"""PhiSNet's SO(3) mixing modules on MI355X: same constructors, parameter names and list-of-orders
call convention as
  nablaDFT.phisnet.nn.modules.pair_mixing.PairMixing   (pair_mixing.py:10-69)
  nablaDFT.phisnet.nn.modules.self_mixing.SelfMixing   (self_mixing.py:10-83)
so they drop into InteractionBlock / ResidualBlock unchanged (state_dict compatible).  The Clebsch-Gordan contraction runs in one
register-resident kernel per call (csrc/so3.hip), the distance-dependent coefficients of PairMixing in one MFMA GEMM over all paths.
``clebsch_gordan`` is the model's own provider (``ClebschGordan()`` module or any callable (l1, l2, L) -> tensor); only its per-path
signs are used -- the values are checked against tensors computed from scratch (nabladft_amd/cg.py).  GPU only, orders <= 4.
"""
import ctypes as C
from typing import List

import numpy as np
import torch
from torch import nn

from . import _lib, cg


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T through the engine's fp32 MFMA GEMMs (nq_linear_*)."""

    @staticmethod
    def forward(ctx, x, W):
        lib = _lib.load()
        x = x.to(torch.float32).contiguous()
        W = W.to(torch.float32).contiguous()
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_forward(_lib.ptr(x), _lib.ptr(W), None, _lib.ptr(y), None, M, N, K, _lib.stream_ptr()))
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        M, K = x.shape
        N = W.shape[0]
        gx = torch.empty_like(x)
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, 0, _lib.stream_ptr()))
        scr = torch.empty(int(lib.nq_weight_grad_scratch_floats(M, N, K)) + 64, device=x.device, dtype=torch.float32)
        gW = torch.empty_like(W)
        _lib.check(lib.nq_linear_weight_grad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), _lib.stream_ptr()))
        return gx, gW


class _MixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, coeff, keep, spec):
        lib = _lib.load()
        o1, o2, oy, pidx, per_row, keep_orders, same = spec
        rows, _, F = x1.shape
        y = torch.empty(rows, (oy + 1) ** 2, F, device=x1.device, dtype=torch.float32)
        stride = coeff.shape[-2] * F if per_row else 0
        _lib.check(lib.nq_so3_mix_forward(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(coeff), _lib.ptr(keep), rows, F, o1, o2, oy, pidx, stride, keep_orders,
                                          _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x1, x2, coeff, keep if keep is not None else x1.new_zeros(0))
        ctx.spec = spec
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x1, x2, coeff, keep = ctx.saved_tensors
        o1, o2, oy, pidx, per_row, keep_orders, same = ctx.spec
        keep = keep if keep.numel() else None
        rows, _, F = x1.shape
        gy = gy.to(torch.float32).contiguous()
        n_en = coeff.shape[-2]
        gx1, gx2 = torch.empty_like(x1), torch.empty_like(x2)
        gk_rows = torch.empty(rows, keep_orders, F, device=x1.device, dtype=torch.float32) if keep is not None else None
        if not per_row and 256 % F == 0 and rows > 0:
            # shared coefficients: dL/dc is reduced over the rows inside the kernel (per-workgroup partials, summed here in block order)
            nblk = int(lib.nq_so3_mix_partial_blocks(rows, F))
            part = torch.empty(nblk, n_en, F, device=x1.device, dtype=torch.float32)
            _lib.check(lib.nq_so3_mix_backward_shared(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(coeff), _lib.ptr(keep), _lib.ptr(gy), rows, F, o1, o2, oy, pidx, keep_orders,
                                                      _lib.ptr(gx1), _lib.ptr(gx2), _lib.ptr(part), _lib.ptr(gk_rows), _lib.stream_ptr()))
            gc = part.sum(0) if nblk > 1 else part[0]
        else:
            gc_rows = torch.empty(rows, n_en, F, device=x1.device, dtype=torch.float32)
            stride = n_en * F if per_row else 0
            _lib.check(lib.nq_so3_mix_backward(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(coeff), _lib.ptr(keep), _lib.ptr(gy), rows, F, o1, o2, oy, pidx, stride,
                                               keep_orders, _lib.ptr(gx1), _lib.ptr(gx2), _lib.ptr(gc_rows), _lib.ptr(gk_rows), _lib.stream_ptr()))
            gc = gc_rows if per_row else gc_rows.sum(0)
        gk = None if keep is None else gk_rows.sum(0)
        return gx1, gx2, gc, gk, None


class _MixFlatFn(torch.autograd.Function):
    """SelfMixing whose coefficients live side by side in a flat parameter buffer (trainer.FlatParameters.attach): they are read as ONE block
    [n_mix + n_keep, F] -- a leaf tensor aliasing that stretch of the flat buffer, its ``.grad`` aliasing the flat gradient -- and their gradient is
    returned to autograd as one block (one accumulation instead of a stack + one per tensor; works under torch.autograd.grad / checkpointing)."""

    @staticmethod
    def forward(ctx, x, block, mod):
        lib = _lib.load()
        n_mix, n_keep = mod._flat[2], mod._flat[3]
        F = mod.num_features
        blk = block.detach()
        coeff = (blk[:n_mix] * mod._sign).contiguous() if n_mix else x.new_zeros(1, F)
        keep = blk[n_mix:]
        rows = x.shape[0]
        y = torch.empty(rows, (mod.order_out + 1) ** 2, F, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_so3_mix_forward(_lib.ptr(x), _lib.ptr(x), _lib.ptr(coeff), _lib.ptr(keep), rows, F, mod.order_in, mod.order_in, mod.order_out,
                                          mod._pidx, 0, n_keep, _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x, coeff, keep)
        ctx.mod = mod
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, coeff, keep = ctx.saved_tensors
        mod = ctx.mod
        n_mix, n_keep = mod._flat[2], mod._flat[3]
        rows, _, F = x.shape
        gy = gy.to(torch.float32).contiguous()
        n_en = coeff.shape[0]
        gx1, gx2 = torch.empty_like(x), torch.empty_like(x)
        gk_c = torch.empty(rows, n_keep, F, device=x.device, dtype=torch.float32)
        if 256 % F == 0 and rows > 0:
            nblk = int(lib.nq_so3_mix_partial_blocks(rows, F))
            part = torch.empty(nblk, n_en, F, device=x.device, dtype=torch.float32)
            _lib.check(lib.nq_so3_mix_backward_shared(_lib.ptr(x), _lib.ptr(x), _lib.ptr(coeff), _lib.ptr(keep), _lib.ptr(gy), rows, F, mod.order_in, mod.order_in,
                                                      mod.order_out, mod._pidx, n_keep, _lib.ptr(gx1), _lib.ptr(gx2), _lib.ptr(part), _lib.ptr(gk_c), _lib.stream_ptr()))
            gc_c = part
        else:
            gc_c = torch.empty(rows, n_en, F, device=x.device, dtype=torch.float32)
            _lib.check(lib.nq_so3_mix_backward(_lib.ptr(x), _lib.ptr(x), _lib.ptr(coeff), _lib.ptr(keep), _lib.ptr(gy), rows, F, mod.order_in, mod.order_in,
                                               mod.order_out, mod._pidx, 0, n_keep, _lib.ptr(gx1), _lib.ptr(gx2), _lib.ptr(gc_c), _lib.ptr(gk_c), _lib.stream_ptr()))
        parts = ([gc_c.sum(0) * mod._sign] if n_mix else []) + [gk_c.sum(0)]
        return gx1 + gx2, torch.cat(parts, dim=0), None


class PackedList:
    """The reference's list-of-orders representation ``[x_0 [..., 1, F], x_1 [..., 3, F], ...]`` backed by ONE packed tensor
    ``packed [rows, (order+1)^2, F]`` -- what the kernels read and write.  Elements are views created on access.  Modules hand the object
    on so that consecutive ops skip the concatenate / split copies; code that treats it as a list (len, indexing, iteration, ``list(xs)``,
    item assignment) still works -- replacing an element drops the packed shortcut."""

    def __init__(self, packed: torch.Tensor, order: int, lead, F: int):
        self.packed, self.order, self.lead, self.F = packed, order, tuple(lead), F
        self._items = None

    def _view(self, L):
        return self.packed[:, L * L:(L + 1) * (L + 1), :].reshape(*self.lead, 2 * L + 1, self.F)

    def _all(self):
        if self._items is None:
            self._items = [self._view(L) for L in range(self.order + 1)]
        return self._items

    def __len__(self):
        return self.order + 1

    def __getitem__(self, k):
        if self._items is None and isinstance(k, int):
            return self._view(k if k >= 0 else k + self.order + 1)
        return self._all()[k]

    def __iter__(self):
        return iter(self._all())

    def __setitem__(self, k, v):
        self._all()[k] = v
        self.packed = None

    def __add__(self, other):          # list concatenation, as for the reference's plain lists
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


def _pack(xs: List[torch.Tensor], order: int, F: int):
    if isinstance(xs, PackedList) and xs.packed is not None and xs.order == order and xs.F == F:
        return xs.packed, xs.lead
    lead = xs[0].shape[:-2]
    x = torch.cat([xs[l].reshape(-1, 2 * l + 1, F) for l in range(order + 1)], dim=1).to(torch.float32).contiguous()
    return x, lead


def _unpack(y: torch.Tensor, order: int, lead, F: int):
    return PackedList(y, order, lead, F)


def _path_index(enabled):
    arr = (C.c_int8 * len(cg.ALL_PATHS))(*([-1] * len(cg.ALL_PATHS)))
    for i, p in enumerate(enabled):
        arr[cg.PATH_ID[p]] = i
    return arr


def _signs(clebsch_gordan, which):
    table = lambda a, b, c: clebsch_gordan(a, b, c).detach().cpu().numpy()
    return cg.path_signs(table, which)


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("nabladft_amd.so3 runs on MI355X only (no CPU fallback): move the tensors to cuda")


class PairMixing(nn.Module):
    def __init__(self, order_in1, order_in2, order_out, num_basis_functions, num_features, clebsch_gordan):
        super().__init__()
        if max(order_in1, order_in2, order_out) > cg.LMAX:
            raise NotImplementedError(f"nabladft_amd.so3: orders up to {cg.LMAX} are built")
        self.order_in1, self.order_in2, self.order_out = order_in1, order_in2, order_out
        self.num_basis_functions, self.num_features = num_basis_functions, num_features
        self.clebsch_gordan = clebsch_gordan
        self._paths = cg.paths(order_in1, order_in2, order_out)
        for (l1, l2, L) in self._paths:
            self.add_module("coeff_{}_{}_{}".format(l1, l2, L), nn.Linear(num_basis_functions, num_features, bias=False))
        self.reset_parameters()
        self.register_buffer("_sign", torch.tensor(_signs(clebsch_gordan, self._paths), dtype=torch.float32), persistent=False)
        self._pidx = _path_index(self._paths)

    def reset_parameters(self):
        for (l1, l2, L) in self._paths:
            nn.init.orthogonal_(self.coeff(l1, l2, L).weight)

    def coeff(self, l1, l2, L):
        return getattr(self, "coeff_{}_{}_{}".format(l1, l2, L))

    def forward(self, x1s, x2s, rbf):
        _require_gpu(rbf)
        F = self.num_features
        x1, lead = _pack(x1s, self.order_in1, F)
        x2, _ = _pack(x2s, self.order_in2, F)
        rows = x1.shape[0]
        rbf2 = rbf.expand(*lead, 1, self.num_basis_functions).reshape(rows, self.num_basis_functions)
        W = torch.cat([self.coeff(*p).weight * s for p, s in zip(self._paths, self._sign)], dim=0)       # [n_paths * F, K], CG sign convention folded in
        coeff = _LinearFn.apply(rbf2, W).view(rows, len(self._paths), F)
        y = _MixFn.apply(x1, x2, coeff, None, (self.order_in1, self.order_in2, self.order_out, self._pidx, True, 0, False))
        return _unpack(y, self.order_out, lead, F)


class SelfMixing(nn.Module):
    def __init__(self, order_in, order_out, num_features, clebsch_gordan):
        super().__init__()
        if max(order_in, order_out) > cg.LMAX:
            raise NotImplementedError(f"nabladft_amd.so3: orders up to {cg.LMAX} are built")
        self.order_in, self.order_out, self.num_features = order_in, order_out, num_features
        self.clebsch_gordan = clebsch_gordan
        self._paths = [(l1, l2, L) for l1 in range(order_in + 1) for l2 in range(l1 + 1, order_in + 1)
                       for L in range(abs(l1 - l2), min(l1 + l2, order_out) + 1)]
        for (l1, l2, L) in self._paths:
            self.register_parameter("mixcoeff_{}_{}_{}".format(l1, l2, L), nn.Parameter(torch.Tensor(num_features)))
        self._keep = min(order_in, order_out) + 1
        for L in range(self._keep):
            self.register_parameter("keepcoeff_{}".format(L), nn.Parameter(torch.Tensor(num_features)))
        self.reset_parameters()
        self.register_buffer("_sign", torch.tensor(_signs(clebsch_gordan, self._paths), dtype=torch.float32).view(-1, 1), persistent=False)
        self._pidx = _path_index(self._paths)

    def reset_parameters(self):
        count = [0 for _ in range(self.order_out + 1)]
        for L in range(self._keep):
            count[L] += 1
        for (_, _, L) in self._paths:
            count[L] += 1
        for L in range(self._keep):
            nn.init.uniform_(self.keepcoeff(L), a=-np.sqrt(3 / count[L]), b=np.sqrt(3 / count[L]))
        for (l1, l2, L) in self._paths:
            nn.init.uniform_(self.mixcoeff(l1, l2, L), a=-np.sqrt(3 / count[L]), b=np.sqrt(3 / count[L]))

    def keepcoeff(self, L):
        return getattr(self, "keepcoeff_{}".format(L))

    def mixcoeff(self, l1, l2, L):
        return getattr(self, "mixcoeff_{}_{}_{}".format(l1, l2, L))

    _flat = None

    def _use_flat_parameters(self, flat):
        """trainer.FlatParameters.attach: switch to the block form if the coefficient vectors are contiguous in the flat buffer."""
        ps = [self.mixcoeff(*p) for p in self._paths] + [self.keepcoeff(L) for L in range(self._keep)]
        blk = flat.block_of(ps)
        if blk is None or blk[1] != len(ps) * self.num_features:
            return False
        n = len(ps)
        block = torch.nn.Parameter(flat.flat.data[blk[0]:blk[0] + blk[1]].view(n, self.num_features))      # a leaf aliasing the flat buffers; NOT registered
        block.grad = flat.flat.grad[blk[0]:blk[0] + blk[1]].view(n, self.num_features)
        self._flat = (flat, blk[0], len(self._paths), self._keep, block)
        return True

    def forward(self, xs):
        _require_gpu(xs[0])
        F = self.num_features
        x, lead = _pack(xs, self.order_in, F)
        if self._flat is not None and torch.is_grad_enabled() and x.requires_grad:
            return _unpack(_MixFlatFn.apply(x, self._flat[4], self), self.order_out, lead, F)
        if self._paths:
            coeff = torch.stack([self.mixcoeff(*p) for p in self._paths]) * self._sign                 # [n_paths, F]
        else:
            coeff = x.new_zeros(1, F)
        keep = torch.stack([self.keepcoeff(L) for L in range(self._keep)]).contiguous()
        y = _MixFn.apply(x, x, coeff.contiguous(), keep, (self.order_in, self.order_in, self.order_out, self._pidx, False, self._keep, True))
        return _unpack(y, self.order_out, lead, F)


# ---- geometry bases (radial part of a13) --------------------------------------------------------------------
class _SphHarmFn(torch.autograd.Function):
    """Y_0..Y_L of (unit) vectors with the adjoint w.r.t. the vectors (nq_sph_harm_backward): the harmonics are differentiated as polynomials of a free
    vector, as autograd does with the reference's closed forms; first order only (forces at inference, create_graph=False)."""

    @staticmethod
    def forward(ctx, u2, L):
        lib = _lib.load()
        out = torch.empty(u2.shape[0], (L + 1) ** 2, device=u2.device, dtype=torch.float32)
        _lib.check(lib.nq_sph_harm(_lib.ptr(u2), u2.shape[0], L, _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(u2)
        ctx.L = L
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        (u2,) = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        gu = torch.empty_like(u2)
        _lib.check(lib.nq_sph_harm_backward(_lib.ptr(u2), _lib.ptr(g), u2.shape[0], ctx.L, _lib.ptr(gu), _lib.stream_ptr()))
        return gu, None


def spherical_harmonics(L: int, u: torch.Tensor) -> List[torch.Tensor]:
    """List over l = 0..L of [..., 2l+1]: same call and conventions as phisnet/nn/spherical_harmonics/spherical_harmonics.py:28-64
    (unit vectors in, no 1/sqrt(4 pi), Condon-Shortley phase, m = -l..l).  L <= 4 on the GPU path.  Differentiable w.r.t. ``u`` (first order) when
    ``u`` requires a gradient (PhiSNet forces)."""
    _require_gpu(u)
    if L > cg.LMAX:
        raise NotImplementedError(f"nabladft_amd.so3.spherical_harmonics: orders up to {cg.LMAX} are built")
    lead = u.shape[:-1]
    if torch.is_grad_enabled() and u.requires_grad:
        out = _SphHarmFn.apply(u.to(torch.float32).reshape(-1, 3).contiguous(), L)
    else:
        lib = _lib.load()
        u2 = u.detach().to(torch.float32).reshape(-1, 3).contiguous()
        out = torch.empty(u2.shape[0], (L + 1) ** 2, device=u.device, dtype=torch.float32)
        _lib.check(lib.nq_sph_harm(_lib.ptr(u2), u2.shape[0], L, _lib.ptr(out), _lib.stream_ptr()))
    return [out[:, l * l:(l + 1) * (l + 1)].reshape(*lead, 2 * l + 1) for l in range(L + 1)]


class _BernsteinFn(torch.autograd.Function):
    """alpha = softplus(_alpha) stays on the device (a 1-element tensor handed to the kernels): no host read of the parameter per call."""

    @staticmethod
    def forward(ctx, r, raw_alpha, mod):
        lib = _lib.load()
        r2 = r.detach().to(torch.float32).reshape(-1).contiguous()
        alpha = torch.nn.functional.softplus(raw_alpha.detach().double()).to(torch.float32).reshape(1).contiguous()
        K = mod.num_basis_functions
        out = torch.empty(r2.shape[0], K, device=r.device, dtype=torch.float32)
        logc, n, v = mod._tables32(r.device)
        _lib.check(lib.nq_bernstein_rbf_dev(_lib.ptr(r2), r2.shape[0], K, _lib.ptr(alpha), mod._cutoff_f, _lib.ptr(logc), _lib.ptr(n), _lib.ptr(v), _lib.ptr(out),
                                            _lib.stream_ptr()))
        ctx.save_for_backward(r2, raw_alpha, alpha, logc, n, v)
        ctx.meta = (mod._cutoff_f, K, r.shape)
        return out.view(*r.shape[:-1], K) if r.shape[-1] == 1 else out.view(*r.shape, K)

    @staticmethod
    @torch.autograd.function.once_differentiable   # a second-order request through the radial path raises instead of returning wrong gradients
    def backward(ctx, g):
        lib = _lib.load()
        r2, raw_alpha, alpha, logc, n, v = ctx.saved_tensors
        cutoff, K, _ = ctx.meta
        g2 = g.to(torch.float32).reshape(-1, K).contiguous()
        rows = torch.empty(r2.shape[0], device=r2.device, dtype=torch.float32)
        _lib.check(lib.nq_bernstein_rbf_grad_alpha_dev(_lib.ptr(r2), _lib.ptr(g2), r2.shape[0], K, _lib.ptr(alpha), cutoff, _lib.ptr(logc), _lib.ptr(n), _lib.ptr(v),
                                                       _lib.ptr(rows), _lib.stream_ptr()))
        g_raw = rows.double().sum() * torch.sigmoid(raw_alpha.detach().double())       # d softplus
        gr = None
        if ctx.needs_input_grad[0]:                                                    # distances differentiated: forces = -dE/dR (first order only)
            gr = torch.empty(r2.shape[0], device=r2.device, dtype=torch.float32)
            _lib.check(lib.nq_bernstein_rbf_grad_r_dev(_lib.ptr(r2), _lib.ptr(g2), r2.shape[0], K, _lib.ptr(alpha), cutoff, _lib.ptr(logc), _lib.ptr(n), _lib.ptr(v),
                                                       _lib.ptr(gr), _lib.stream_ptr()))
            gr = gr.view(ctx.meta[2])
        return gr, g_raw.to(raw_alpha.dtype).reshape(raw_alpha.shape), None


class ExponentialBernsteinRadialBasisFunctions(nn.Module):
    """Same constructor, buffers (cutoff, logc, n, v) and parameter (_alpha) as the reference classes of that name
    (phisnet/nn/modules/exponential_bernstein_radial_basis_functions.py:13-41, qhnet/layers.py:92-120).  forward(r [..., 1]) -> [..., K].
    Gradient: w.r.t. ``_alpha``, and (first order) w.r.t. the distances when they require a gradient (PhiSNet forces)."""

    def __init__(self, num_basis_functions, cutoff, ini_alpha=0.5, dtype=torch.float32):
        super().__init__()
        self.num_basis_functions, self.ini_alpha = num_basis_functions, ini_alpha
        logfactorial = np.zeros(num_basis_functions)
        for i in range(2, num_basis_functions):
            logfactorial[i] = logfactorial[i - 1] + np.log(i)
        v = np.arange(0, num_basis_functions)
        n = (num_basis_functions - 1) - v
        logbinomial = logfactorial[-1] - logfactorial[v] - logfactorial[n]
        self.register_buffer("cutoff", torch.tensor(cutoff, dtype=dtype))
        self.register_buffer("logc", torch.tensor(logbinomial, dtype=dtype))
        self.register_buffer("n", torch.tensor(n, dtype=dtype))
        self.register_buffer("v", torch.tensor(v, dtype=dtype))
        self.register_parameter("_alpha", nn.Parameter(torch.tensor(1.0, dtype=dtype)))
        self.reset_parameters()

    def reset_parameters(self):
        x = torch.tensor(float(self.ini_alpha), dtype=torch.float64)
        nn.init.constant_(self._alpha, float(x + torch.log(-torch.expm1(-x))))          # softplus_inverse (phisnet/nn/functional.py)

    def _tables32(self, device):
        """float32 device copies of the buffers the kernels read, rebuilt when the module's buffers change: dtype / device moves (new storage) and
        in-place writes such as load_state_dict (same storage, new version counters)."""
        key = (self.logc.data_ptr(), str(device), self.logc._version, self.n._version, self.v._version, self.cutoff._version)
        if getattr(self, "_t32_key", None) != key:
            self._t32 = tuple(t.to(device=device, dtype=torch.float32).contiguous() for t in (self.logc, self.n, self.v))
            self._cutoff_f = float(self.cutoff)
            self._t32_key = key
        return self._t32

    def forward(self, r):
        _require_gpu(r)
        self._tables32(r.device)
        return _BernsteinFn.apply(r, self._alpha, self)


# ---- the other radial bases of PhiSNet ------------------------------------------------------------------------------
class _RadialFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, raw_alpha, mod):
        lib = _lib.load()
        r2 = r.detach().to(torch.float32).reshape(-1).contiguous()
        alpha = float(torch.nn.functional.softplus(raw_alpha.detach().double())) if mod._kind in (2, 3) else 0.0
        K = mod.num_basis_functions
        t = [b.to(device=r.device, dtype=torch.float32).contiguous() for b in mod._tables()]
        t += [None] * (3 - len(t))
        out = torch.empty(r2.shape[0], K, device=r.device, dtype=torch.float32)
        _lib.check(lib.nq_radial_basis(mod._kind, _lib.ptr(r2), r2.shape[0], K, alpha, float(mod.cutoff), float(mod._width()), _lib.ptr(t[0]), _lib.ptr(t[1]),
                                       _lib.ptr(t[2]), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(r2, raw_alpha, *[x for x in t if x is not None])
        ctx.meta = (mod._kind, alpha, float(mod.cutoff), float(mod._width()), K)
        return out.view(*r.shape[:-1], K) if r.shape[-1] == 1 else out.view(*r.shape, K)

    @staticmethod
    def backward(ctx, g):
        kind, alpha, cutoff, width, K = ctx.meta
        r2, raw_alpha, *t = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("d(radial basis)/dr (forces) is built for the exponential Bernstein basis only (basis_functions='exp-bernstein')")
        if kind not in (2, 3):
            return None, torch.zeros_like(raw_alpha), None            # "_alpha" of the Gaussian basis "doesn't do anything"
        lib = _lib.load()
        t += [None] * (3 - len(t))
        g2 = g.to(torch.float32).reshape(-1, K).contiguous()
        rows = torch.empty(r2.shape[0], device=r2.device, dtype=torch.float32)
        _lib.check(lib.nq_radial_basis_grad_alpha(kind, _lib.ptr(r2), _lib.ptr(g2), r2.shape[0], K, alpha, cutoff, width, _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]),
                                                  _lib.ptr(rows), _lib.stream_ptr()))
        g_raw = rows.double().sum() * torch.sigmoid(raw_alpha.detach().double())
        return None, g_raw.to(raw_alpha.dtype).reshape(raw_alpha.shape), None


def _bernstein_tables(K):
    logfactorial = np.zeros(K)
    for i in range(2, K):
        logfactorial[i] = logfactorial[i - 1] + np.log(i)
    v = np.arange(0, K)
    n = (K - 1) - v
    return logfactorial[-1] - logfactorial[v] - logfactorial[n], n, v


class _RadialBase(nn.Module):
    """Same constructors, buffers and parameter as the reference classes of these names (phisnet/nn/modules/*_radial_basis_functions.py; float64 buffers
    there until ``model.to(dtype)``).  forward(r [..., 1]) -> [..., K]; gradient w.r.t. ``_alpha`` only."""
    _kind = 0

    def _softplus_inverse_init(self, ini_alpha):
        x = torch.tensor(float(ini_alpha), dtype=torch.float64)
        nn.init.constant_(self._alpha, float(x + torch.log(-torch.expm1(-x))))

    def _width(self):
        return getattr(self, "width", torch.tensor(0.0))

    def forward(self, r):
        _require_gpu(r)
        alpha = self._alpha if hasattr(self, "_alpha") else r.new_zeros(())
        return _RadialFn.apply(r, alpha, self)


class GaussianRadialBasisFunctions(_RadialBase):
    _kind = 1

    def __init__(self, num_basis_functions, cutoff, dtype=torch.float32):
        super().__init__()
        self.num_basis_functions = num_basis_functions
        self.register_buffer("cutoff", torch.tensor(cutoff, dtype=dtype))
        self.register_buffer("center", torch.linspace(0, cutoff, num_basis_functions, dtype=torch.float64).to(dtype))
        self.register_buffer("width", torch.tensor(num_basis_functions / cutoff, dtype=dtype))
        self.register_parameter("_alpha", nn.Parameter(torch.tensor(1.0, dtype=dtype)))

    def reset_parameters(self):
        pass

    def _tables(self):
        return [self.center]


class ExponentialGaussianRadialBasisFunctions(_RadialBase):
    _kind = 2

    def __init__(self, num_basis_functions, cutoff, ini_alpha=0.5, dtype=torch.float32):
        super().__init__()
        self.num_basis_functions, self.ini_alpha = num_basis_functions, ini_alpha
        self.register_buffer("cutoff", torch.tensor(cutoff, dtype=dtype))
        self.register_buffer("center", torch.linspace(1, 0, num_basis_functions, dtype=torch.float64).to(dtype))
        self.register_buffer("width", torch.tensor(1.0 * num_basis_functions, dtype=dtype))
        self.register_parameter("_alpha", nn.Parameter(torch.tensor(1.0, dtype=dtype)))
        self.reset_parameters()

    def reset_parameters(self):
        self._softplus_inverse_init(self.ini_alpha)

    def _tables(self):
        return [self.center]


class OverlapBernsteinRadialBasisFunctions(_RadialBase):
    _kind = 3

    def __init__(self, num_basis_functions, cutoff, ini_alpha=0.5, dtype=torch.float32):
        super().__init__()
        self.num_basis_functions, self.ini_alpha = num_basis_functions, ini_alpha
        logc, n, v = _bernstein_tables(num_basis_functions)
        self.register_buffer("cutoff", torch.tensor(cutoff, dtype=dtype))
        self.register_buffer("logc", torch.tensor(logc, dtype=dtype))
        self.register_buffer("n", torch.tensor(n, dtype=dtype))
        self.register_buffer("v", torch.tensor(v, dtype=dtype))
        self.register_parameter("_alpha", nn.Parameter(torch.tensor(1.0, dtype=dtype)))
        self.reset_parameters()

    def reset_parameters(self):
        self._softplus_inverse_init(self.ini_alpha)

    def _tables(self):
        return [self.logc, self.n, self.v]


class BernsteinRadialBasisFunctions(_RadialBase):
    _kind = 4

    def __init__(self, num_basis_functions, cutoff, dtype=torch.float32):
        super().__init__()
        self.num_basis_functions = num_basis_functions
        logc, n, v = _bernstein_tables(num_basis_functions)
        self.register_buffer("cutoff", torch.tensor(cutoff, dtype=dtype))
        self.register_buffer("logc", torch.tensor(logc, dtype=dtype))
        self.register_buffer("n", torch.tensor(n, dtype=dtype))
        self.register_buffer("v", torch.tensor(v, dtype=dtype))

    def reset_parameters(self):
        pass

    def _tables(self):
        return [self.logc, self.n, self.v]


RADIAL_BASES = {"exp-bernstein": ExponentialBernsteinRadialBasisFunctions, "exp-gaussian": ExponentialGaussianRadialBasisFunctions,
                "gaussian": GaussianRadialBasisFunctions, "bernstein": BernsteinRadialBasisFunctions, "overlap-bernstein": OverlapBernsteinRadialBasisFunctions}
