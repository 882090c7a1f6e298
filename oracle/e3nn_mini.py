"""TEST INFRASTRUCTURE -- CPU restatement of the part of e3nn 0.5.1 that nablaDFT's QHNet uses.  *** parity unpinned ***

e3nn (pinned ``e3nn==0.5.1`` in /root/reference/setup.py:26-52) is an un-vendored third-party dependency of
/root/reference/nablaDFT/qhnet/{layers,qhnet}.py (imports at layers.py:5-7, qhnet.py:6-7); it is not under /root/reference and not installed
in the build container, so nothing here can be checked against the package itself.  The algorithms below are restated from the package's
published behaviour:

  Irrep / Irreps                 e3nn.o3.Irrep, e3nn.o3.Irreps (layout of a feature vector: for each entry ``mul x ir`` a block
                                 [mul, 2l+1] row-major, blocks concatenated)
  wigner_3j                      e3nn.o3.wigner_3j: SU(2) Clebsch-Gordan coefficients (Racah formula, Condon-Shortley) taken to e3nn's real
                                 basis with Q_l = (-i)^l * (real -> complex change of basis), then Frobenius-normalised
  spherical_harmonics            e3nn.o3.spherical_harmonics: Y_0 = 1, Y_1 = (x, y, z), Y_{l+1} ~ w3j(l, 1, l+1) . Y_l . Y_1, unit norm per l on the
                                 unit sphere; 'component' normalisation multiplies by sqrt(2l+1)
  TensorProduct                  e3nn.o3.TensorProduct (modes uvu / uuu / uvw, irrep_normalization 'component'|'norm', path_normalization
                                 'element'; weights flattened in instruction order; coefficient sqrt(alpha),
                                 alpha = ir_out.dim * path_weight / sum_{instructions into the same output} fan-in)
  ElementwiseTensorProduct, Norm e3nn.o3.ElementwiseTensorProduct, e3nn.o3.Norm
  Linear                         e3nn.o3.Linear (per matching irrep W[mul_in, mul_out] / sqrt(sum of mul_in), N(0,1) init, bias on 0e outputs)
  FullyConnectedNet              e3nn.nn.FullyConnectedNet (x @ W / sqrt(h_in); activation rescaled to unit second moment by the Monte-Carlo
                                 constant of e3nn.math.normalize2mom: 10^6 float64 normal samples from torch.Generator().manual_seed(0))

It is used in two ways, both inside tests / fixture generation only:
  * oracle/qhnet_ref_import.py registers this module as ``e3nn`` so that the REAL reference files qhnet/layers.py and qhnet/qhnet.py run on top of
    it -- every QHNet-specific line (path selection, the shadowed-variable normalisation, dst||dst invariants, layer wiring, Expansion, block
    assembly) is then the reference's own code, and only the e3nn semantics above remain "[memory]";
  * the CPU tests check its internal consistency (equivariance under random rotations, Norm/inner-product identities, agreement of the
    spherical harmonics with the in-tree PhiSNet closed forms, agreement of the 3j tensors with the in-tree Clebsch-Gordan table up to sign).
"""
import collections
import math
from functools import lru_cache

import torch
from torch import nn

# ----------------------------------------------------------------------------------------------------------------------------------------
# Irrep / Irreps


class Irrep(tuple):
    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                name = l.strip()
                p = {"e": 1, "o": -1, "y": None}[name[-1]]
                l = int(name[:-1])
                if p is None:
                    p = (-1) ** l
            elif isinstance(l, tuple):
                l, p = l
        assert isinstance(l, int) and l >= 0 and p in (-1, 1)
        return super().__new__(cls, (l, p))

    @property
    def l(self):  # noqa: E743
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self.l + 1

    def is_scalar(self):
        return self.l == 0 and self.p == 1

    def __mul__(self, other):
        other = Irrep(other)
        p = self.p * other.p
        for l in range(abs(self.l - other.l), self.l + other.l + 1):
            yield Irrep(l, p)

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


class _MulIr(tuple):
    def __new__(cls, mul, ir=None):
        if ir is None:
            mul, ir = mul
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self.mul * self.ir.dim

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip() != "":
                for part in irreps.split("+"):
                    part = part.strip()
                    if "x" in part:
                        mul, ir = part.split("x")
                        out.append(_MulIr(int(mul), Irrep(ir)))
                    else:
                        out.append(_MulIr(1, Irrep(part)))
        elif irreps is None:
            pass
        else:
            for item in irreps:
                if isinstance(item, (str, Irrep)):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    def slices(self):
        s, i = [], 0
        for mul_ir in self:
            s.append(slice(i, i + mul_ir.dim))
            i += mul_ir.dim
        return s

    @property
    def dim(self):
        return sum(mul_ir.dim for mul_ir in self)

    @property
    def num_irreps(self):
        return sum(mul for mul, _ in self)

    def __getitem__(self, i):
        x = super().__getitem__(i)
        if isinstance(i, slice):
            return Irreps(x)
        return x

    def __contains__(self, ir):
        ir = Irrep(ir)
        return ir in (irrep for _, irrep in self)

    def count(self, ir):
        ir = Irrep(ir)
        return sum(mul for mul, irrep in self if ir == irrep)

    def __add__(self, other):
        return Irreps(super().__add__(Irreps(other)))

    def simplify(self):
        out = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def sort(self):
        Ret = collections.namedtuple("sort", ["irreps", "p", "inv"])
        out = sorted((ir, i, mul) for i, (mul, ir) in enumerate(self))
        inv = tuple(i for _, i, _ in out)
        p = [0] * len(inv)
        for k, i in enumerate(inv):
            p[i] = k
        return Ret(Irreps([(mul, ir) for ir, _, mul in out]), tuple(p), inv)

    def __repr__(self):
        return "+".join(f"{mul_ir}" for mul_ir in self)


# ----------------------------------------------------------------------------------------------------------------------------------------
# Wigner 3j in e3nn's real basis


def _fact(n):
    return math.factorial(n)


def _su2_cg(j1, m1, j2, m2, j3, m3):
    """<j1 m1 j2 m2 | j3 m3>, Condon-Shortley convention (Racah's formula), integer j."""
    if m3 != m1 + m2 or j3 < abs(j1 - j2) or j3 > j1 + j2:
        return 0.0
    c = math.sqrt((2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3) / _fact(j1 + j2 + j3 + 1))
    c *= math.sqrt(_fact(j3 + m3) * _fact(j3 - m3) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2))
    s = 0.0
    for k in range(0, j1 + j2 - j3 + 1):
        den = [k, j1 + j2 - j3 - k, j1 - m1 - k, j2 + m2 - k, j3 - j2 + m1 + k, j3 - j1 - m2 + k]
        if min(den) < 0:
            continue
        d = 1
        for x in den:
            d *= _fact(x)
        s += (-1) ** k / d
    return c * s


def change_basis_real_to_complex(l):
    """q[complex m, real index]; the trailing (-i)^l makes every real-basis Clebsch-Gordan tensor real."""
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _w3j64(l1, l2, l3):
    C = torch.zeros(2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1, dtype=torch.complex128)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                C[l1 + m1, l2 + m2, l3 + m1 + m2] = _su2_cg(l1, m1, l2, m2, l3, m1 + m2)
    Q1, Q2, Q3 = change_basis_real_to_complex(l1), change_basis_real_to_complex(l2), change_basis_real_to_complex(l3)
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert float(C.imag.abs().max()) < 1e-9, (l1, l2, l3)
    C = C.real
    C = torch.where(C.abs() < 1e-14, torch.zeros_like(C), C)
    return C / C.norm()


def wigner_3j(l1, l2, l3, dtype=None, device=None):
    assert abs(l2 - l3) <= l1 <= l2 + l3
    return _w3j64(int(l1), int(l2), int(l3)).to(dtype=dtype or torch.get_default_dtype(), device=device)


# ----------------------------------------------------------------------------------------------------------------------------------------
# spherical harmonics


def _sh_unit(lmax, u):
    """list over l of [..., 2l+1], each of unit Euclidean norm for unit u (e3nn coordinates: Y_1 = (x, y, z))."""
    out = [torch.ones_like(u[..., :1]), u]
    for l in range(1, lmax):
        w = _w3j64(l, 1, l + 1).to(u.dtype)
        y = torch.einsum("ijk,...i,...j->...k", w, out[l], u)
        # a harmonic polynomial of degree l+1 built equivariantly from unit-norm pieces has a constant norm on the sphere
        pole = torch.zeros(3, dtype=torch.float64)
        pole[1] = 1.0
        ref = [torch.ones(1, dtype=torch.float64), pole]
        for k in range(1, l + 1):
            ref.append(torch.einsum("ijk,i,j->k", _w3j64(k, 1, k + 1), ref[k], pole))
            ref[-1] = ref[-1] / ref[-1].norm()
        c = torch.einsum("ijk,i,j->k", _w3j64(l, 1, l + 1), ref[l], pole).norm()
        out.append(y / c.to(u.dtype))
    return out[: lmax + 1]


def spherical_harmonics(l, x, normalize, normalization="integral"):
    if isinstance(l, Irreps):
        ls = [ir.l for _, ir in l]
    elif isinstance(l, int):
        ls = [l]
    else:
        ls = list(l)
    if normalize:
        x = torch.nn.functional.normalize(x, dim=-1)
    # normalize=False (escn.py:170-176): the harmonic POLYNOMIALS on the raw vector, |x|^l Y_l(x / |x|) -- the recursion below is homogeneous of degree l
    ys = _sh_unit(max(ls), x)
    fac = {"component": lambda k: math.sqrt(2 * k + 1), "norm": lambda k: 1.0, "integral": lambda k: math.sqrt((2 * k + 1) / (4 * math.pi))}[normalization]
    return torch.cat([ys[k] * fac(k) for k in ls], dim=-1)


# ----------------------------------------------------------------------------------------------------------------------------------------
# tensor products

Instruction = collections.namedtuple("Instruction", "i_in1 i_in2 i_out connection_mode has_weight path_weight path_shape")


def _prod(x):
    out = 1
    for a in x:
        out *= a
    return out


class TensorProduct(nn.Module):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, in1_var=None, in2_var=None, out_var=None, irrep_normalization=None,
                 path_normalization=None, internal_weights=None, shared_weights=None, **_ignored):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        irrep_normalization = irrep_normalization or "component"
        path_normalization = path_normalization or "element"
        instructions = [tuple(x) if len(x) == 6 else tuple(x) + (1.0,) for x in instructions]
        ins = []
        for i1, i2, io, mode, has_w, pw in instructions:
            m1, m2, mo = self.irreps_in1[i1].mul, self.irreps_in2[i2].mul, self.irreps_out[io].mul
            shape = {"uvw": (m1, m2, mo), "uvu": (m1, m2), "uvv": (m1, m2), "uuw": (m1, mo), "uuu": (m1,), "uvuv": (m1, m2)}[mode]
            ins.append(Instruction(i1, i2, io, mode, has_w, pw, shape))
        in1_var = [1.0] * len(self.irreps_in1) if in1_var is None else in1_var
        in2_var = [1.0] * len(self.irreps_in2) if in2_var is None else in2_var
        out_var = [1.0] * len(self.irreps_out) if out_var is None else out_var

        def num_elements(i):
            return {"uvw": self.irreps_in1[i.i_in1].mul * self.irreps_in2[i.i_in2].mul, "uvu": self.irreps_in2[i.i_in2].mul,
                    "uvv": self.irreps_in1[i.i_in1].mul, "uuw": self.irreps_in1[i.i_in1].mul, "uuu": 1, "uvuv": 1}[i.connection_mode]

        final = []
        for i in ins:
            ir1, ir2, iro = self.irreps_in1[i.i_in1].ir, self.irreps_in2[i.i_in2].ir, self.irreps_out[i.i_out].ir
            assert ir1.p * ir2.p == iro.p and abs(ir1.l - ir2.l) <= iro.l <= ir1.l + ir2.l
            alpha = {"component": iro.dim, "norm": ir1.dim * ir2.dim, "none": 1}[irrep_normalization]
            if path_normalization == "element":
                x = sum(in1_var[k.i_in1] * in2_var[k.i_in2] * num_elements(k) for k in ins if k.i_out == i.i_out)
            elif path_normalization == "path":
                x = in1_var[i.i_in1] * in2_var[i.i_in2] * num_elements(i) * len([k for k in ins if k.i_out == i.i_out])
            else:
                x = 1
            if x > 0.0:
                alpha /= x
            alpha *= out_var[i.i_out]
            alpha *= i.path_weight
            final.append(i._replace(path_weight=math.sqrt(alpha)))
        self.instructions = final
        self.weight_numel = sum(_prod(i.path_shape) for i in final if i.has_weight)
        if shared_weights is False and internal_weights is None:
            internal_weights = False
        if shared_weights is None and internal_weights is None:
            shared_weights = internal_weights = True
        if shared_weights is None:
            shared_weights = True
        if internal_weights is None:
            internal_weights = shared_weights and any(i.has_weight for i in final)
        self.internal_weights, self.shared_weights = internal_weights, shared_weights
        if internal_weights and self.weight_numel > 0:
            self.weight = nn.Parameter(torch.randn(self.weight_numel))
        else:
            self.register_buffer("weight", torch.Tensor())
        self.register_buffer("output_mask", torch.ones(self.irreps_out.dim))

    def forward(self, x1, x2, weight=None):
        if weight is None:
            weight = self.weight
        lead = x1.shape[:-1]
        x1 = x1.reshape(-1, x1.shape[-1])
        x2 = x2.reshape(-1, x2.shape[-1])
        z = x1.shape[0]
        if self.weight_numel > 0 and not self.shared_weights:
            weight = weight.reshape(-1, self.weight_numel)
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        a = [x1[:, s].reshape(z, mi.mul, mi.ir.dim) for s, mi in zip(s1, self.irreps_in1)]
        b = [x2[:, s].reshape(z, mi.mul, mi.ir.dim) for s, mi in zip(s2, self.irreps_in2)]
        outs = [None] * len(self.irreps_out)
        off = 0
        for i in self.instructions:
            ir1, ir2, iro = self.irreps_in1[i.i_in1].ir, self.irreps_in2[i.i_in2].ir, self.irreps_out[i.i_out].ir
            w = None
            if i.has_weight:
                n = _prod(i.path_shape)
                w = weight[..., off:off + n].reshape((-1,) + i.path_shape) if not self.shared_weights else weight[off:off + n].reshape(i.path_shape)
                off += n
            C = wigner_3j(ir1.l, ir2.l, iro.l, dtype=x1.dtype, device=x1.device)
            X1, X2 = a[i.i_in1], b[i.i_in2]
            mode = i.connection_mode
            if mode == "uvu":
                if w is None:
                    r = torch.einsum("ijk,zui,zvj->zuk", C, X1, X2)
                elif self.shared_weights:
                    r = torch.einsum("uv,ijk,zui,zvj->zuk", w, C, X1, X2)
                else:
                    r = torch.einsum("zuv,ijk,zui,zvj->zuk", w, C, X1, X2)
            elif mode == "uuu":
                if w is None:
                    r = torch.einsum("ijk,zui,zuj->zuk", C, X1, X2)
                elif self.shared_weights:
                    r = torch.einsum("u,ijk,zui,zuj->zuk", w, C, X1, X2)
                else:
                    r = torch.einsum("zu,ijk,zui,zuj->zuk", w, C, X1, X2)
            elif mode == "uvw":
                if self.shared_weights:
                    r = torch.einsum("uvw,ijk,zui,zvj->zwk", w, C, X1, X2)
                else:
                    r = torch.einsum("zuvw,ijk,zui,zvj->zwk", w, C, X1, X2)
            else:
                raise NotImplementedError(mode)
            r = (i.path_weight * r).reshape(z, -1)
            outs[i.i_out] = r if outs[i.i_out] is None else outs[i.i_out] + r
        outs = [o if o is not None else x1.new_zeros(z, mi.dim) for o, mi in zip(outs, self.irreps_out)]
        return torch.cat(outs, dim=-1).reshape(*lead, self.irreps_out.dim)


class ElementwiseTensorProduct(TensorProduct):
    def __init__(self, irreps_in1, irreps_in2, filter_ir_out=None, irrep_normalization=None, **kw):
        irreps_in1, irreps_in2 = Irreps(irreps_in1).simplify(), Irreps(irreps_in2).simplify()
        assert irreps_in1.num_irreps == irreps_in2.num_irreps
        a, b = list(irreps_in1), list(irreps_in2)
        i = 0
        while i < len(a):      # split the entries so that multiplicities match pairwise
            m1, ir1 = a[i]
            m2, ir2 = b[i]
            if m1 < m2:
                b[i] = _MulIr(m1, ir2)
                b.insert(i + 1, _MulIr(m2 - m1, ir2))
            if m2 < m1:
                a[i] = _MulIr(m2, ir1)
                a.insert(i + 1, _MulIr(m1 - m2, ir1))
            i += 1
        out, instr = [], []
        for i, ((mul, ir1), (mul2, ir2)) in enumerate(zip(a, b)):
            assert mul == mul2
            for ir in ir1 * ir2:
                if filter_ir_out is not None and ir not in filter_ir_out:
                    continue
                instr.append((i, i, len(out), "uuu", False))
                out.append((mul, ir))
        super().__init__(Irreps(a), Irreps(b), Irreps(out), instr, irrep_normalization=irrep_normalization, **kw)


class Norm(nn.Module):
    def __init__(self, irreps_in, squared=False):
        super().__init__()
        irreps_in = Irreps(irreps_in).simplify()
        irreps_out = Irreps([(mul, "0e") for mul, _ in irreps_in])
        instr = [(i, i, i, "uuu", False, ir.dim) for i, (mul, ir) in enumerate(irreps_in)]
        self.tp = TensorProduct(irreps_in, irreps_in, irreps_out, instr, irrep_normalization="component")
        self.irreps_in, self.irreps_out, self.squared = irreps_in, irreps_out.simplify(), squared

    def forward(self, features):
        out = self.tp(features, features)
        return out if self.squared else out.relu().sqrt()


class Linear(nn.Module):
    def __init__(self, irreps_in, irreps_out, internal_weights=None, shared_weights=None, instructions=None, biases=False, path_normalization="element",
                 **_ignored):
        super().__init__()
        self.irreps_in, self.irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        if instructions is None:
            instructions = [(i, o) for i, (_, a) in enumerate(self.irreps_in) for o, (_, b) in enumerate(self.irreps_out) if a == b]
        self.paths = []
        for i, o in instructions:
            fan = sum(self.irreps_in[k if path_normalization == "element" else i].mul for k, oo in instructions if oo == o)
            self.paths.append((i, o, (self.irreps_in[i].mul, self.irreps_out[o].mul), (fan if fan else 1.0) ** -0.5))
        if isinstance(biases, bool):
            biases = [biases and ir.is_scalar() for _, ir in self.irreps_out]
        self.bias_on = list(biases)
        self.weight_numel = sum(_prod(s) for _, _, s, _ in self.paths)
        self.bias_numel = sum(mi.dim for b, mi in zip(self.bias_on, self.irreps_out) if b)
        self.weight = nn.Parameter(torch.randn(self.weight_numel))
        if self.bias_numel > 0:
            self.bias = nn.Parameter(torch.zeros(self.bias_numel))
        else:
            self.register_buffer("bias", torch.Tensor())
        self.register_buffer("output_mask", torch.ones(self.irreps_out.dim))

    def forward(self, x):
        lead = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1])
        z = x.shape[0]
        xs = [x[:, s].reshape(z, mi.mul, mi.ir.dim) for s, mi in zip(self.irreps_in.slices(), self.irreps_in)]
        outs = [None] * len(self.irreps_out)
        off = 0
        for i, o, shape, pw in self.paths:
            w = self.weight[off:off + _prod(shape)].reshape(shape)
            off += _prod(shape)
            r = pw * torch.einsum("uw,zui->zwi", w, xs[i])
            outs[o] = r if outs[o] is None else outs[o] + r
        boff = 0
        for o, (b, mi) in enumerate(zip(self.bias_on, self.irreps_out)):
            if b:
                bb = self.bias[boff:boff + mi.dim].reshape(1, mi.mul, mi.ir.dim)
                boff += mi.dim
                outs[o] = bb.expand(z, -1, -1) if outs[o] is None else outs[o] + bb
        outs = [o.reshape(z, -1) if o is not None else x.new_zeros(z, mi.dim) for o, mi in zip(outs, self.irreps_out)]
        return torch.cat(outs, dim=-1).reshape(*lead, self.irreps_out.dim)


# ----------------------------------------------------------------------------------------------------------------------------------------
# e3nn.nn.FullyConnectedNet


def normalize2mom_constant(f):
    """e3nn.math.normalize2mom: cst = E[f(z)^2]^(-1/2) over 10^6 float64 standard-normal samples drawn from a CPU generator seeded with 0."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    with torch.no_grad():
        return float(f(z).pow(2).mean().pow(-0.5))


class _Act(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f
        self.cst = normalize2mom_constant(f)
        self._is_id = abs(self.cst - 1) < 1e-4

    def forward(self, x):
        return self.f(x) if self._is_id else self.f(x).mul(self.cst)


class _Layer(nn.Module):
    def __init__(self, h_in, h_out, act, var_in, var_out):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(h_in, h_out))
        self.act, self.h_in, self.var_in, self.var_out = act, h_in, var_in, var_out

    def forward(self, x):
        if self.act is not None:
            w = self.weight / (self.h_in * self.var_in) ** 0.5
            x = self.act(x @ w)
            return x * self.var_out ** 0.5
        w = self.weight / (self.h_in * self.var_in / self.var_out) ** 0.5
        return x @ w


class FullyConnectedNet(nn.Sequential):
    def __init__(self, hs, act=None, variance_in=1, variance_out=1, out_act=False):
        super().__init__()
        self.hs = list(hs)
        if act is not None:
            act = _Act(act)
        var_in = variance_in
        for i, (h1, h2) in enumerate(zip(self.hs, self.hs[1:])):
            last = i == len(self.hs) - 2
            var_out = variance_out if last else 1
            a = (act if out_act else None) if last else act
            setattr(self, f"layer{i}", _Layer(h1, h2, a, var_in, var_out))
            var_in = var_out


# ----------------------------------------------------------------------------------------------------------------------------------------
# angles and S2 grids (used by the eSCN fixtures: escn/so3.py:380-381,453-470) -- restated from e3nn 0.5.1's documented conventions [memory], PARITY UNPINNED:
#   angles_to_matrix(a, b, c) = Ry(a) Rx(b) Ry(c);  xyz_to_angles: beta = acos(y), alpha = atan2(x, z);  grid: beta_b = (b + 1/2) pi / B, alpha_a = 2 pi a / A;
#   ToS2Grid (normalization="integral") samples the integral-normalised real harmonics; FromS2Grid integrates with the equiangular quadrature weights of
#   Driscoll-Healy type, so that from_grid(to_grid(x)) = x for l <= lmax when B >= 2 (lmax + 1).  The classes expose the factors ``shb`` / ``sha`` whose
#   contraction the reference forms (escn/so3.py:457,467); here the factorisation is the trivial one (sha = identity over the longitudes).
def xyz_to_angles(xyz):
    xyz = torch.nn.functional.normalize(xyz, p=2, dim=-1).clamp(-1, 1)
    return torch.atan2(xyz[..., 0], xyz[..., 2]), torch.acos(xyz[..., 1])


def _rot_x(a):
    c, s, o, z = a.cos(), a.sin(), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([torch.stack([o, z, z], -1), torch.stack([z, c, -s], -1), torch.stack([z, s, c], -1)], -2)


def _rot_y(a):
    c, s, o, z = a.cos(), a.sin(), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)


def angles_to_matrix(alpha, beta, gamma):
    alpha, beta, gamma = torch.broadcast_tensors(alpha, beta, gamma)
    return _rot_y(alpha) @ _rot_x(beta) @ _rot_y(gamma)


def s2_grid(res_beta, res_alpha):
    betas = (torch.arange(res_beta, dtype=torch.float64) + 0.5) / res_beta * math.pi
    alphas = torch.arange(res_alpha, dtype=torch.float64) / res_alpha * 2 * math.pi
    return betas, alphas


def s2_quadrature_weights(res_beta):
    """Weights w_b of the equiangular grid beta_b = (b + 1/2) pi / B (B even) with sum_b w_b f(beta_b) = int_0^pi f sin(beta) d beta for band-limited f
    (Driscoll & Healy 1994): w_b = (2 / (B/2)) sin(beta_b) sum_{k < B/2} sin((2k + 1) beta_b) / (2k + 1) ... normalised to sum to 2."""
    half = res_beta // 2
    betas, _ = s2_grid(res_beta, 1)
    k = torch.arange(half, dtype=torch.float64)
    w = torch.stack([(2.0 / half) * torch.sin(b) * (torch.sin((2 * k + 1) * b) / (2 * k + 1)).sum() for b in betas])
    return w * (2.0 / w.sum())


def _s2_samples(lmax, res_beta, res_alpha):
    betas, alphas = s2_grid(res_beta, res_alpha)
    b, a = torch.meshgrid(betas, alphas, indexing="ij")
    xyz = torch.stack([b.sin() * a.sin(), b.cos(), b.sin() * a.cos()], dim=-1)                      # e3nn's angles_to_xyz
    return spherical_harmonics(list(range(lmax + 1)), xyz, True, "integral")                          # [B, A, (lmax+1)^2]


def _s2_degree_factor(lmax, normalization):
    """Per-coefficient factor a_l of ToS2Grid relative to the integral-normalised harmonics (FromS2Grid divides by it, so from_grid(to_grid(x)) = x in every
    normalisation): "integral" 1; "component" sqrt(4 pi / ((2l + 1)(lmax + 1))) -- unit-variance coefficients give a unit-variance signal; "norm"
    sqrt(4 pi / (lmax + 1)).  [memory of e3nn 0.5.1's s2grid.py, PARITY UNPINNED; EquiformerV2 uses "component", equiformer_v2_oc20.py:282]"""
    assert normalization in ("integral", "component", "norm")
    f = []
    for l in range(lmax + 1):
        a = {"integral": 1.0, "component": math.sqrt(4 * math.pi / ((2 * l + 1) * (lmax + 1))), "norm": math.sqrt(4 * math.pi / (lmax + 1))}[normalization]
        f += [a] * (2 * l + 1)
    return torch.tensor(f, dtype=torch.float64)


class ToS2Grid(nn.Module):
    def __init__(self, lmax=None, res=None, normalization="component", dtype=None, device=None):
        super().__init__()
        Y = (_s2_samples(lmax, res[0], res[1]) * _s2_degree_factor(lmax, normalization)).to(torch.get_default_dtype())
        self.register_buffer("sha", torch.eye(res[1], dtype=Y.dtype))                                # [a, m]
        self.register_buffer("shb", Y.permute(1, 0, 2).contiguous())                                 # [m, b, i]


class FromS2Grid(nn.Module):
    def __init__(self, res=None, lmax=None, normalization="component", lmax_in=None, dtype=None, device=None):
        super().__init__()
        assert res[0] % 2 == 0
        Y = _s2_samples(lmax, res[0], res[1]) / _s2_degree_factor(lmax, normalization)
        w = s2_quadrature_weights(res[0]) * (2 * math.pi / res[1])                                    # d(cos beta) d(alpha) per grid point
        F = (Y * w[:, None, None]).to(torch.get_default_dtype())
        self.register_buffer("sha", torch.eye(res[1], dtype=F.dtype))                                # [a, m]
        self.register_buffer("shb", F.permute(1, 0, 2).contiguous())                                 # [m, b, i]


def install():
    """Registers this module as ``e3nn`` / ``e3nn.o3`` / ``e3nn.nn`` in sys.modules (fixture generation only)."""
    import sys
    import types
    me = sys.modules[__name__]
    e3 = types.ModuleType("e3nn")
    o3 = types.ModuleType("e3nn.o3")
    for name in ("Irrep", "Irreps", "wigner_3j", "spherical_harmonics", "TensorProduct", "ElementwiseTensorProduct", "Norm", "Linear", "xyz_to_angles",
                 "angles_to_matrix", "ToS2Grid", "FromS2Grid"):
        setattr(o3, name, getattr(me, name))
    nn_mod = types.ModuleType("e3nn.nn")
    nn_mod.FullyConnectedNet = FullyConnectedNet
    e3.o3, e3.nn = o3, nn_mod
    sys.modules["e3nn"], sys.modules["e3nn.o3"], sys.modules["e3nn.nn"] = e3, o3, nn_mod
    return e3
