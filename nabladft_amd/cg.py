"""Real-basis Clebsch-Gordan tensors (unit Frobenius norm, i.e. real Wigner 3j symbols up to sign) computed from scratch: complex
Clebsch-Gordan coefficients by Racah's formula, transformed to real spherical harmonics (Condon-Shortley phase, components ordered
m = -l..l, Y_1 ~ (y, z, x)) -- the basis of PhiSNet's spherical_harmonics (nablaDFT/phisnet/nn/spherical_harmonics/spherical_harmonics.py:10-25).

For given (l1, l2, L) the rotation-invariant tensor is unique up to its sign.  ``canonical(l1, l2, L)`` fixes the sign by construction
(real part of the transformed coefficients for even l1+l2+L, imaginary part for odd); the kernels (csrc/so3.hip, generated table
csrc/cg_l4.inc) hard-code these tensors.  A model brings its own table (PhiSNet: ClebschGordan buffers ``cg_{l1}_{l2}_{L}`` loaded
from clebsch_gordan_coefficients_L10.npz, phisnet/nn/modules/clebsch_gordan.py:13-28) whose per-path signs are an arbitrary
convention: ``path_signs`` checks that table against the canonical tensors and returns the signs, which are folded into the path
coefficients at run time.
"""
from functools import lru_cache
from math import factorial as _f, sqrt

import numpy as np

LMAX = 4


def _cg_complex(j1, m1, j2, m2, J, M):
    if m1 + m2 != M or J < abs(j1 - j2) or J > j1 + j2 or abs(M) > J:
        return 0.0
    pref = sqrt((2 * J + 1) * _f(J + j1 - j2) * _f(J - j1 + j2) * _f(j1 + j2 - J) / _f(j1 + j2 + J + 1))
    pref *= sqrt(_f(J + M) * _f(J - M) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    s = 0.0
    for k in range(0, j1 + j2 - J + 1):
        den = [k, j1 + j2 - J - k, j1 - m1 - k, j2 + m2 - k, J - j2 + m1 + k, J - j1 - m2 + k]
        if min(den) < 0:
            continue
        d = 1
        for x in den:
            d *= _f(x)
        s += (-1) ** k / d
    return pref * s


def _real_from_complex(l):
    """U with Y_real = U @ Y_complex; rows m = -l..l (sin-type for m < 0, cos-type for m > 0), columns mu = -l..l."""
    n = 2 * l + 1
    u = np.zeros((n, n), dtype=complex)
    for m in range(-l, l + 1):
        i = m + l
        if m == 0:
            u[i, l] = 1
        elif m > 0:
            u[i, l + m] = (-1) ** m / sqrt(2)
            u[i, l - m] = 1 / sqrt(2)
        else:
            a = -m
            u[i, l - a] = 1j / sqrt(2)
            u[i, l + a] = -1j * (-1) ** a / sqrt(2)
    return u


@lru_cache(maxsize=None)
def canonical(l1, l2, L):
    """float64 [2l1+1, 2l2+1, 2L+1], unit norm, sum_{m1 m2} T[m1,m2,M] Y_{l1 m1} Y_{l2 m2} transforms like Y_{L M}."""
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * L + 1))
    for a in range(-l1, l1 + 1):
        for b in range(-l2, l2 + 1):
            if abs(a + b) <= L:
                C[a + l1, b + l2, a + b + L] = _cg_complex(l1, a, l2, b, L, a + b)
    T = np.einsum("ia,jb,kc,abc->ijk", _real_from_complex(l1), _real_from_complex(l2), np.conj(_real_from_complex(L)), C.astype(complex)) / sqrt(2 * L + 1)
    out = T.real if (l1 + l2 + L) % 2 == 0 else T.imag
    out = np.where(np.abs(out) < 1e-14, 0.0, out)
    assert abs((out ** 2).sum() - 1.0) < 1e-12
    return out


def paths(order_in1=LMAX, order_in2=LMAX, order_out=LMAX):
    """(l1, l2, L) in the loop order of PairMixing.forward (phisnet/nn/modules/pair_mixing.py:55-68)."""
    return [(l1, l2, L) for l1 in range(order_in1 + 1) for l2 in range(order_in2 + 1) for L in range(abs(l1 - l2), min(l1 + l2, order_out) + 1)]


ALL_PATHS = paths()
PATH_ID = {p: i for i, p in enumerate(ALL_PATHS)}


def path_signs(table, which):
    """table(l1, l2, L) -> array-like [2l1+1, 2l2+1, 2L+1] (the model's own CG provider).  Returns +-1 per path of ``which`` after checking
    that every tensor equals +-canonical (same real basis, same normalisation); raises ValueError otherwise."""
    signs = []
    for (l1, l2, L) in which:
        t = np.asarray(table(l1, l2, L), dtype=np.float64)
        c = canonical(l1, l2, L)
        if t.shape != c.shape:
            raise ValueError(f"Clebsch-Gordan tensor ({l1},{l2},{L}) has shape {t.shape}, expected {c.shape}")
        ov = float((t * c).sum())
        if abs(abs(ov) - 1.0) > 1e-6 or np.abs(t - np.sign(ov) * c).max() > 1e-6:
            raise ValueError(f"Clebsch-Gordan tensor ({l1},{l2},{L}) is not +-(unit-norm real 3j tensor in the m=-l..l Condon-Shortley basis)")
        signs.append(1.0 if ov > 0 else -1.0)
    return signs


# ---- e3nn's convention (QHNet) ------------------------------------------------------------------------------------------------------------
def _e3nn_real_to_complex(l):
    """Change of basis q[complex m, real index] of e3nn 0.5.1 (o3._wigner.change_basis_real_to_complex; third-party, restated from the
    published behaviour -- SURVEY.md Appendix A): the usual real <-> complex relation times (-i)^l, which makes every real-basis coupling
    tensor real.  e3nn orders the real components m = -l..l with Y_1 = (x, y, z) of ITS input; QHNet feeds (y, z, x) (qhnet.py:268), so in
    molecule coordinates the basis is the Condon-Shortley one of ``canonical`` and only a sign per (l1, l2, L) can differ."""
    n = 2 * l + 1
    q = np.zeros((n, n), dtype=complex)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / sqrt(2)
        q[l + m, l - abs(m)] = -1j / sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j_e3nn(l1, l2, L):
    """e3nn.o3.wigner_3j(l1, l2, L) restated (unit Frobenius norm): complex Clebsch-Gordan coefficients (Racah) in e3nn's real basis."""
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * L + 1), dtype=complex)
    for a in range(-l1, l1 + 1):
        for b in range(-l2, l2 + 1):
            if abs(a + b) <= L:
                C[a + l1, b + l2, a + b + L] = _cg_complex(l1, a, l2, b, L, a + b)
    T = np.einsum("ij,kl,mn,ikn->jlm", _e3nn_real_to_complex(l1), _e3nn_real_to_complex(l2), np.conj(_e3nn_real_to_complex(L).T), C)
    assert np.abs(T.imag).max() < 1e-12
    T = np.where(np.abs(T.real) < 1e-14, 0.0, T.real)
    return T / np.sqrt((T ** 2).sum())


@lru_cache(maxsize=None)
def e3nn_sign(l1, l2, L):
    """+-1 with wigner_3j_e3nn(l1, l2, L) == sign * canonical(l1, l2, L) (checked)."""
    t, c = wigner_3j_e3nn(l1, l2, L), canonical(l1, l2, L)
    ov = float((t * c).sum())
    if abs(abs(ov) - 1.0) > 1e-9 or np.abs(t - np.sign(ov) * c).max() > 1e-9:
        raise AssertionError(f"e3nn-convention 3j tensor ({l1},{l2},{L}) is not +-canonical")
    return 1.0 if ov > 0 else -1.0
