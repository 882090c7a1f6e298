"""The arithmetic of the split-bf16 GEMM engine (csrc/gemm_split.h), emulated in numpy on the CPU: no kernel runs here -- this pins the MATH the kernel relies on.
  * every float32 value is the sum of three bfloat16 pieces obtained by round-to-nearest-even of the running remainder, up to 2^-25 of its magnitude
    (the two subtractions are exact in float32);
  * the six piece products h h' + h m' + m h' + h l' + l h' + m m', accumulated in float32 per block of 16 k like the matrix instruction does, reproduce a
    float32 dot product as well as a plain float32 accumulation does (measured against float64), also for wide dynamic range and for cancelling sums;
  * three pieces products (h h' + h m' + m h') do NOT (that variant was measured faster and rejected)."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    s = (r - m).astype(np.float32)
    return h, m, bf16_rne(s), r, s


def split_dot(A, B, terms=6):
    """A [M, K] . B [N, K]^T with the engine's arithmetic: exact piece products (float64 holds a 16-bit x 16-bit product and a 16-term sum of them exactly enough),
    one float32 rounding per (term, block of 16 k) into the accumulator, small terms first."""
    pa, pb = split3(A)[:3], split3(B)[:3]
    order = [(2, 2), (2, 1), (1, 2), (1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)][9 - terms:]
    acc = np.zeros((A.shape[0], B.shape[0]), np.float32)
    for k0 in range(0, A.shape[1], 16):
        for ia, ib in order:
            blk = pa[ia][:, k0:k0 + 16].astype(np.float64) @ pb[ib][:, k0:k0 + 16].astype(np.float64).T
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    return acc


def f32_dot(A, B):
    """plain float32: one rounding per multiply-add, k ascending (what v_mfma_f32_32x32x2_f32 does two k at a time)."""
    acc = np.zeros((A.shape[0], B.shape[0]), np.float32)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k + 1] * B[:, k][None, :]).astype(np.float32)
    return acc


def test_three_pieces_reproduce_every_float32_value():
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    e = (bits >> 23) & 0xFF
    x = x[(e >= 40) & (e <= 215)]                                   # normal numbers whose third piece is still normal, no overflow in the rounding
    h, m, l, r, s = split3(x)
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))      # the remainders are exact in float32
    assert np.array_equal(s.astype(np.float64), r.astype(np.float64) - m.astype(np.float64))
    err = np.abs(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64) - x.astype(np.float64))
    assert float((err / np.abs(x.astype(np.float64))).max()) <= 2.0 ** -25
    assert np.all(np.abs(r) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(s) <= np.abs(x) * 2.0 ** -16)
    for p in (h, m, l):
        assert np.array_equal(p, bf16_rne(p))                       # every piece IS a bfloat16


def _errors(A, B):
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    scale = np.abs(ref).max()
    return (np.abs(split_dot(A, B) - ref).max() / scale, np.abs(f32_dot(A, B) - ref).max() / scale, np.abs(split_dot(A, B, terms=3) - ref).max() / scale)


def test_six_piece_products_are_as_accurate_as_float32_accumulation():
    rng = np.random.default_rng(1)
    cases = {
        "uniform": (rng.uniform(-1, 1, (48, 256)), rng.uniform(-1, 1, (40, 256))),
        "one-signed": (rng.uniform(0, 1, (48, 256)), rng.uniform(0, 1, (40, 256))),
        "seven decades per row": (rng.standard_normal((48, 256)) * np.exp(4 * rng.standard_normal((48, 1))), rng.standard_normal((40, 256))),
        "seven decades along k": (rng.standard_normal((48, 256)) * np.exp(4 * rng.standard_normal((1, 256))), rng.standard_normal((40, 256))),
    }
    for name, (A, B) in cases.items():
        e_split, e_f32, e_three = _errors(A.astype(np.float32), B.astype(np.float32))
        assert e_split <= 1.5 * e_f32 + 1e-8, (name, e_split, e_f32)
        assert e_three > 2 * e_split and e_three > 1.5 * e_f32, (name, e_three, e_split, e_f32)     # without the 2^-16-level terms: visibly worse than float32
    # cancellation: the result is 1e-4 of the sum of magnitudes; both arithmetics lose the same digits
    A = rng.standard_normal((16, 512)).astype(np.float32)
    B = rng.standard_normal((16, 512)).astype(np.float32)
    A[:, 256:] = A[:, :256]
    B[:, 256:] = -B[:, :256] * (1 + 1e-4 * rng.standard_normal((16, 256))).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.abs(split_dot(A, B) - ref).max() <= 1.5 * np.abs(f32_dot(A, B) - ref).max() + 2.0 ** -24 * mag.max()
