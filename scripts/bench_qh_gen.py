"""PairNetLayer tensor product with generated weights (csrc/qhgen.hip) against the materialised path it replaces, at QHNet sizes (C = 128, 65 paths, hidden 128):
    materialised forward = product h1 @ W1 + product h2 @ W2^T + b2 + k_qh_tp (reads both [P, 8320] factors);  generated = two fragment pre-splits + k_qh_tp_gen.
Also checks the outputs against each other (relative to the largest output).
    python scripts/bench_qh_gen.py [--molecules 16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nabladft_amd import _lib


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=16)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    n, C, K, NP = 42, 128, 128, 65
    N = a.molecules * n
    own = torch.arange(N, device=dev).repeat_interleave(n - 1)
    base = (own // n) * n
    k = torch.arange(n - 1, device=dev).repeat(N)
    col = base + k + (k >= (own - base)).long()
    R = own.numel()
    own32, col32 = own.int().contiguous(), col.int().contiguous()
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(N, 25, C, generator=g).to(dev)
    h1, h2 = torch.randn(R, K, generator=g).to(dev), torch.randn(R, K, generator=g).to(dev)
    W1 = (torch.randn(K, NP * C, generator=g) / K ** 0.5).to(dev)          # x @ W layout
    W2 = (torch.randn(NP * C, K, generator=g) / K ** 0.5).to(dev)          # nn.Linear layout
    b2 = torch.randn(NP * C, generator=g).to(dev)
    w1, w2 = torch.empty(R, NP * C, device=dev), torch.empty(R, NP * C, device=dev)
    y_mat, y_gen = torch.empty(R, 25, C, device=dev), torch.empty(R, 25, C, device=dev)
    nfl = int(lib.nq_qh_gen_fragment_floats(C, K))
    frag = torch.empty(2 * nfl, device=dev)
    st = _lib.stream_ptr

    def gemms():
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(h1), _lib.ptr(W1), _lib.ptr(w1), R, K, NP * C, 0, st()))
        _lib.check(lib.nq_linear_forward(_lib.ptr(h2), _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(w2), None, R, NP * C, K, st()))

    def tp_mat():
        _lib.check(lib.nq_qh_tp_forward(_lib.ptr(x), 25, _lib.ptr(own32), None, _lib.ptr(col32), _lib.ptr(w1), _lib.ptr(w2), R, C, 0, _lib.ptr(y_mat), st()))

    def presplit():
        _lib.check(lib.nq_qh_gen_presplit(_lib.ptr(W1), None, K, C, 0, _lib.ptr(frag), st()))
        _lib.check(lib.nq_qh_gen_presplit(_lib.ptr(W2), None, K, C, 1, _lib.ptr(frag[nfl:]), st()))

    def tp_gen():
        _lib.check(lib.nq_qh_tp_forward_gen(_lib.ptr(x), _lib.ptr(own32), _lib.ptr(col32), _lib.ptr(h1), _lib.ptr(h2), _lib.ptr(frag), _lib.ptr(frag[nfl:]), _lib.ptr(b2), R, C, K,
                                            _lib.ptr(y_gen), st()))
    t_g, t_t = timed(gemms), timed(tp_mat)
    t_p, t_f = timed(presplit), timed(tp_gen)
    ref = (h1.double() @ W1.double())
    ref2 = (h2.double() @ W2.double().T + b2.double())
    e1 = float((w1.double() - ref).abs().max() / ref.abs().max()), float((w2.double() - ref2).abs().max() / ref2.abs().max())
    err = float((y_gen - y_mat).abs().max() / y_mat.abs().max())
    print(f"rows {R} (molecules {a.molecules}), C {C}, hidden {K}")
    print(f"materialised: two generator products {t_g:.3f} ms + k_qh_tp {t_t:.3f} ms = {t_g + t_t:.3f} ms per layer forward")
    print(f"generated   : fragment pre-split {t_p:.3f} ms (once per step) + k_qh_tp_gen {t_f:.3f} ms")
    print(f"max |y_gen - y_mat| / max |y_mat| = {err:.2e}   (materialised factors vs float64: {e1[0]:.1e}, {e1[1]:.1e})")


if __name__ == "__main__":
    main()
