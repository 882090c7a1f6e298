"""GEMM micro-benchmark: times the three contraction layouts at the PaiNN step's real shapes for every kernel
variant (nq_set_gemm_variant), HIP events on the launch stream, interleaved rounds; checks results vs torch."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nabladft_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
N, E = 42788, 801078
shapes_nt = [("nt W1", N, 128, 128), ("nt W2/V2", N, 384, 128), ("nt U", 3 * N, 256, 128), ("nt V1", N, 128, 256), ("nt Wr", E // 4, 384, 100)]
shapes_nn = [("nn U acc", 3 * N, 256, 128), ("nn U acc 2x", 6 * N, 256, 128), ("nn W2", 2 * N, 384, 128), ("nn V1", 2 * N, 128, 256), ("nn W1 acc", 2 * N, 128, 128)]
shapes_tn = [("tn W2", 2 * N, 384, 128), ("tn W1", 2 * N, 128, 128), ("tn U", 6 * N, 256, 128), ("tn V1", 2 * N, 128, 256), ("tn Wr", 2 * E, 384, 100)]
st = _lib.stream_ptr()
g = torch.Generator(device="cpu").manual_seed(0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


rows = []
for kind, shapes in (("nt", shapes_nt), ("nn", shapes_nn), ("tn", shapes_tn)):
    for name, M, Nn, K in shapes:
        if kind == "nt":
            A, W, b = torch.randn(M, K, device=dev), torch.randn(Nn, K, device=dev) * 0.1, torch.randn(Nn, device=dev)
            Cc = torch.empty(M, Nn, device=dev)
            fn = lambda: _lib.check(lib.nq_linear_forward(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(Cc), None, M, Nn, K, st))
            ref = lambda: A[:4096] @ W.T + b
            got = lambda: Cc[:4096]
        elif kind == "nn":
            Gm, W = torch.randn(M, Nn, device=dev), torch.randn(Nn, K, device=dev) * 0.1
            Cc = torch.zeros(M, K, device=dev)
            acc = 1 if "acc" in name else 0
            fn = lambda: _lib.check(lib.nq_linear_input_grad(_lib.ptr(Gm), _lib.ptr(W), _lib.ptr(Cc), M, Nn, K, acc, st))
            ref = lambda: Gm[:4096] @ W
            got = None
        else:
            Gm, X = torch.randn(M, Nn, device=dev), torch.randn(M, K, device=dev)
            out = torch.empty(Nn, K, device=dev)
            scr = torch.empty(lib.nq_weight_grad_scratch_floats(M, Nn, K), device=dev)
            fn = lambda: _lib.check(lib.nq_linear_weight_grad(_lib.ptr(Gm), _lib.ptr(X), _lib.ptr(out), M, Nn, K, _lib.ptr(scr), st))
            ref = lambda: (Gm.double().T @ X.double()).float()
            got = lambda: out
        flops = 2.0 * M * Nn * K
        line = f"{name:12s} M={M:8d} n={Nn:4d} k={K:4d} "
        for v in range(4):
            lib.nq_set_gemm_variant(v)
            if kind == "nn":
                Cc.zero_()
                lib.nq_set_gemm_variant(v)
                fn()
                torch.cuda.synchronize()
                err = float((Cc[:4096] - ref()).abs().max() / ref().abs().max())
            ms = timeit(fn)
            if kind != "nn":
                err = float((got() - ref()).abs().max() / ref().abs().max())
            line += f"| v{v}: {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF err {err:.0e} "
        print(line, flush=True)
lib.nq_set_gemm_variant(1)
