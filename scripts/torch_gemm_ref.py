"""Development: what the vendor GEMM (rocBLAS / hipBLASLt behind torch.mm) reaches at the step's shapes, fp32, for comparison with gemm.hip."""
import torch

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
N, E = 42788, 801078
shapes = [("nt W1", N, 128, 128), ("nt W2/V2", N, 384, 128), ("nt U", 3 * N, 256, 128), ("nt V1", N, 128, 256), ("nt U 2048", 6 * N, 256, 128),
          ("nn U", 3 * N, 256, 128), ("nn U 2x", 6 * N, 256, 128), ("nn W2", 2 * N, 384, 128), ("nn V1", 2 * N, 128, 256), ("nn W1", 2 * N, 128, 128),
          ("tn W2", 2 * N, 384, 128), ("tn W1", 2 * N, 128, 128), ("tn U", 6 * N, 256, 128), ("tn V1", 2 * N, 128, 256)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, M, Nn, K in shapes:
    kind = name[:2]
    if kind == "nt":
        A, W = torch.randn(M, K, device=dev), torch.randn(Nn, K, device=dev)
        out = torch.empty(M, Nn, device=dev)
        fn = lambda: torch.mm(A, W.t(), out=out)
    elif kind == "nn":
        G, W = torch.randn(M, Nn, device=dev), torch.randn(Nn, K, device=dev)
        out = torch.empty(M, K, device=dev)
        fn = lambda: torch.mm(G, W, out=out)
    else:
        G, X = torch.randn(M, Nn, device=dev), torch.randn(M, K, device=dev)
        out = torch.empty(Nn, K, device=dev)
        fn = lambda: torch.mm(G.t(), X, out=out)
    ms = timeit(fn)
    print(f"{name:12s} M={M:8d} n={Nn:4d} k={K:4d} | torch.mm {ms:7.3f} ms {2.0 * M * Nn * K / ms / 1e9:6.1f} TF")
