"""Development aid: nq_linear_* against torch on odd shapes."""
import torch, itertools
from nabladft_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = _lib.stream_ptr()
torch.manual_seed(0)
for M, N, K in [(158, 48, 88), (158, 48, 48), (158, 48, 8), (158, 16, 128), (158, 48, 16), (26, 32, 96), (26, 1, 32), (158, 1, 48), (158, 8, 8), (104, 16, 32), (158, 48, 24), (158, 8, 24),
                (158, 128, 88), (158, 48, 96), (158, 48, 64), (158, 48, 80), (1808, 512, 1152), (64, 256, 1280)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); g = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    _lib.check(lib.nq_linear_forward(_lib.ptr(x), _lib.ptr(W), None, _lib.ptr(y), None, M, N, K, st))
    e1 = ((y - x @ W.T).abs().max() / (x @ W.T).abs().max()).item()
    gx = torch.empty(M, K, device=dev)
    _lib.check(lib.nq_linear_input_grad(_lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, 0, st))
    e2 = ((gx - g @ W).abs().max() / (g @ W).abs().max()).item()
    gW = torch.empty(N, K, device=dev)
    scr = torch.empty(int(lib.nq_weight_grad_scratch_floats(M, N, K)) + 64, device=dev)
    _lib.check(lib.nq_linear_weight_grad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), st))
    e3 = ((gW - g.T @ x).abs().max() / (g.T @ x).abs().max()).item()
    print(f"M{M} N{N} K{K}: fwd {e1:.1e} dgrad {e2:.1e} wgrad {e3:.1e}")
