"""How far the chunked-sum gradient is from the full-batch gradient (tests/test_engine_gpu.py::test_bench_sized_batch_linearity_and_determinism) under the two
GEMM engines, and how far the engines are from each other on the full batch.  Prints the worst tensors (relative to the tensor's largest entry)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import painn_ref as R          # noqa: E402  (debug script: the oracle only provides the synthetic conformers and parameters)
import nabladft_amd as nq                  # noqa: E402
from nabladft_amd import _lib              # noqa: E402
from tests.test_engine_gpu import _model   # noqa: E402

dev = torch.device("cuda:0")
cfg = R.PaiNNConfig()
model = _model(cfg, R.make_params(cfg, seed=23), dev)
B, C = 2048, 8
pos, z, batch, _, _ = R.gen_conformers(7, B)
g = torch.Generator().manual_seed(3)
w_e, w_f = torch.randn(B, generator=g), torch.randn(pos.shape[0], 3, generator=g)
names = [k for k, _ in model.named_parameters()]


def grads(lo, hi):
    s = (batch >= lo) & (batch < hi)
    b = nq.Batch(pos[s], z[s], batch[s] - lo).to(dev)
    for p in model.parameters():
        p.grad = None
    e, f = model(b)
    ((e * w_e[lo:hi].to(dev)).sum() + (f * w_f[s].to(dev)).sum()).backward()
    return [p.grad.double().clone() for p in model.parameters()]


def run(variant):
    _lib.load().nq_set_gemm_variant(variant)
    full = grads(0, B)
    acc = [torch.zeros_like(t) for t in full]
    for c in range(C):
        for a, t in zip(acc, grads(c * B // C, (c + 1) * B // C)):
            a += t
    return full, acc


def worst(xs, ys, n=4):
    scale = max(float(t.abs().max()) for t in ys)
    r = sorted(((float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3 * scale), k) for a, b, k in zip(xs, ys, names)), reverse=True)
    return [(f"{v:.2e}", k) for v, k in r[:n]]


full_s, acc_s = run(1)
full_x, acc_x = run(1 | 32)
full_g, acc_g = run(1 | 16)               # the generic round-1 kernels: exact f32 MFMA too, another summation order over k
_lib.load().nq_set_gemm_variant(1)
print("chunks vs full, split-bf16 engine :", worst(acc_s, full_s))
print("chunks vs full, exact-f32 engine  :", worst(acc_x, full_x))
print("full: split vs exact              :", worst(full_s, full_x))
print("chunk sums: split vs exact        :", worst(acc_s, acc_x))
print("full: exact (k_gemm2) vs exact (generic kernels, other k order):", worst(full_g, full_x))
print("full: split vs exact (generic)    :", worst(full_s, full_g))
