"""Micro-benchmark of the per-row tensor-product kernels (csrc/qhnet.hip) at QHNet sizes: C = 128, P ordered pairs / E edges of B 42-atom conformers.
    python scripts/bench_qh_tp.py [--molecules 16] [--variants 0,1,2]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nabladft_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=16)
    ap.add_argument("--variants", default="0,1,2")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    n, C = 42, 128
    N = a.molecules * n
    own = torch.arange(N, device=dev).repeat_interleave(n - 1)
    base = (own // n) * n
    k = torch.arange(n - 1, device=dev).repeat(N)
    col = base + k + (k >= (own - base)).long()
    R = own.numel()
    own32, col32 = own.int().contiguous(), col.int().contiguous()
    x = torch.randn(N, 25, C, device=dev)
    sh = torch.randn(R, 25, device=dev)
    gy = torch.randn(R, 25, C, device=dev)
    res = {}
    for var in [int(v) for v in a.variants.split(",")]:
        lib.nq_qh_set_tp_variant(var)
        for name, ps, uvu in (("uuu", 0, False), ("uvu", 1, True)):
            npth = lib.nq_qh_tp_num_paths(ps)
            w1, w2 = torch.randn(R, npth, C, device=dev), torch.randn(R, npth, C, device=dev)
            y = torch.empty(R, 25, C, device=dev)
            g1, g2, gw1, gw2 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(w1), torch.empty_like(w2)
            def fwd():
                _lib.check(lib.nq_qh_tp_forward(_lib.ptr(x), 25, _lib.ptr(own32), _lib.ptr(sh) if uvu else None, None if uvu else _lib.ptr(col32), _lib.ptr(w1), _lib.ptr(w2), R, C, ps, _lib.ptr(y), _lib.stream_ptr()))
            def bwd():
                _lib.check(lib.nq_qh_tp_backward(_lib.ptr(x), 25, _lib.ptr(own32), _lib.ptr(sh) if uvu else None, None if uvu else _lib.ptr(col32), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(gy), None, R, C, ps,
                                                 _lib.ptr(g1), None if uvu else _lib.ptr(g2), _lib.ptr(gw1), _lib.ptr(gw2), _lib.stream_ptr()))
            for tag, fn, nbytes in (("fwd", fwd, R * (2 * npth * C + 3 * 25 * C) * 4.0), ("bwd", bwd, R * (4 * npth * C + 5 * 25 * C) * 4.0)):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                res[f"v{var}_{name}_{tag}"] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}
    print("rows", R)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
