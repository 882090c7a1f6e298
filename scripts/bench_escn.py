"""eSCN (BASELINE.json configs[4]: config/model/escn-oc.yaml -- 8 layers, lmax 6 / mmax 2, 128 sphere channels, 256 hidden, cutoff 8 A, 40 neighbours,
128 sphere samples; AdamW(amsgrad, betas 0.9/0.95, lr 1e-3), loss = L1(E) + 100 * L2(F); config/escn-oc.yaml: batch_size 8, no gradient clip) training-step
timing on one MI355X in fp32: graph + frames + Wigner rows -> forward -> loss -> backward -> AdamW, on synthetic drug-like conformers already resident in HBM.

    python scripts/bench_escn.py [--molecules 16] [--steps 10] [--warmup 3] [--kernels] [--cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(num_targets=1, use_pbc=False, regress_forces=True, otf_graph=True, use_grid=True, distance_function="gaussian", basis_width_scalar=1.0,
           show_timing_info=False, max_neighbors=40, cutoff=8.0, max_num_elements=65, num_layers=8, lmax_list=[6], mmax_list=[2], sphere_channels=128,
           hidden_channels=256, edge_channels=128, num_sphere_samples=128, distance_resolution=0.02)          # config/model/escn-oc.yaml:5-25
MFMA_F32_PEAK_TFLOPS = 157.3
SPLIT_PEAK_TFLOPS = 2500.0 / 6   # csrc/gemm_split.h: six bf16 piece products per f32 product on the 2.5 PFLOP/s bf16 matrix pipe


def gemm_roof(what):
    """(kernel label, peak) of the dense launches for the engine that runs: the split-bf16 engine carries the large products unless NQ_GEMM_F32=1."""
    if os.environ.get("NQ_GEMM_F32", "0") not in ("", "0"):
        return f"k_gemm2 ({what}; exact-f32 MFMA)", MFMA_F32_PEAK_TFLOPS
    return (f"k_gemm3 ({what}; f32 values split exactly into three bf16 pieces, six bf16 MFMA piece products per product, f32 accumulate; the small launches "
            "stay on the exact-f32 k_gemm2)"), SPLIT_PEAK_TFLOPS


class Batch:
    pass


def synthetic_batch(molecules, seed, device, world=1, rank=0, size="drug", cost_model="escn"):
    import torch
    from nabladft_amd.synth import gen_rank_conformers
    (pos, z, batch, y, f), spread = gen_rank_conformers(seed, molecules, world, rank, size, cost_model)
    molecules = int(y.shape[0])          # this rank's share of the cost-balanced global batch
    b = Batch()
    b.pos, b.z, b.batch, b.y, b.forces = pos.to(device), z.to(device), batch.to(device), y.to(device), f.to(device)
    b.cost_spread = spread
    return b


def build(device, seed=23):
    import torch
    from nabladft_amd.escn import eSCN
    torch.manual_seed(seed)
    return eSCN(**CFG).to(device)


def loss_fn(E, F, b):
    import torch
    return (E - b.y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - b.forces, dim=-1).mean()


def run(molecules=16, steps=10, warmup=3, kernels=True, device=None, seed=1, world=1, rank=0, sync=None, precision="f32", size="drug"):
    """precision "bf16": the bias-free Dense products (the SO(2) convolutions and the grid MLP: >95 % of the flops) on bf16 MFMA with fp32 accumulation
    (nabladft_amd.gemnet_oc.set_gemm_precision); everything else fp32."""
    import torch
    from nabladft_amd import _lib, gemnet_oc
    gemnet_oc.set_gemm_precision(precision)
    from nabladft_amd import dist as nqdist
    from nabladft_amd.trainer import FlatParameters
    dev = device or torch.device("cuda", torch.cuda.current_device())
    net = build(dev)
    # every rank draws the same global batch of `molecules` x world conformers and keeps its cost-balanced share (dist.shard_by_cost, proxy "escn")
    batches = [synthetic_batch(molecules, seed * 100 + k, dev, world, rank, size, "escn") for k in range(4)]
    flat = FlatParameters(net.parameters())
    opt = torch.optim.AdamW([flat.flat], lr=1e-3, betas=(0.9, 0.95), amsgrad=True, weight_decay=0)
    from nabladft_amd.trainer import OverlappedAllReduce
    ov = OverlappedAllReduce(flat) if world > 1 else None     # gradient buckets reduced on a side stream while the backward is still running

    def step(i):
        b = batches[i % len(batches)]
        flat.zero_grad()
        E, F = net(b)
        loss = loss_fn(E, F, b)
        loss.backward()
        if ov is not None:
            ov.finish()                                   # bucketed all-reduce started by the hooks during backward; mean over ranks
        opt.step()
        return loss

    sync = sync or torch.cuda.synchronize
    if world > 1:
        nqdist.broadcast_(flat.flat.data)
    for i in range(warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    sync()
    dt = time.perf_counter() - t0
    G = net.build_graph(batches[0])
    out = {"workload": "eSCN (config/model/escn-oc.yaml: 8 layers, lmax 6 / mmax 2, 128 sphere channels, 256 hidden, cutoff 8 A, 40 neighbours, 128 sphere samples) train "
                       "step: graph, frames, Wigner rows, forward, L1(E) + 100 L2(F), backward, AdamW(amsgrad); synthetic ~42-atom conformers",
           "value": molecules * steps / dt, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dt / steps, "molecules_per_step": molecules, "sizes": str(size),
           "load_balance": {"cost_model": "escn", "this_run_predicted_spread": getattr(batches[0], "cost_spread", 0.0),
                            "predicted_spread_8_ranks_10_to_90_atoms_by_conformers_per_rank": nqdist.spread_table("escn")}, "atoms": G.N, "edges": G.E,
           "parameters": net.num_params, "_dt": dt, "final_loss": float(loss.detach()), "dtype": precision, "data": "synthetic",
           "parity": "pinned to the reference eSCN classes run on CPU (tests/golden/escn_*.npz); the five e3nn symbols under them are restated (unpinned), the Wigner "
                     "J matrices equal the reference's Jd.pt"}
    if kernels:
        gemnet_oc.GEMM_FLOPS[0] = 0.0
        step(0)
        fwd_flops = gemnet_oc.GEMM_FLOPS[0]
        gemnet_oc.GEMM_FLOPS[0] = None
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        ks = sorted(((k, v[0] / steps, v[1] // steps) for k, v in prof.items()), key=lambda x: -x[1])
        out["device_ms_per_step_nq_kernels"] = sum(v[0] for v in prof.values()) / steps
        out["kernel_ms_per_step"] = {k: [round(ms, 4), int(n)] for k, ms, n in ks[:24]}
        # dense products: every launcher records its 2 M N K with the event pair (nq_profile_read2), so flops and time cover exactly the same launches
        gemm_ms = sum(v[0] for v in prof.values() if v[2] > 0) / steps
        gemm_fl = sum(v[2] for v in prof.values()) / steps
        out["gemm_ms_per_step"] = gemm_ms
        out["gemm_classes_TFLOPs"] = {k: round(v[2] / max(v[0], 1e-9) / 1e9, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[2] > 0}
        out["dense_flops_counted_per_step"] = gemm_fl
        ach = gemm_fl / (max(gemm_ms, 1e-9) * 1e-3) / 1e12
        label, peak = gemm_roof("SO(2) convolution and grid MLP layers")
        out["roofline"] = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                           "frac_of_exact_f32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS, "gemm_ms_per_step": gemm_ms, "flops_per_step": gemm_fl}
    gemnet_oc.set_gemm_precision("f32")
    return out


def cpu_baseline(seconds_budget=25.0, conformers=2):
    """oracle/escn_ref.py (torch CPU, fp32; pinned to the reference classes' golden vectors) forward + loss + backward on ONE synthetic conformer."""
    import torch
    from nabladft_amd.escn import eSCN
    from nabladft_amd.synth import gen_conformers
    from oracle import escn_ref as R
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pos, z, batch, y, f = gen_conformers(101, conformers)
    sizes = torch.bincount(batch).tolist()
    torch.manual_seed(23)
    net = eSCN(**CFG)
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, p in net.named_parameters():
        if p.requires_grad:
            P[k].requires_grad_(True)
    del net
    times, t_start = [], time.perf_counter()
    while True:                                       # first step = warm-up (lazy tables); then the median of up to 5 steps inside the budget
        t0 = time.perf_counter()
        E, F = R.forward(P, CFG, pos, z, sizes)
        R.loss(E, F, y, f).backward()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > seconds_budget or len(times) >= 6:
            break
    timed = times[1:] if len(times) > 1 else times
    dt = sorted(timed)[len(timed) // 2] / conformers
    n = len(timed)
    return {"value": 1.0 / dt, "unit": "conformer-steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{conformers} synthetic conformers ({pos.shape[0]} atoms), eSCN yaml configuration, forward + loss + backward of oracle/escn_ref.py, median of {n} steps after one warm-up step, "
                      f"torch {torch.__version__} CPU fp32, no optimizer step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32")
    a = ap.parse_args()
    out = run(a.molecules, a.steps, a.warmup, a.kernels, precision=a.precision)
    if a.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
