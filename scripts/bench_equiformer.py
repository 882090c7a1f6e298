"""EquiformerV2 (BASELINE.json configs[4], second model: config/model/equiformer_v2_oc20.yaml -- 12 blocks, lmax 6 / mmax 2, 128 sphere channels, 8 heads, cutoff 12 A,
30 neighbours; AdamW(lr 4e-4, weight_decay 1e-3), loss = 2 L1(E) + 100 L2(F); config/equiformer_v2_oc20.yaml: batch_size 2, no gradient clip) training-step timing
on one MI355X in fp32, training mode (attention dropout 0.1 and drop-path 0.05 active): graph + frames + Wigner rows -> forward -> loss -> backward -> AdamW,
on synthetic drug-like conformers already resident in HBM.

    python scripts/bench_equiformer.py [--molecules 16] [--steps 10] [--warmup 3] [--kernels] [--cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_escn import MFMA_F32_PEAK_TFLOPS, gemm_roof, synthetic_batch  # noqa: E402

CFG = dict(use_pbc=False, regress_forces=True, otf_graph=True, norm_type="layer_norm_sh", use_atom_edge_embedding=True, share_atom_edge_embedding=False,
           distance_function="gaussian", num_distance_basis=512, attn_activation="silu", use_s2_act_attn=False, use_attn_renorm=True, ffn_activation="silu",
           use_gate_act=False, use_grid_mlp=True, use_sep_s2_act=True, alpha_drop=0.1, drop_path_rate=0.05, proj_drop=0.0, weight_init="uniform",
           max_neighbors=30, max_radius=12.0, max_num_elements=65, num_layers=12, sphere_channels=128, attn_hidden_channels=64, num_heads=8,
           attn_alpha_channels=64, attn_value_channels=16, ffn_hidden_channels=128, lmax_list=[6], mmax_list=[2], num_sphere_samples=128,
           edge_channels=128)                                                              # config/model/equiformer_v2_oc20.yaml:5-41


def build(device, seed=23):
    import torch
    from nabladft_amd.equiformer_v2 import EquiformerV2_OC20
    torch.manual_seed(seed)
    return EquiformerV2_OC20(**CFG).to(device).train()


def loss_fn(E, F, b):
    import torch
    return 2.0 * (E - b.y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - b.forces, dim=-1).mean()


def run(molecules=16, steps=10, warmup=3, kernels=True, device=None, seed=1, world=1, rank=0, sync=None, precision="f32", size="drug"):
    import torch
    from nabladft_amd import _lib, gemnet_oc
    gemnet_oc.set_gemm_precision(precision)
    from nabladft_amd import dist as nqdist
    from nabladft_amd.trainer import FlatParameters
    dev = device or torch.device("cuda", torch.cuda.current_device())
    net = build(dev)
    # every rank draws the same global batch of `molecules` x world conformers and keeps its cost-balanced share (dist.shard_by_cost, proxy "equiformer_v2")
    batches = [synthetic_batch(molecules, seed * 100 + k, dev, world, rank, size, "equiformer_v2") for k in range(4)]
    flat = FlatParameters(net.parameters())
    opt = torch.optim.AdamW([flat.flat], lr=4e-4, weight_decay=1e-3)
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
    out = {"workload": "EquiformerV2 (config/model/equiformer_v2_oc20.yaml: 12 blocks, lmax 6 / mmax 2, 128 sphere channels, 8 heads, cutoff 12 A, 30 neighbours) train step "
                       "in training mode (attention dropout, drop-path): graph, frames, Wigner rows, forward, 2 L1(E) + 100 L2(F), backward, AdamW; synthetic ~42-atom "
                       "conformers",
           "value": molecules * steps / dt, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dt / steps, "molecules_per_step": molecules, "sizes": str(size),
           "load_balance": {"cost_model": "equiformer_v2", "this_run_predicted_spread": getattr(batches[0], "cost_spread", 0.0),
                            "predicted_spread_8_ranks_10_to_90_atoms_by_conformers_per_rank": nqdist.spread_table("equiformer_v2")}, "atoms": G.N, "edges": G.E,
           "parameters": net.num_params, "_dt": dt, "final_loss": float(loss.detach()), "dtype": precision, "data": "synthetic",
           "parity": "pinned to the reference EquiformerV2 classes run on CPU in eval mode (tests/golden/equiformer_*.npz); the four e3nn symbols under them are "
                     "restated (unpinned), the Wigner J matrices equal the reference's Jd.pt"}
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
        out["kernel_ms_per_step"] = {k: [round(ms, 4), int(n)] for k, ms, n in ks[:28]}
        # dense products: every launcher records its 2 M N K with the event pair (nq_profile_read2), so flops and time cover exactly the same launches
        gemm_ms = sum(v[0] for v in prof.values() if v[2] > 0) / steps
        gemm_fl = sum(v[2] for v in prof.values()) / steps
        out["gemm_ms_per_step"] = gemm_ms
        out["gemm_classes_TFLOPs"] = {k: round(v[2] / max(v[0], 1e-9) / 1e9, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[2] > 0}
        out["dense_flops_counted_per_step"] = gemm_fl
        ach = gemm_fl / (max(gemm_ms, 1e-9) * 1e-3) / 1e12
        label, peak = gemm_roof("SO(2) convolutions, radial functions, grid MLPs")
        out["roofline"] = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                           "frac_of_exact_f32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS, "gemm_ms_per_step": gemm_ms, "flops_per_step": gemm_fl}
    gemnet_oc.set_gemm_precision("f32")
    return out


def cpu_baseline(seconds_budget=25.0, conformers=2):
    """oracle/equiformer_ref.py (torch CPU, fp32; pinned to the reference classes' golden vectors) forward + loss + backward on ONE synthetic conformer."""
    import torch
    from nabladft_amd.equiformer_v2 import EquiformerV2_OC20
    from nabladft_amd.synth import gen_conformers
    from oracle import equiformer_ref as R
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pos, z, batch, y, f = gen_conformers(101, conformers)
    sizes = torch.bincount(batch).tolist()
    torch.manual_seed(23)
    net = EquiformerV2_OC20(**CFG)
    P = {k: v.detach().clone() for k, v in net.state_dict().items() if not k.startswith(("SO3_grid", "blocks.")) or k.split(".")[-1] in ("weight", "bias", "alpha_dot", "affine_weight")}
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
            "sample": f"{conformers} synthetic conformers ({pos.shape[0]} atoms), EquiformerV2 yaml configuration, forward (eval mode) + loss + backward of oracle/equiformer_ref.py, "
                      f"median of {n} steps after one warm-up step, torch {torch.__version__} CPU fp32, no optimizer step"}


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
