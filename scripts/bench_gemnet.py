"""GemNet-OC (BASELINE.json configs[2]: config/model/gemnet-oc.yaml -- 4 blocks, atom 256 / edge 512, 128 radial / 7 spherical functions, 12 A cutoffs,
neighbour caps 30 / 20 / 8, quadruplet + atom-edge + edge-atom + atom-atom interactions, direct coupled forces; AdamW(amsgrad, betas 0.9/0.95, lr 1e-3),
loss = L1(E) + 100 * L2(F), gradient-norm clip 10.0, config/gemnet-oc.yaml: batch_size 8) training-step timing on one MI355X in fp32: graphs -> forward -> loss -> backward -> AdamW, on synthetic drug-like
conformers already resident in HBM.  ``run()`` is what ``bench.py --model gemnet`` calls.

    python scripts/bench_gemnet.py [--molecules 16] [--steps 10] [--warmup 3] [--kernels] [--cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COMMON = dict(num_targets=1, num_before_skip=2, num_after_skip=2, num_concat=1, num_atom=3, num_output_afteratom=3, num_global_out_layers=2,
              regress_forces=True, direct_forces=True, use_pbc=False, scale_backprop_forces=False, enforce_max_neighbors_strictly=True,
              rbf={"name": "gaussian"}, rbf_spherical=None, envelope={"name": "polynomial", "exponent": 5}, cbf={"name": "spherical_harmonics"},
              sbf={"name": "legendre_outer"}, extensive=True, forces_coupled=True, output_init="HeOrthogonal", activation="silu", scale_file=None,
              quad_interaction=True, atom_edge_interaction=True, edge_atom_interaction=True, atom_interaction=True, scale_basis=True)
CFG = dict(COMMON, num_spherical=7, num_radial=128, num_blocks=4, emb_size_atom=256, emb_size_edge=512, emb_size_trip_in=64, emb_size_trip_out=64,
           emb_size_quad_in=32, emb_size_quad_out=32, emb_size_aint_in=64, emb_size_aint_out=64, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32,
           num_atom_emb_layers=0, cutoff=12.0, cutoff_qint=12.0, cutoff_aeaint=12.0, cutoff_aint=12.0, max_neighbors=30, max_neighbors_qint=8,
           max_neighbors_aeaint=20, max_neighbors_aint=1000)            # config/model/gemnet-oc.yaml:5-60
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense (MI355X_MICROARCH.md); the 5 PF headline includes 2:1 sparsity
HBM_PEAK_GBS = 8000.0


class Batch:
    pass


def synthetic_batch(molecules, seed, device, world=1, rank=0, size="drug", cost_model="gemnet_oc"):
    import torch
    from nabladft_amd.synth import gen_rank_conformers
    (pos, z, batch, y, f), spread = gen_rank_conformers(seed, molecules, world, rank, size, cost_model)
    molecules = int(y.shape[0])          # this rank's share of the cost-balanced global batch
    b = Batch()
    b.pos, b.z, b.batch, b.y, b.forces = pos.to(device), z.to(device), batch.to(device), y.to(device), f.to(device)
    cnt = torch.bincount(batch, minlength=molecules)
    b.ptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)]).to(device)
    b.cost_spread = spread
    return b


def build(device, seed=23):
    import torch
    from nabladft_amd.gemnet_oc import GemNetOC
    torch.manual_seed(seed)
    return GemNetOC(**CFG).to(device)


def loss_fn(E, F, b):
    import torch
    return (E - b.y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - b.forces, dim=-1).mean()          # config/model/gemnet-oc.yaml:78-85


def run(molecules=16, steps=10, warmup=3, kernels=True, device=None, seed=1, world=1, rank=0, sync=None, precision="f32", size="drug"):
    import torch
    from nabladft_amd import _lib, gemnet_oc
    gemnet_oc.set_gemm_precision(precision)
    from nabladft_amd import dist as nqdist
    from nabladft_amd.trainer import FlatParameters
    dev = device or torch.device("cuda", torch.cuda.current_device())
    net = build(dev)
    # every rank draws the same global batch of `molecules` x world conformers and keeps its cost-balanced share (dist.shard_by_cost, proxy "gemnet_oc")
    batches = [synthetic_batch(molecules, seed * 100 + k, dev, world, rank, size, "gemnet_oc") for k in range(4)]
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
        flat.clip_grad_norm_(10.0)                                     # config/gemnet-oc.yaml:19-20 (gradient_clip_val 10.0, norm)
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
    G = net.get_graphs_and_indices(batches[0])
    out = {"workload": "GemNet-OC (config/model/gemnet-oc.yaml: 4 blocks, atom 256 / edge 512, 128 rbf, 7 spherical, 12 A cutoffs, caps 30/20/8, all four extra "
                       "interactions, direct coupled forces) train step: graphs, forward, L1(E) + 100 L2(F), backward, clip 10.0, AdamW(amsgrad); synthetic ~42-atom conformers",
           "value": molecules * steps / dt, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dt / steps, "molecules_per_step": molecules, "sizes": str(size),
           "load_balance": {"cost_model": "gemnet_oc", "this_run_predicted_spread": getattr(batches[0], "cost_spread", 0.0),
                            "predicted_spread_8_ranks_10_to_90_atoms_by_conformers_per_rank": nqdist.spread_table("gemnet_oc")}, "atoms": G.N,
           "edges": {"a2a": G.Ea2a, "main": G.Em, "a2ee2a": G.Ea, "qint": G.Eq, "qint_x_main_rows": G.Tin}, "parameters": net.num_params, "_dt": dt,
           "final_loss": float(loss.detach()), "dtype": precision, "data": "synthetic",
           "parity": "pinned to the reference GemNetOC classes run on CPU (tests/golden/gemnet_*.npz); torch_scatter / torch_sparse / torch_cluster restated"}
    if kernels:
        gemnet_oc.GEMM_FLOPS[0] = 0.0
        gemnet_oc.GEMM_BYTES[0] = 0.0
        step(0)
        fwd_flops, fwd_bytes = gemnet_oc.GEMM_FLOPS[0], gemnet_oc.GEMM_BYTES[0]
        gemnet_oc.GEMM_FLOPS[0] = None
        gemnet_oc.GEMM_BYTES[0] = None
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        tot = sum(v[0] for v in prof.values()) / steps
        ks = sorted(((k, v[0] / steps, v[1] // steps) for k, v in prof.items()), key=lambda x: -x[1])
        out["device_ms_per_step_nq_kernels"] = tot
        out["kernel_ms_per_step"] = {k: [round(ms, 4), int(n)] for k, ms, n in ks[:24]}
        gemm_ms = sum(v[0] for k, v in prof.items() if v[2] > 0 or k.startswith("bf16")) / steps      # dense products (+ the per-forward bf16 weight re-pack)
        out["gemm_classes_TFLOPs"] = {k: round(v[2] / max(v[0], 1e-9) / 1e9, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[2] > 0}
        out["gemm_bf16_ms_per_step"] = sum(ms for k, ms, _ in ks if "bf16" in k)
        fl = sum(v[2] for v in prof.values()) / steps                      # exact: every dense launcher records its 2 M N K (nq_profile_read2)
        by = 3.0 * fwd_bytes                                                # each of the three products reads two operands and writes one of the same sizes
        ach = fl / (max(gemm_ms, 1e-9) * 1e-3) / 1e12
        if precision == "bf16":
            # the Dense products run on bf16 MFMA (2.5 PFLOP/s dense) with fp32 activations in HBM: at these shapes (K = 256 ... 512) the compulsory bytes take
            # longer at 8 TB/s than the flops at the bf16 peak, so HBM is the roof that binds; both times are reported
            t_mfma, t_hbm = fl / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3, by / (HBM_PEAK_GBS * 1e9) * 1e3
            bound = "hbm" if t_hbm >= t_mfma else "mfma"
            achieved = by / (max(gemm_ms, 1e-9) * 1e-3) / 1e9 if bound == "hbm" else ach
            peak = HBM_PEAK_GBS if bound == "hbm" else MFMA_BF16_PEAK_TFLOPS
            out["roofline"] = {"kernel": "k_gemm_bf16 (Dense layers: bf16 MFMA, fp32 accumulate; fp32 activations in HBM)", "bound": bound, "achieved": achieved, "peak": peak,
                               "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": achieved / peak, "traffic": None, "gemm_ms_per_step": gemm_ms,
                               "flops_per_step": fl, "algorithmic_bytes_per_step": by, "mfma_bf16_bound_ms": t_mfma, "hbm_bound_ms": t_hbm,
                               "achieved_TFLOPs": ach, "frac_of_bf16_mfma_peak": ach / MFMA_BF16_PEAK_TFLOPS}
        else:
            from bench_escn import gemm_roof
            label, peak = gemm_roof("Dense layers")
            out["roofline"] = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "frac_of_exact_f32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "gemm_ms_per_step": gemm_ms, "flops_per_step": fl,
                               "algorithmic_bytes_per_step": by, "hbm_bound_ms": by / (HBM_PEAK_GBS * 1e9) * 1e3}
    gemnet_oc.set_gemm_precision("f32")
    return out


def cpu_baseline(seconds_budget=25.0, conformers=1):   # (this oracle needs ~6 s per conformer-step on the GPU box's host: one conformer keeps >= 3 timed steps inside the budget)
    """oracle/gemnet_ref.py (torch CPU, fp32; pinned to the reference classes' golden vectors) forward + loss + backward on ONE synthetic conformer."""
    import torch
    from nabladft_amd.gemnet_oc import GemNetOC
    from nabladft_amd.synth import gen_conformers
    from oracle import gemnet_ref as R
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pos, z, batch, y, f = gen_conformers(101, conformers)
    sizes = torch.bincount(batch).tolist()
    torch.manual_seed(23)
    net = GemNetOC(**CFG)
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, _ in net.named_parameters():
        if not k.endswith("scale_factor"):
            P[k].requires_grad_(True)
    for i in range(CFG["num_blocks"] + 1):                              # state_dict aliases of the shared modules must be the same tensors
        for k in list(P):
            if k.startswith(f"out_blocks.{i}.seq_energy_pre."):
                P[k] = P[k.replace(".seq_energy_pre.", ".layers.")]
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
            "sample": f"{conformers} synthetic conformers ({pos.shape[0]} atoms), GemNet-OC yaml configuration, forward + loss + backward of oracle/gemnet_ref.py (index lists "
                      f"built by Python loops, as part of the step), median of {n} steps after one warm-up step, torch {torch.__version__} CPU fp32, no optimizer step"}


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
