"""QHNet (BASELINE.json configs[3]: config/qhnet.yaml -- lmax 4, hidden 128, bottleneck 32, 5 layers, 32 radial functions, cutoff 12 bohr, def2-SVP
blocks, batch 2, AdamW(amsgrad, betas 0.9/0.95, lr 5e-4), EMA 0.9999) training-step timing on one MI355X:
forward -> packed Hamiltonian -> HamiltonianLoss -> backward -> AdamW -> EMA update, on synthetic drug-like conformers (positions in bohr)
already resident in HBM.  ``run()`` is what ``bench.py --model qhnet`` and the default bench record's ``hamiltonian`` leg call.

    python scripts/bench_qhnet.py [--molecules 2] [--steps 10] [--warmup 3] [--kernels] [--cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ORBITALS = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 7: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2], 9: [0, 0, 0, 1, 1, 2],
            16: [0, 0, 0, 0, 1, 1, 1, 2], 17: [0, 0, 0, 0, 1, 1, 1, 2], 35: [0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]}
CFG = dict(in_node_features=1, sh_lmax=4, hidden_size=128, bottle_hidden_size=32, num_gnn_layers=5, max_radius=12, num_nodes=83, radius_embed_dim=32)
BOHR = 1.8897261246
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3


class Batch:
    pass


def synthetic_batch(molecules, seed, device, world=1, rank=0, size="drug", cost_model="qhnet"):
    """``molecules`` drug-like conformers of the bench generator (nabladft_amd/synth.py, Angstrom) in bohr, as a PyG-style batch."""
    import torch
    from nabladft_amd.synth import gen_rank_conformers
    (pos, z, batch, _, _), spread = gen_rank_conformers(seed, molecules, world, rank, size, cost_model)
    molecules = int(torch.bincount(batch).shape[0])          # this rank's share of the cost-balanced global batch
    b = Batch()
    b.pos, b.z, b.batch = (pos * BOHR).to(device), z.to(device), batch.to(device)
    cnt = torch.bincount(batch, minlength=molecules)
    b.ptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)]).to(device)
    b.num_nodes = int(pos.shape[0])
    b.cost_spread = spread
    return b


def build(device, seed=23):
    import torch
    from nabladft_amd.qhnet import QHNet
    torch.manual_seed(seed)
    return QHNet(**CFG, orbitals=ORBITALS).to(device)


def gemm_flops_per_step(net, N, E, P):
    """Dense fp32 MFMA work of one training step (forward + input gradient + weight gradient = 3x the forward flops) of the per-pair / per-edge
    weight generators and heads, the GEMMs SURVEY.md 8(d) prices: 2 * rows * in * out per forward application."""
    C, Cb = net.hs, net.hbs
    f = 0.0
    for layer in net.e3_gnn_layer:
        wn = layer.conv.fc_node.hs[2]
        f += 2.0 * E * 32 * wn * 2                                      # fc_node.layer1 + layer_l0.layer1: [E,32] x [32, paths*C]
    for _ in net.e3_gnn_node_pair_layer:
        f += 2.0 * P * C * 65 * C * 2                                   # fc_node_pair.layer1 + fc.2: [P,128] x [128, 8320]
    f += 2.0 * P * C * net.expand_ij["hamiltonian"].num_path_weight     # fc_ij.2
    return 3.0 * f


def pmc_traffic_bytes(kernel_prefix, molecules):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/r0N_pmc_traffic_qhnet.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), only if it was taken at this batch size."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic_qhnet.json") for r in (9, 8, 7, 6, 5, 4, 3, 2)) if os.path.exists(q)), None)
    if path is None:
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("batch") != molecules:
        return None
    for name, v in rec.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            return 1024.0 * (2.0 * v.get("fetch_kb_per_launch", 0.0) + v.get("write_kb_per_launch", 0.0))
    return None


_KERNEL_OF = {"qh_tp_uuu_fwd": "k_qh_tp<0, false, false", "qh_tp_uuu_bwd": "k_qh_tp<0, false, true", "qh_tp_uvu_fwd": "k_qh_tp<1, true, false",
              "qh_tp_uvu_bwd": "k_qh_tp<1, true, true", "qh_exp_fwd": "k_qh_exp_fwd", "qh_exp_bwd": "k_qh_exp_bwd"}


def run(molecules=2, steps=10, warmup=3, kernels=True, device=None, seed=1, world=1, rank=0, sync=None, size="drug"):
    """One rank's share of the job: ``molecules`` conformers per step on this GPU; with world > 1 the flat gradient is all-reduced (mean) each step
    (conformers are independent graphs: data-parallel, no other collective).  Returns the record with THIS rank's wall time in ``_dt``."""
    import torch
    from nabladft_amd import _lib
    from nabladft_amd.ema import ExponentialMovingAverage
    from nabladft_amd.hamiltonian import HamiltonianLoss
    from nabladft_amd.trainer import FlatParameters
    dev = device or torch.device("cuda", torch.cuda.current_device())
    net = build(dev)
    from nabladft_amd import dist as nqdist
    # every rank draws the same global batch of `molecules` x world conformers and keeps its cost-balanced share (dist.shard_by_cost, proxy "qhnet")
    batches = [synthetic_batch(molecules, seed * 100 + k, dev, world, rank, size, "qhnet") for k in range(4)]
    flat = FlatParameters(net.parameters())
    opt = torch.optim.AdamW([flat.flat], lr=5e-4, betas=(0.9, 0.95), amsgrad=True)
    from nabladft_amd.trainer import OverlappedAllReduce
    ov = OverlappedAllReduce(flat) if world > 1 else None     # gradient buckets reduced on a side stream while the backward is still running
    ema = ExponentialMovingAverage([flat.flat], decay=0.9999)
    loss_fn = HamiltonianLoss()
    targets = []
    with torch.no_grad():
        for b in batches:
            h = net(b, packed=True)
            g = torch.Generator(device="cpu").manual_seed(b.num_nodes)
            targets.append((h + 0.05 * torch.randn(h.shape, generator=g).to(dev)).detach())

    def step(i):
        b, t = batches[i % len(batches)], targets[i % len(batches)]
        flat.zero_grad()
        loss = loss_fn(net(b, packed=True), t)
        loss.backward()
        if ov is not None:
            ov.finish()                                   # bucketed all-reduce started by the hooks during backward; mean over ranks
        opt.step()
        ema.update()
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
    b0 = batches[0]
    N, E, P = b0.num_nodes, int(b0.edge_index.shape[1]), int(b0.full_edge_index.shape[1])
    out = {"workload": "QHNet (config/qhnet.yaml: lmax 4, hidden 128, bottleneck 32, 5 layers, 32 rbf, cutoff 12 bohr, def2-SVP blocks) train step: graphs, "
                       "forward, HamiltonianLoss on the packed blocks, backward, AdamW(amsgrad), EMA; synthetic ~42-atom conformers in bohr",
           "value": molecules * steps / dt, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dt / steps, "molecules_per_step": molecules, "sizes": str(size),
           "load_balance": {"cost_model": "qhnet", "this_run_predicted_spread": getattr(batches[0], "cost_spread", 0.0),
                            "predicted_spread_8_ranks_10_to_90_atoms_by_conformers_per_rank": nqdist.spread_table("qhnet")}, "atoms": N,
           "edges_within_cutoff": E, "ordered_pairs": P, "orbitals": int(net.last_plan.m_total), "parameters": net.get_number_of_parameters(),
           "_dt": dt, "final_loss": float(loss.detach()), "dtype": "f32", "data": "synthetic", "parity": "pinned to the reference QHNet classes; e3nn arithmetic restated (unpinned)"}
    if kernels:
        _lib.profile_enable(True)
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        tot = sum(v[0] for v in prof.values()) / steps
        ks = sorted(((k, v[0] / steps, v[1] // steps) for k, v in prof.items()), key=lambda x: -x[1])
        out["device_ms_per_step_nq_kernels"] = tot
        out["kernel_ms_per_step"] = {k: [round(ms, 4), int(n)] for k, ms, n in ks[:16]}
        # roofline of the dominant class.  The [P, 128] x [128, 8320] generators are MFMA-bound; the Clebsch-Gordan kernels stream the
        # per-pair weight arrays: algorithmic bytes = weights read (2 factors) + irreps in / out per row.
        gemm_ms = sum(v[0] for v in prof.values() if v[2] > 0) / steps      # every dense launcher (incl. the row-mapped spherical linears) records 2 M N K
        out["gemm_classes_TFLOPs"] = {k: round(v[2] / max(v[0], 1e-9) / 1e9, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[2] > 0}
        fl = sum(v[2] for v in prof.values()) / steps
        dom, dom_ms, dom_n = ks[0]
        C = net.hs
        # per-launch bytes.  COMPULSORY (what must cross HBM however the step is fused) = the irreps rows read and written per (pair / edge) row;
        # the per-row weight factors ([rows, paths * C], written by the generator GEMMs and re-read here, plus their adjoints in the reverse kernels)
        # are MATERIALISED INTERMEDIATES of the current, unfused formulation: reported separately, never part of `achieved` / `frac`.
        comp = {"qh_tp_uuu_fwd": P * (3 * 25 * C) * 4.0, "qh_tp_uuu_bwd": P * (5 * 25 * C) * 4.0,
                "qh_tp_uvu_fwd": E * (2 * 25 * C) * 4.0, "qh_tp_uvu_bwd": E * (3 * 25 * C) * 4.0,
                "qh_exp_fwd": P * (800 + 1024) * 4.0, "qh_exp_bwd": P * (1600 + 1024) * 4.0}
        mat = {"qh_tp_uuu_fwd": P * (2 * 65 * C) * 4.0, "qh_tp_uuu_bwd": P * (4 * 65 * C) * 4.0, "qh_tp_uvu_fwd": E * (2 * 42 * C) * 4.0,
               "qh_tp_uvu_bwd": E * (4 * 42 * C) * 4.0, "qh_exp_fwd": P * 8320 * 4.0, "qh_exp_bwd": P * 2 * 8320 * 4.0}
        if dom in comp:
            per_launch = comp[dom]
            avg_ms = dom_ms / max(dom_n, 1)
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            traffic = pmc_traffic_bytes(_KERNEL_OF.get(dom, dom), molecules)
            out["roofline"] = {"kernel": _KERNEL_OF.get(dom, dom) + ">", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                               "algorithmic_bytes_per_launch": per_launch, "materialised_intermediate_bytes_per_launch": mat[dom],
                               "frac_incl_materialised_intermediates": (per_launch + mat[dom]) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "avg_launch_ms": avg_ms, "launches_per_step": dom_n}
        else:
            ach = fl / (max(gemm_ms, 1e-9) * 1e-3) / 1e12
            from bench_escn import gemm_roof
            label, peak = gemm_roof("weight generators [rows,128]x[128,8320] and heads")
            out["roofline"] = {"kernel": label, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "frac_of_exact_f32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "gemm_ms_per_step": gemm_ms, "flops_per_step": fl}
        out["gemm_tflops"] = fl / (max(gemm_ms, 1e-9) * 1e-3) / 1e12
    return out


def cpu_baseline(seconds_budget=30.0, atoms=None, conformers=1):   # (this oracle needs ~6 s per conformer-step on the GPU box's host: one conformer keeps >= 3 timed steps inside the budget)
    """oracle/qhnet_ref.py (pure torch CPU, fp32) forward + loss + backward on ONE synthetic conformer of the same generator."""
    import torch
    from nabladft_amd.synth import gen_conformers
    from oracle import qhnet_ref as Q
    from oracle.qhnet_params import make_state
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pos, z, batch, _, _ = gen_conformers(101, conformers)
    pos = pos * BOHR
    cnt = torch.bincount(batch)
    ptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])
    net_names = None
    from nabladft_amd.qhnet import QHNet
    m = QHNet(**CFG, orbitals=ORBITALS)
    P = {k: v.requires_grad_(True) for k, v in make_state([(k, tuple(p.shape)) for k, p in m.named_parameters()], 7).items()}
    del m, net_names
    times, t_start = [], time.perf_counter()
    while True:                                       # first step = warm-up; then the median of up to 5 steps inside the budget
        t0 = time.perf_counter()
        H = Q.forward(P, CFG, ORBITALS, pos, z, ptr)
        loss = Q.hamiltonian_loss(H, torch.zeros_like(H), torch.ones_like(H))
        loss.backward()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > seconds_budget or len(times) >= 6:
            break
    timed = times[1:] if len(times) > 1 else times
    dt = sorted(timed)[len(timed) // 2] / conformers
    n = len(timed)
    return {"value": 1.0 / dt, "unit": "conformer-steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{conformers} synthetic conformers ({pos.shape[0]} atoms), QHNet full configuration, forward + loss + backward of oracle/qhnet_ref.py, "
                      f"median of {n} steps after one warm-up step, torch {torch.__version__} CPU fp32, no optimizer step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true")
    a = ap.parse_args()
    out = run(a.molecules, a.steps, a.warmup, a.kernels)
    if a.cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
