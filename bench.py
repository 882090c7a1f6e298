#!/usr/bin/env python
"""bench.py -- conformer-steps/s of one PaiNN (PaiNN-OC config) training step on MI355X.

A step = neighbour list + forward energies + forces (adjoint sweep) + L1/L2 loss + backward to the
parameter gradients (tangent + dual-reverse sweeps) + gradient all-reduce (RCCL, N>1) + clip + AdamW,
on a batch of synthetic drug-like conformers already resident in HBM.  fp32 throughout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver: RCCL needs it before the HSA runtime starts
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F, L, R, CUTOFF, KNBR = 128, 6, 100, 5.0, 100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_32x32x2_f32) dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
SPLIT_TERMS = 6                # csrc/gemm_split.h: bf16 piece products per f32 product -> 2500 / 6 = 416.7 TFLOP/s of f32-accurate products


def nqdist_mod():
    from nabladft_amd import dist as nqdist
    return nqdist


def collective_name():
    nq = nqdist_mod()
    if not nq.active():
        return "none (1 rank)"
    if nq.native_comm(create=False) is not None:
        return "rccl via the C ABI (nq_allreduce)"
    return {"nccl": "rccl via torch.distributed (backend nccl)"}.get(dist.get_backend(), dist.get_backend())


def collective_record(step=None):
    """What the gradient collective of this job is and how many ranks the collective library ITSELF sees (ncclCommCount of the native communicator, or the
    world size torch.distributed built its RCCL communicator with), so that the driver can check the rank count independently of --gpus."""
    nq = nqdist_mod()
    if not nq.active():
        return {"backend": "none", "ranks_seen": 1, "ranks_seen_source": "no process group", "path": "none", "name": collective_name(), "allreduce_exposed_ms": None}
    native = nq.native_comm(create=False) is not None
    backend = dist.get_backend()
    exposed = step.allreduce_exposed_ms() if step is not None and hasattr(step, "allreduce_exposed_ms") else None
    # ranks_seen_source says what the number is: ncclCommCount of the communicator this job reduces with (an independent check), or torch.distributed's own
    # world size (then it restates the launcher's count: ADVICE r5)
    return {"backend": "rccl" if (native or backend == "nccl") else backend, "ranks_seen": int(nq.ranks_seen()),
            "ranks_seen_source": "ncclCommCount" if native else "torch world_size", "path": "native" if native else "torch",
            "name": collective_name(), "allreduce_exposed_ms": exposed}


def check_ranks(expected):
    """Abort (exit code != 0) when the collective library does not see the number of ranks this job was launched for."""
    nq = nqdist_mod()
    seen = nq.ranks_seen() if nq.active() else 1
    if seen != expected:
        print(f"bench.py: the collective library sees {seen} ranks but --gpus is {expected}", file=sys.stderr, flush=True)
        sys.exit(3)


def dist_on():
    """A process group exists and its collectives run: world > 1, or the forced 1-rank group of the single-GPU RCCL test (NQ_DIST_FORCE=1)."""
    return nqdist_mod().active()


def make_batches(seed, n_batches, B, device):
    """n_batches distinct batches of B conformers: 64 generated molecules per batch seed, replicated with
    independent random rotations and 0.02 A jitter (keeps the generator's statistics, costs O(64) python)."""
    import nabladft_amd as nq
    from nabladft_amd.synth import gen_conformers
    out = []
    for k in range(n_batches):
        base = min(B, 64)
        pos, z, batch, _, _ = gen_conformers(seed * 1000 + k, base)
        rng = np.random.Generator(np.random.PCG64(seed * 7919 + k))
        counts = torch.bincount(batch)
        reps = (B + base - 1) // base
        P, Z, Bt = [], [], []
        for r in range(reps):
            for m in range(base):
                if r * base + m >= B:
                    break
                sel = batch == m
                q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
                p = pos[sel].numpy() @ q.T.astype(np.float32) + rng.normal(0, 0.02, size=(int(counts[m]), 3)).astype(np.float32)
                P.append(torch.tensor(p.astype(np.float32)))
                Z.append(z[sel])
                Bt.append(torch.full((int(counts[m]),), r * base + m, dtype=torch.long))
        pos_b, z_b, b_b = torch.cat(P), torch.cat(Z), torch.cat(Bt)
        y = torch.tensor(rng.normal(0, 1, size=B).astype(np.float32))
        f = torch.tensor(rng.normal(0, 0.05, size=(pos_b.shape[0], 3)).astype(np.float32))
        out.append(nq.Batch(pos_b, z_b, b_b, y, f).to(device))
    return out


def gemm_flops(name, N, E, launches):
    """Average flop per launch of a role-tagged GEMM class, e.g. 'gemm_tn:Wr[384x100]' or 'gemm_nt:W2[n=384,k=128]'."""
    kind, rest = name.split(":", 1)
    tag = rest.split("[")[0]
    dims = [int(x) for x in rest.split("[")[1].rstrip("]").replace("n=", "").replace("k=", "").replace("x", ",").split(",")]
    rows_single = {"Wr": E, "U": 3 * N, "sn:W2": E // 2}.get(tag, N)       # SchNet's filter network runs per undirected pair
    if kind == "gemm_tn":          # weight gradients: always over the stacked (primal+tangent) rows
        return 2.0 * (2 * rows_single) * dims[0] * dims[1]
    # nt / nn: 1x rows in forward / force adjoint / tangent, 2x rows in the dual reverse -> average over the step's launches
    per_layer = launches / float(L) if tag not in ("O1",) else launches
    if kind == "gemm_nt":
        mult = 1.0                                      # forward and tangent passes: single rows each
    else:
        mult = 1.5                                      # nn: force adjoint (1x) + dual reverse (2x)
    return 2.0 * rows_single * mult * dims[0] * dims[1]


# HIP-event launcher class -> rocprofv3 kernel name (for the PMC traffic file) and sweep multiplicity
_MSG = {"msgf_fwd": ("k_msgf_fwd<false", 1), "msgf_tan": ("k_msgf_fwd<true", 2), "msgf_rev_force": ("k_msgf_rev<false", 1),
        "msgf_rev_dual": ("k_msgf_rev<true", 2), "msgf_rev_dual_ng": ("k_msgf_rev_nopair<true", 2)}


# SchNet streaming kernels: bytes that must cross HBM once per launch (edge arrays [E][F] fp32, 128-B window records, node rows [N][F])
# (pair arrays have E/2 rows)
_SN = {"sn_filter1": lambda N, E: E / 2 * (F * 4 + 128.0), "sn_filter1_tan": lambda N, E: E / 2 * (F * 4 + 132.0),
       "sn_filter1_rev_force": lambda N, E: E / 2 * (F * 4 + 136.0), "sn_filter1_rev_dual": lambda N, E: E / 2 * (4 * F * 4 + 132.0),
       "sn_conv": lambda N, E: E / 2 * F * 4.0 + 2 * N * F * 4.0, "sn_conv_tan": lambda N, E: E * F * 4.0 + 3 * N * F * 4.0,
       "sn_conv_dual": lambda N, E: E * F * 4.0 + 4 * N * F * 4.0, "sn_pair_rev_force": lambda N, E: E * F * 4.0 + 2 * N * F * 4.0,
       "sn_pair_rev_dual": lambda N, E: E * F * 4.0 + 4 * N * F * 4.0}


def _kernel_in_build(name):
    """True if the kernel's identifier (e.g. 'k_msgf_rev' of 'k_msgf_rev<true, 2>') is a symbol of the libnablaq.so this run loaded."""
    ident = name.split("<")[0].strip().encode()
    so = os.path.join(ROOT, "nabladft_amd", "libnablaq.so")
    try:
        with open(so, "rb") as fh:
            return ident in fh.read()
    except OSError:
        return False


def pmc_traffic_bytes(kernel_prefix, batch):
    """(HBM bytes per launch, source) from the newest committed PMC summary (profiles/r0N_pmc_traffic.json; FETCH_SIZE doubled per the
    gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as reported).  Refused (None, reason) if the file was taken at another batch size,
    if it does not hold this kernel, or if the kernel it names is not in the library this run loaded (a stale file of an earlier build)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic.json") for r in (9, 8, 7, 6, 5, 4, 3, 2, 1)) if os.path.exists(q)), None)
    if path is None:
        return None, "no PMC summary committed"
    with open(path) as fh:
        rec = json.load(fh)
    tag = os.path.basename(path)
    if rec.get("batch") != batch:
        return None, f"{tag} was taken at batch {rec.get('batch')}"
    for name, v in rec.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            if not _kernel_in_build(name):
                return None, f"{tag} names {name}, which is not in this build"
            return 1024.0 * (2.0 * v.get("fetch_kb_per_launch", 0.0) + v.get("write_kb_per_launch", 0.0)), f"profiles/{tag} (rocprofv3 --pmc, separate passes; FETCH x2 per the gfx950 note)"
    return None, f"{tag} does not hold {kernel_prefix}"


def rocprof_avg_launch_us(kernel_prefix):
    """Average launch duration (us) of the kernel in the newest committed rocprofv3 --kernel-trace --stats summary of the PaiNN bench (profiles/r0N_rocprofv3_kernel_stats_painn_b2048.csv),
    or None: lets the record show the HIP-event figure next to the profiler's."""
    import csv
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_rocprofv3_kernel_stats_painn_b2048.csv") for r in (9, 8, 7, 6, 5)) if os.path.exists(q)), None)
    if path is None or kernel_prefix is None:
        return None, None
    ident = kernel_prefix.split(",")[0].split(" ")[0]
    try:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = (row.get("Name") or row.get("KernelName") or "").replace("void ", "")
                if name.startswith(ident):
                    avg = row.get("AverageNs") or row.get("Average")
                    return (float(avg) / 1e3 if avg else None), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        pass
    return None, os.path.relpath(path, ROOT)


def roofline_record(dom, avg_ms, launches, n_atoms, E, batch):
    rec = _roofline_record(dom, avg_ms, launches, n_atoms, E, batch)
    us, src = rocprof_avg_launch_us(rec.get("kernel"))
    rec["rocprof_avg_launch_us"], rec["rocprof_source"] = us, src
    return rec


def _roofline_record(dom, avg_ms, launches, n_atoms, E, batch):
    """Roofline entry of the dominant launcher class.  Message kernels are HBM-bound: algorithmic bytes per launch =
    SURVEY 8(d)'s message share of one layer, (8*N*F*4 + 24*E) bytes, x2 for the sweeps that carry (primal, tangent) pairs.
    GEMM classes are bound by the fp32 matrix cores: flop per launch from the role tag."""
    if dom in _MSG:
        prefix, mult = _MSG[dom]
        nbytes = mult * (8.0 * n_atoms * F * 4 + 24.0 * E)
        ach = nbytes / (avg_ms * 1e-3) / 1e9
        traffic, tsrc = pmc_traffic_bytes(prefix, batch)
        return {"kernel": f"{prefix}, {F // 64}>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": tsrc, "traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,   # measured HBM bytes per second of the launch
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": avg_ms, "launches_per_step": launches}
    if dom.startswith("gemm"):
        flops = gemm_flops(dom, n_atoms, E, launches)
        ach = flops / (avg_ms * 1e-3) / 1e12
        return {"kernel": f"k_gemm {dom}", "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches}
    if dom in _SN:
        nbytes = _SN[dom](n_atoms, E)
        ach = nbytes / (avg_ms * 1e-3) / 1e9
        return {"kernel": "k_" + dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": avg_ms, "launches_per_step": launches}
    if dom == "gwr_mol":
        # rbf_proj gradient, molecule per workgroup (csrc/molpair.hip): reads the 20 node rows of the layer once (12 primal / tangent + 8 adjoint: 20 N F floats)
        # and the per-pair records (two bf16 pieces of the matrix-core A operand = 256 B + 32 B of scalars per pair and 32-channel slice; L2 serves three of the four slices)
        nbytes = 20.0 * n_atoms * F * 4                                   # the contract figure: node rows only
        design_bytes = nbytes + (E / 2.0) * (256 + 32) * (F // 32)        # + the kernel's own per-pair record streams (before the L2 reuse across slices)
        ach = nbytes / (avg_ms * 1e-3) / 1e9
        traffic, tsrc = pmc_traffic_bytes("k_gwr_mol", batch)
        return {"kernel": "k_gwr_mol (+ k_gwr_mol_reduce)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": tsrc, "traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                "algorithmic_bytes_per_launch": nbytes, "design_bytes_per_launch": design_bytes, "avg_launch_ms": avg_ms, "launches_per_step": launches,
                "matrix_core_flops_per_launch": 2.0 * 32 * 32 * 16 * 9 * (E / 2.0 / 8.0) * (F // 32),
                "note": "bound by VALU issue and per-molecule latency (2 barriers + one exposed 100-kB row load per molecule), not by bytes: per pair and 32-channel slice "
                        "~53 VALU instructions + 9/8 v_mfma_f32_32x32x16_bf16 (bf16 hi/lo split, 8 pairs per contraction: round 6; profiles/r06_gwr_mol_variants.txt)"}
    if dom == "gwr_sorted":
        flops = 2.0 * 26 * E * 3 * F          # 26 FMAs per (edge, column)
        ach = flops / (avg_ms * 1e-3) / 1e12
        return {"kernel": "k_gwr_sorted", "bound": "hbm", "achieved": (2.0 * E * 3 * F * 4) / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (2.0 * E * 3 * F * 4) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes("k_gwr_sorted", batch)[0],
                "valu_tflops": ach, "avg_launch_ms": avg_ms, "launches_per_step": launches}
    return {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "avg_launch_ms": avg_ms, "launches_per_step": launches}


# fp32 operations per (directed edge, channel) of the message-path kernels, counted from csrc/edge.hip (FMA = 2): the 13-tap filter is 39 FMA + 3 mul
# (phi) or 78 FMA + 6 mul (phi and psi); then the per-edge message arithmetic of each flavour; the dual reverse runs its second direction on half of the edges;
# k_gwr_sorted: 26 FMA per (pair, column) = 13 per directed edge and column, three column parts.
_MSG_FLOP_PER_EDGE_CHANNEL = {"msgf_fwd": 81 + 16, "msgf_tan": 162 + 39, "msgf_rev_force": 162 + 39, "msgf_rev_dual": 162 + 103 + 27, "gwr_sorted": 3 * 26,
                              "msgf_rev_dual_ng": 162 + 76, "gwr_mol": 27 + 3 * 26}   # without pair rows the dual flavour loses the second direction; k_gwr_mol recomputes gphi / gpsi (~54 flop per directed edge and channel) and contracts 26 FMA per (pair, column)


def gemm_engine_name():
    return "exact-f32" if os.environ.get("NQ_GEMM_F32", "0") not in ("", "0") else "split-bf16"


def gemm_accuracy_probe(dev, restore_variant=1):
    """max |C - C64| / max |C64| of the forward product [40000 x 128] x [128 x 256]^T (626 tiles: split engine by default) on both engines, C64 = torch float64."""
    from nabladft_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    M, N, K = 40000, 256, 128
    A, W = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.1).to(dev)
    ref = A.double() @ W.double().T
    out = {"shape": [M, N, K]}
    try:
        for name, var in (("split_bf16_engine", 1), ("exact_f32_engine", 1 | 32)):
            lib.nq_set_gemm_variant(var)
            Cd = torch.empty(M, N, device=dev)
            _lib.check(lib.nq_linear_forward(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(Cd), None, M, N, K, _lib.stream_ptr()))
            out[name] = float((Cd.double() - ref).abs().max() / ref.abs().max())
    finally:
        lib.nq_set_gemm_variant(restore_variant)
    return out


# Compulsory HBM traffic of the node-level kernels of the PaiNN step, in units of one [N][F] fp32 array per launch (what the kernel must read + write once;
# csrc/node.hip).  "single" = forward / force-adjoint / tangent flavour, "dual" = the (primal, tangent) flavour of the dual reverse sweep.
_NODE_NF = {"upd_a": 6 + 1 + 1 + 2, "upd_b": 1 + 3 + 1 + 3 + 3 + 4, "silu_rev": 4 + 2, "silu_tan": 3}   # silu_rev: the dual flavour (Z, TZ, G, GT in; G, GT out); the single flavours live in GEMM epilogues


def kernel_table(kernels, prof, steps, n_atoms, E):
    """Every launcher class of the step against BOTH roofs where a byte / flop count is defined: message kernels and node kernels by their compulsory HBM bytes,
    dense products by 2MNK (recorded by the launchers) and by the bytes of their row operands (A read once, C written once; weights are L2-resident)."""
    NF = n_atoms * F * 4.0
    rows = []
    for name, ms, n in kernels:
        r = {"name": name, "ms_per_step": round(ms, 4), "launches_per_step": int(n)}
        sec = ms * 1e-3
        if name in _MSG:
            b = _MSG[name][1] * (8.0 * n_atoms * F * 4 + 24.0 * E) * n
            r.update(bound="hbm", GBps=b / sec / 1e9, frac=b / sec / 1e9 / HBM_PEAK_GBS)
        elif name == "gwr_sorted":
            b = (2.0 * (E / 2) * 3 * F * 4 + 128.0 * E / 2) * n          # one g_phi and one g_psi row per pair + the window record
            r.update(bound="hbm", GBps=b / sec / 1e9, frac=b / sec / 1e9 / HBM_PEAK_GBS)
        elif name == "upd_rev":
            b = (22 + 44 + 11 + 22) * NF * (n / 4.0)                     # rev1 + rev2, dual and single flavours: 4 launches per layer
            r.update(bound="hbm", GBps=b / sec / 1e9, frac=b / sec / 1e9 / HBM_PEAK_GBS)
        elif name in _NODE_NF:
            b = _NODE_NF[name] * NF * n * (1.5 if name in ("upd_a", "upd_b") else 1.0)   # forward (1x) + tangent (2x operands) flavours averaged
            r.update(bound="hbm", GBps=b / sec / 1e9, frac=b / sec / 1e9 / HBM_PEAK_GBS)
        elif name.startswith("gemm") and prof is not None and name in prof and prof[name][2] > 0:
            fl = prof[name][2] / steps
            dims = [int(x) for x in name.split("[")[1].rstrip("]").replace("n=", "").replace("k=", "").replace("x", ",").split(",")]
            d0, d1 = dims[0], dims[1]
            mrows = fl / (2.0 * d0 * d1)                                  # rows streamed per step by this class
            b = 4.0 * mrows * (d0 + d1)
            r.update(bound="mfma/hbm", TFLOPs=fl / sec / 1e12, frac_mfma_f32=fl / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, GBps=b / sec / 1e9,
                     frac_hbm=b / sec / 1e9 / HBM_PEAK_GBS)
        rows.append({k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in r.items()})
    return rows


def step_bounds(kernels, n_atoms, E, batch, ms_per_step, prof=None, steps=1, engine=None):
    """Step-level roofs of the PaiNN training step: the arithmetic the step executes (GEMM flops recorded by every dense launcher + the message-path kernels'
    VALU flops) against the pipes it runs on, and SURVEY 8(d)'s 17.8 MB / conformer-step of compulsory HBM traffic against 8 TB/s.  The larger time binds.
    Pipes: message path = f32 VALU (157.3 TFLOP/s; the exact-f32 MFMA has the same peak).  Dense products: the exact-f32 engine = 157.3; the split-bf16
    engine (default for the large launches: every f32 value split exactly into three bf16 pieces, six piece products per product, f32 accumulation) =
    2500 / 6 = 416.7 TFLOP/s of f32-accurate products.  Both roofs are reported; `binding_roof` uses the engine that ran."""
    engine = engine or gemm_engine_name()
    if prof is not None:      # exact: every dense launcher records its 2 M N K with the event pair (nq_profile_read2)
        gemm = sum(v[2] for v in prof.values()) / steps
    else:
        gemm = sum(gemm_flops(k, n_atoms, E, n) * n for k, _, n in kernels if k.startswith("gemm"))
    msg = sum(_MSG_FLOP_PER_EDGE_CHANNEL[k] * float(E) * F * n for k, _, n in kernels if k in _MSG_FLOP_PER_EDGE_CHANNEL)
    t_fp32 = (gemm + msg) / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3
    t_split = (gemm / (MFMA_BF16_PEAK_TFLOPS / SPLIT_TERMS * 1e12) + msg / (MFMA_F32_PEAK_TFLOPS * 1e12)) * 1e3
    t_arith = t_fp32 if engine == "exact-f32" else t_split
    t_hbm = 17.8e6 * batch / (HBM_PEAK_GBS * 1e9) * 1e3
    bind = "arithmetic" if t_arith >= t_hbm else "hbm"
    return {"flops_per_step": gemm + msg, "gemm_flops_per_step": gemm, "message_valu_flops_per_step": msg, "gemm_engine": engine,
            "fp32_peak_TFLOPs": MFMA_F32_PEAK_TFLOPS, "split_bf16_peak_TFLOPs_of_f32_products": MFMA_BF16_PEAK_TFLOPS / SPLIT_TERMS,
            "fp32_bound_ms": t_fp32, "split_engine_bound_ms": t_split, "hbm_bound_ms": t_hbm, "binding_roof": bind,
            "achieved_TFLOPs": (gemm + msg) / (ms_per_step * 1e-3) / 1e12,
            "frac_of_fp32_roof": t_fp32 / ms_per_step, "frac_of_split_engine_roof": t_split / ms_per_step, "frac_of_hbm_roof": t_hbm / ms_per_step,
            "frac_of_binding_roof": max(t_arith, t_hbm) / ms_per_step,
            "note": "the step is arithmetic-bound: the HBM fraction cannot exceed hbm_bound_ms / (arithmetic bound)"}


def _graph_replay(which):
    """scripts/bench_graphed.py: the step at the reference's batch size on one PREPARED batch, eager and as a replayed HIP graph (is the step host-bound?)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gc
    import bench_graphed
    gc.collect()                                           # no autograd graph of an earlier eager step may be alive during the capture
    try:
        return bench_graphed.run(which)
    except Exception as e:                                 # a failed capture must not cost the record its other numbers
        return {"error": repr(e)[:300]}


def bench_gemnet(args, rank, world, local_dev, dev):
    """--model gemnet: BASELINE.json configs[2] (config/model/gemnet-oc.yaml) through scripts/bench_gemnet.py; same JSON contract, conformer-steps/s, fp32."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_gemnet as BG

    def sync():
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier(device_ids=[local_dev]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    mol = args.batch if args.batch != 2048 else 16
    rec = BG.run(mol, args.steps, args.warmup, kernels=not args.no_roofline and rank == 0 and world == 1, device=dev, world=world, rank=rank, sync=sync)
    t = torch.tensor([rec.pop("_dt")], device=dev, dtype=torch.float64)
    if dist_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    bf = b8 = big = None
    if world == 1 and not args.no_roofline:
        b8 = BG.run(8, args.steps, args.warmup, kernels=False, device=dev) if mol != 8 else None
        big = {}
        for prec in ("f32", "bf16"):
            r64 = BG.run(64, 4, 3, kernels=False, device=dev, precision=prec)
            big[prec] = {k: r64[k] for k in ("value", "unit", "ms_per_step", "atoms")}
        bf = BG.run(mol, args.steps, args.warmup, kernels=True, device=dev, precision="bf16")
        bf = {"what": "same step with the Dense products (forward, input and weight gradients) on bf16 MFMA, fp32 accumulation, fp32 sums over edges / triplets / "
                      "quadruplets, fp32 master weights and optimizer -- the mode BASELINE.json names for this configuration; not parity-grade (operands rounded to bf16)",
              "value": bf["value"], "unit": bf["unit"], "ms_per_step": bf["ms_per_step"], "final_loss": bf["final_loss"], "roofline": bf.get("roofline"),
              "kernel_ms_per_step": bf.get("kernel_ms_per_step")}
    if rank == 0:
        cpu = BG.cpu_baseline() if world == 1 and not args.no_cpu_baseline else None
        out = {"metric": "conformer-steps/sec (fwd+bwd) + MAE(E,F) vs CPU reference", "value": mol * world * args.steps / dt, "unit": "conformer-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": dtype_string(), "data": "synthetic",
               "config": {"workload": rec.pop("workload") + f"; {mol} conformers/GPU/step", "conformers_per_gpu": mol, "atoms_per_step_per_gpu": rec["atoms"],
                          "edges": rec["edges"], "parallelism": f"dp{world}"},
               "final_loss": rec["final_loss"], "roofline": rec.get("roofline"), "cpu_baseline": cpu, "kernel_ms_per_step": rec.get("kernel_ms_per_step"), "gemm_classes_TFLOPs": rec.get("gemm_classes_TFLOPs"),
               "parity": rec.get("parity"), "bf16_mode": bf,
               "reference_batch_size_8": None if b8 is None else {k: b8[k] for k in ("value", "unit", "ms_per_step", "atoms")},
               "reference_batch_size_8_prepared_eager_vs_graph_replay": _graph_replay("gemnet") if world == 1 and not args.no_roofline else None,
               "batch_64": big if bf is not None else None}
        emit(out)
    finish(local_dev)


def bench_escn(args, rank, world, local_dev, dev, which="escn"):
    """--model escn / equiformer: BASELINE.json configs[4] (config/model/escn-oc.yaml, config/model/equiformer_v2_oc20.yaml) through scripts/bench_escn.py /
    scripts/bench_equiformer.py; same JSON contract, conformer-steps/s, fp32."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_escn
    import bench_equiformer
    BE = bench_escn if which == "escn" else bench_equiformer
    ref_batch = 8 if which == "escn" else 2                       # config/escn-oc.yaml:11, config/equiformer_v2_oc20.yaml:11

    def sync():
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier(device_ids=[local_dev]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    mol = args.batch if args.batch != 2048 else 16
    rec = BE.run(mol, args.steps, args.warmup, kernels=not args.no_roofline and rank == 0 and world == 1, device=dev, world=world, rank=rank, sync=sync)
    t = torch.tensor([rec.pop("_dt")], device=dev, dtype=torch.float64)
    if dist_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    b8 = BE.run(ref_batch, args.steps, args.warmup, kernels=False, device=dev) if world == 1 and not args.no_roofline and mol != ref_batch else None
    bf = None
    if world == 1 and not args.no_roofline:
        bf = BE.run(mol, args.steps, args.warmup + 2, kernels=False, device=dev, precision="bf16")
        bf = {"what": "same step with the bias-free Dense products (SO(2) convolutions, grid MLP) on bf16 MFMA, fp32 accumulation; not parity-grade",
              "value": bf["value"], "unit": bf["unit"], "ms_per_step": bf["ms_per_step"], "final_loss": bf["final_loss"], "roofline": bf.get("roofline"),
              "kernel_ms_per_step": bf.get("kernel_ms_per_step")}
    if rank == 0:
        cpu = BE.cpu_baseline() if world == 1 and not args.no_cpu_baseline else None
        out = {"metric": "conformer-steps/sec (fwd+bwd) + MAE(E,F) vs CPU reference", "value": mol * world * args.steps / dt, "unit": "conformer-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": dtype_string(), "data": "synthetic",
               "config": {"workload": rec.pop("workload") + f"; {mol} conformers/GPU/step", "conformers_per_gpu": mol, "atoms_per_step_per_gpu": rec["atoms"],
                          "edges": rec["edges"], "parallelism": f"dp{world}"},
               "final_loss": rec["final_loss"], "roofline": rec.get("roofline"), "cpu_baseline": cpu, "kernel_ms_per_step": rec.get("kernel_ms_per_step"), "gemm_classes_TFLOPs": rec.get("gemm_classes_TFLOPs"),
               "parity": rec.get("parity"), "bf16_mode": bf,
               f"reference_batch_size_{ref_batch}": None if b8 is None else {k: b8[k] for k in ("value", "unit", "ms_per_step", "atoms")},
               f"reference_batch_size_{ref_batch}_prepared_eager_vs_graph_replay": _graph_replay(which) if world == 1 and not args.no_roofline else None}
        emit(out)
    finish(local_dev)


def bench_qhnet(args, rank, world, local_dev, dev):
    """--model qhnet: BASELINE.json configs[3] (config/qhnet.yaml) through scripts/bench_qhnet.py; same JSON contract, conformer-steps/s."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_qhnet as BQ

    def sync():
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier(device_ids=[local_dev]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    mol = args.batch if args.batch != 2048 else 16
    rec = BQ.run(mol, args.steps, args.warmup, kernels=not args.no_roofline and rank == 0 and world == 1, device=dev, world=world, rank=rank, sync=sync)
    t = torch.tensor([rec.pop("_dt")], device=dev, dtype=torch.float64)
    if dist_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        small = BQ.run(2, args.steps, args.warmup, kernels=False, device=dev) if world == 1 and not args.no_roofline and mol != 2 else None
        cpu = BQ.cpu_baseline() if world == 1 and not args.no_cpu_baseline else None
        out = {"metric": "conformer-steps/sec (fwd+bwd) + MAE(H) vs CPU reference", "value": mol * world * args.steps / dt, "unit": "conformer-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": dtype_string(), "data": "synthetic",
               "config": {"workload": rec.pop("workload") + f"; {mol} conformers/GPU/step", "conformers_per_gpu": mol, "atoms_per_step_per_gpu": rec["atoms"],
                          "ordered_pairs": rec["ordered_pairs"], "edges_within_cutoff": rec["edges_within_cutoff"], "parallelism": f"dp{world}"},
               "final_loss": rec["final_loss"], "roofline": rec.get("roofline"), "cpu_baseline": cpu, "kernel_ms_per_step": rec.get("kernel_ms_per_step"), "gemm_classes_TFLOPs": rec.get("gemm_classes_TFLOPs"),
               "gemm_tflops": rec.get("gemm_tflops"), "reference_batch_size_2": None if small is None else {k: small[k] for k in ("value", "unit", "ms_per_step", "atoms", "ordered_pairs")},
               "reference_batch_size_2_prepared_eager_vs_graph_replay": _graph_replay("qhnet") if world == 1 and not args.no_roofline else None}
        emit(out)
    finish(local_dev)


COMPACT_LIMIT = 4000     # bytes: the driver parses the LAST stdout line; round 3's 23-KB line was not parsed
DTYPE_SPLIT = ("f32 (dense products >= 192 tiles: f32 values split exactly into 3 bf16 pieces, 6 MFMA piece products, f32 accumulate; "
               "non-finite operands give NaN where plain f32 gives inf)")


def dtype_string():
    return "f32" if os.environ.get("NQ_GEMM_F32", "0") not in ("", "0") else DTYPE_SPLIT


def _short(x, n=160):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def _num(x):
    return float(f"{x:.6g}") if isinstance(x, float) else x


def _pick(d, keys):
    return None if d is None else {k: _num(_short(d[k])) for k in keys if k in d and d[k] is not None}


def compact_record(full, full_path=None):
    """The one line the driver parses: the contract fields + roofline + cpu_baseline + accuracy, < COMPACT_LIMIT bytes.  Everything else
    (other models' legs, kernel tables, engine probes) stays in the full record (gpurun_out/bench_full.json, copied to profiles/)."""
    rec = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                          "dtype", "data")}
    cfg = dict(full.get("config") or {})
    cfg["workload"] = _short(cfg.get("workload", ""), 300)
    rec["config"] = {k: _num(v) for k, v in cfg.items()}
    if full.get("final_loss") is not None:
        rec["final_loss"] = _num(full["final_loss"])
    rf = full.get("roofline")
    if rf is not None:
        r = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "avg_launch_ms",
                       "rocprof_avg_launch_us", "rocprof_source", "launches_per_step", "device_ms_per_step_all_kernels"))
        r.setdefault("traffic", None)
        if rf.get("step") is not None:
            r["step"] = _pick(rf["step"], ("flops_per_step", "gemm_flops_per_step", "message_valu_flops_per_step", "gemm_engine", "fp32_bound_ms",
                                           "split_engine_bound_ms", "hbm_bound_ms", "binding_roof", "achieved_TFLOPs", "frac_of_binding_roof", "frac_of_hbm_roof"))
        rec["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb is not None:
        rec["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "host_cpus", "kind", "sample"))
        if cb.get("all_cores"):
            rec["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores")) if cb["all_cores"].get("value") is not None else "--full only"
    if full.get("mae_vs_cpu_reference") is not None:
        mv = full["mae_vs_cpu_reference"]
        rec["mae_vs_cpu_reference"] = {k: (_num(v) if not isinstance(v, (str, dict)) else v) for k, v in mv.items() if k != "small_batch"}
        if isinstance(mv.get("small_batch"), dict):   # the B = 32 sample (small-batch kernels): the three headline errors only
            rec["mae_vs_cpu_reference"]["small_batch"] = {k: _num(mv["small_batch"][k]) for k in ("conformers", "max_rel_energy", "max_rel_forces", "max_rel_grad") if k in mv["small_batch"]}
    for extra in ("reference_batch_size_32", "reference_batch_size_8", "reference_batch_size_2"):
        if full.get(extra) is not None:
            rec[extra] = _pick(full[extra], ("value", "unit", "ms_per_step"))
    if full.get("sibling_config") is not None:      # BASELINE.json configs[1] as literally written (config/painn.yaml: schnetpack PaiNN), one number
        sc = full["sibling_config"]
        rec["sibling_config"] = {"workload": _short(sc.get("workload", ""), 60), "value": _num(sc.get("value")), "ms_per_step": _num(sc.get("ms_per_step")),
                                 "parity": "unpinned" if "unpinned" in str(sc.get("parity", "")) else "pinned"}
    if full.get("sustained") is not None:
        rec["sustained"] = _pick(full["sustained"], ("value", "unit", "steps", "seconds"))
    if full.get("kernel_ms_per_step"):
        rec["kernel_ms_per_step"] = dict(list(full["kernel_ms_per_step"].items())[:6])
    if full_path:
        rec["full_record"] = full_path
    for drop in ("kernel_ms_per_step", "sustained", "reference_batch_size_32", "reference_batch_size_8", "reference_batch_size_2", "mae_vs_cpu_reference"):
        if len(json.dumps(rec)) <= COMPACT_LIMIT:
            break
        rec.pop(drop, None)
    assert len(json.dumps(rec)) <= COMPACT_LIMIT, len(json.dumps(rec))
    return rec


def emit(full):
    """Full record -> gpurun_out/bench_full.json; compact record -> the last stdout line."""
    path = os.path.join("gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, path), "w") as fh:
            json.dump(full, fh)
    except OSError:
        path = None
    global _LINE
    _LINE = json.dumps(compact_record(full, path))


_LINE = None


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def finish(local_dev):
    """Print the record as the LAST line of the job's stdout.  librccl printf()s a line ("Librccl path : ...") into the C stdio buffer of every rank, which
    a piped stdout only flushes at process exit -- after Python's prints (seen on the GPU box: the record was the second-to-last line).  So: every rank
    flushes its C buffers, the ranks meet at a barrier, the group is torn down, buffers are flushed again, the other ranks exit, and rank 0 prints last."""
    multi = False
    if dist_on():
        multi = dist.get_world_size() > 1
        _flush_c_stdio()
        dist.barrier(device_ids=[local_dev]) if dist.get_backend() == "nccl" else dist.barrier()
        nqdist_mod().drop_native_comm()
        dist.destroy_process_group()
    sys.stdout.flush()
    _flush_c_stdio()
    if _LINE is not None:
        if multi:
            time.sleep(1.5)          # the other ranks are exiting (their exit flushes anything a library buffered after the barrier)
        print(_LINE, flush=True)


WORKLOADS = {
    "painn-oc": "PaiNN-OC (nablaDFT/painn_pyg, config/model/painn-oc.yaml: F=128 L=6 R=100 rc=5A K=100) energy+forces train step incl. "
                "neighbour list, L1+L2 loss, grad all-reduce, clip 5.0, AdamW lr 5e-4",
    "schnet-spk": "SchNet (config/schnet.yaml -> schnetpack SchNet F=128 L=6 R=100 rc=5A cosine cutoff, Atomwise+Forces; restated, parity unpinned) "
                  "energy+forces train step incl. neighbour list, MSE+MSE loss, grad all-reduce, no clip, AdamW lr 1e-4 wd 0.01",
    "qhnet": "QHNet (config/qhnet.yaml) Hamiltonian training step -- see scripts/bench_qhnet.py",
    "gemnet": "GemNet-OC (config/model/gemnet-oc.yaml) energy + direct forces training step -- see scripts/bench_gemnet.py",
    "escn": "eSCN (config/model/escn-oc.yaml) energy + direct forces training step -- see scripts/bench_escn.py",
    "equiformer": "EquiformerV2 (config/model/equiformer_v2_oc20.yaml) energy + direct forces training step -- see scripts/bench_equiformer.py",
    "painn-spk": "PaiNN (config/painn.yaml -> schnetpack PaiNN F=128 L=6 R=100 rc=5A cosine cutoff, Atomwise+Forces; restated, parity unpinned) "
                 "energy+forces train step incl. neighbour list, MSE+MSE loss, grad all-reduce, no clip, AdamW lr 1e-4 wd 0.01",
}


def build_step(kind, dev):
    """(model, FusedTrainStep) for a workload with its config's optimiser settings."""
    import nabladft_amd as nq
    if kind == "painn-oc":
        model = nq.PaiNN(F, L, R, CUTOFF, KNBR, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100).to(dev)
        return model, nq.FusedTrainStep(model, lr=5e-4, weight_decay=0.0, max_grad_norm=5.0)     # painn-oc.yaml optimizer + clip
    from nabladft_amd import spk
    rep_cls = spk.SchNet if kind == "schnet-spk" else spk.PaiNN
    pot = spk.NeuralNetworkPotential(
        representation=rep_cls(n_atom_basis=F, n_interactions=L, radial_basis=spk.GaussianRBF(n_rbf=R, cutoff=CUTOFF), cutoff_fn=spk.CosineCutoff(CUTOFF)),
        input_modules=[spk.PairwiseDistances()], output_modules=[spk.Atomwise(n_in=F, output_key="energy"), spk.Forces()],
        postprocessors=[spk.AddOffsets("energy", add_mean=True)]).to(dev)
    return pot, nq.FusedTrainStep(pot, lr=1e-4, weight_decay=0.01, max_grad_norm=0.0)            # config/model/painn.yaml:48-52, config/painn.yaml:18-19


def cpu_baseline_spk(kind="painn-spk", seconds_budget=25.0):
    """Same as cpu_baseline for the schnetpack-style models: oracle/spk_painn_ref.py / spk_schnet_ref.py (PARITY UNPINNED restatements)."""
    from oracle import painn_ref as Rf
    import nabladft_amd as nq
    if kind == "schnet-spk":
        from oracle import spk_schnet_ref as S
        scfg = S.SchNetConfig()
        P = S.make_schnet_params(scfg, seed=23)
        S.spk_train_step, S.spk_param_shapes = S.schnet_train_step, S.schnet_param_shapes
    else:
        from oracle import spk_painn_ref as S
        scfg = S.SpkPaiNNConfig()
        P = S.make_spk_params(scfg, seed=23)
    pos, z, batch, y, ft = Rf.gen_conformers(12345, 32)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    e_ref, f_ref, loss_ref, g_ref = S.spk_train_step(P, scfg, pos, z, batch, y, ft)
    warm = time.perf_counter() - t0
    times = []
    while sum(times) + warm < seconds_budget and len(times) < 5:
        t0 = time.perf_counter()
        S.spk_train_step(P, scfg, pos, z, batch, y, ft)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times)) if times else warm
    out = {"value": 32.0 / med, "unit": "conformer-steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
           "sample": f"B=32 synthetic conformers ({pos.shape[0]} atoms), {kind} restatement (parity unpinned), median of {max(len(times), 1)} "
                     f"steps incl. O(n^2) neighbour list, torch {torch.__version__} CPU fp32 without optimizer step"}
    dev = torch.device("cuda", torch.cuda.current_device())
    pot, _ = build_step(kind, dev)
    pot.load_state_dict(P, strict=False)
    fs = nq.FusedTrainStep(pot, max_grad_norm=0.0)
    loss = float(fs(nq.Batch(pos, z, batch, y, ft).to(dev), update=False))
    e, f = fs.energy.cpu(), fs.forces.cpu()
    g_spk = fs.grad.cpu() if pot._index is None else torch.zeros(pot._n_spk, dtype=torch.float32).index_add_(0, pot._index.cpu(), fs.grad.cpu())
    gref = torch.cat([g_ref[k].reshape(-1) for k, _ in S.spk_param_shapes(scfg)])
    parity = {"mae_energy": float((e - e_ref).abs().mean()), "mae_forces": float((f - f_ref).abs().mean()),
              "max_rel_energy": float((e - e_ref).abs().max() / e_ref.abs().max()),
              "max_rel_forces": float((f - f_ref).abs().max() / f_ref.abs().max()),
              "rel_loss": abs(loss - float(loss_ref)) / abs(float(loss_ref)),
              "max_rel_grad": float((g_spk - gref).abs().max() / gref.abs().max()),
              "mean_abs_energy_ref": float(e_ref.abs().mean()), "mean_abs_forces_ref": float(f_ref.abs().mean())}
    return out, parity


def _lib_cap():
    from nabladft_amd import _lib
    return _lib.load().nq_painn_molecule_lds_atoms()


def cpu_baseline(seconds_budget=25.0, all_cores=False, gemm_variant=1):
    """The oracle (pure-torch CPU restatement of the reference path, autograd forces + double backward)
    timed on this box's host cores on a bounded sample: B=32 conformers of the same generator, full config."""
    from oracle import painn_ref as Rf
    cfg = Rf.PaiNNConfig()
    params = Rf.make_params(cfg, seed=23)
    pos, z, batch, y, ft = Rf.gen_conformers(12345, 32)
    # torch-CPU on tiny graphs stops scaling (and collapses with oversubscription) well below the socket size:
    # use 16 threads (the reference's dataloader default is 8 workers; the survey container had 8 cores)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    ei, _, _ = Rf.build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    # the same work as the GPU step: forward, forces, loss, double backward, gradient clipping and AdamW (torch's own optimiser on the flat CPU gradient)
    names_o = [k for k, _ in Rf.param_shapes(cfg)]
    flat_p = torch.nn.Parameter(torch.cat([params[k].reshape(-1) for k in names_o]).clone())
    opt = torch.optim.AdamW([flat_p], lr=5e-4, weight_decay=0.01)

    def cpu_step():
        _, _, _, g = Rf.train_step(params, cfg, pos, z, batch, y, ft)
        flat_p.grad = torch.cat([g[k].reshape(-1) for k in names_o])
        torch.nn.utils.clip_grad_norm_([flat_p], 5.0)
        opt.step()

    t0 = time.perf_counter()
    cpu_step()          # warm-up (includes graph build inside)
    warm = time.perf_counter() - t0
    times = []
    while sum(times) + warm < seconds_budget and len(times) < 5:
        t0 = time.perf_counter()
        cpu_step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times)) if times else warm
    out = {"value": 32.0 / med, "unit": "conformer-steps/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
           "sample": f"B=32 synthetic conformers ({pos.shape[0]} atoms, {ei.shape[1]} edges), PaiNN-OC config, "
                     f"median of {max(len(times), 1)} steps, torch {torch.__version__} CPU fp32 incl. gradient clipping + AdamW (the same work as the GPU step)"}
    # second timing line on every host core (north_star: "the GPU box's host cores"); torch-CPU on graphs of this size usually runs SLOWER there than on 16 threads
    ncpu = os.cpu_count() or 1
    if ncpu > cores and not all_cores:
        out["all_cores"] = {"value": None, "cores": ncpu, "sample": "measured by --full only: one step on every core takes minutes (0.12-0.13 conformer-steps/s on 256 threads, "
                                                                    "profiles/r06_bench_full.json) -- torch-CPU collapses under oversubscription on graphs of this size"}
    if ncpu > cores and all_cores:
        torch.set_num_threads(ncpu)
        t0 = time.perf_counter()
        cpu_step()
        t_all = time.perf_counter() - t0
        out["all_cores"] = {"value": 32.0 / t_all, "cores": ncpu, "sample": "one step of the same B=32 sample on os.cpu_count() threads"}
        torch.set_num_threads(cores)
    # accuracy half of the metric: the HIP path vs this CPU reference on the identical inputs and weights.  Two samples:
    #   * 128 conformers (> 4096 atoms, ~7 s of CPU): the paths the TIMED region runs -- full-width message rows (>= 2048 atoms per launch), rows claimed from
    #     counters, the per-molecule rbf_proj gradient (k_gwr_mol) and the fused update block;
    #   * the B = 32 sample of the timing above (two-slice rows, pair-row gradient: the small-batch paths), kept as a second entry.
    import nabladft_amd as nq
    from nabladft_amd import _lib as _lib_mod
    dev = torch.device("cuda", torch.cuda.current_device())
    m = nq.PaiNN(F, L, R, CUTOFF, KNBR, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100)
    m.load_state_dict(params, strict=False)
    m.to(dev)
    fs = nq.FusedTrainStep(m, max_grad_norm=0.0)
    names = [k for k, _ in Rf.param_shapes(cfg)]

    def parity_of(sample, ei_=None):
        pos_, z_, batch_, y_, ft_ = sample
        e_ref, f_ref, loss_ref, g_ref = Rf.train_step(params, cfg, pos_, z_, batch_, y_, ft_, ei_)
        loss = float(fs(nq.Batch(pos_, z_, batch_, y_, ft_).to(dev), update=False))
        e, f = fs.energy.cpu(), fs.forces.cpu()
        gref = torch.cat([g_ref[k].reshape(-1) for k in names])
        return {"conformers": int(batch_.max()) + 1, "atoms": int(pos_.shape[0]),
                "mae_energy": float((e - e_ref).abs().mean()), "mae_forces": float((f - f_ref).abs().mean()),
                "max_rel_energy": float((e - e_ref).abs().max() / e_ref.abs().max()),
                "max_rel_forces": float((f - f_ref).abs().max() / f_ref.abs().max()),
                "rel_loss": abs(loss - float(loss_ref)) / abs(float(loss_ref)),
                "max_rel_grad": float((fs.grad.cpu() - gref).abs().max() / gref.abs().max()),
                "mean_abs_energy_ref": float(e_ref.abs().mean()), "mean_abs_forces_ref": float(f_ref.abs().mean())}
    big = Rf.gen_conformers(4321, 128)
    # the timed region's products are all >= 192 tiles (split-bf16 engine); at 128 conformers most are not and would fall to the exact-f32 engine:
    # bit 6 of the variant word sends every eligible product of this sample to the engine the timed region uses
    split = gemm_engine_name() == "split-bf16"
    if split:
        _lib_mod.load().nq_set_gemm_variant(gemm_variant | 64)
    try:
        parity = parity_of(big)
    finally:
        if split:
            _lib_mod.load().nq_set_gemm_variant(gemm_variant)
    lds_cap = int(_lib_cap())
    n_max = int(torch.bincount(big[2]).max())
    parity["accuracy_path"] = ("molgw" if n_max <= lds_cap else "molgw+pair_rows(mixed)") + "+full_rows" + ("+fused_update" if F == 128 and os.environ.get("NQ_NO_FUSED_UPDATE") != "1" else "") + ("+split_bf16_products" if split else "+exact_f32_products")
    assert big[0].shape[0] >= 4096, "the accuracy sample must be large enough for the default large-batch paths"
    parity["small_batch"] = dict(parity_of((pos, z, batch, y, ft), ei), accuracy_path="pair_rows+two_slice_rows" + ("+fused_update" if F == 128 and os.environ.get("NQ_NO_FUSED_UPDATE") != "1" else ""))
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="conformers per GPU per step (weak scaling)")
    ap.add_argument("--sustain", type=float, default=8.0, help="seconds of extra (untimed) steps after the timed region: keeps the GPU visibly busy for coarse samplers; 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--full", action="store_true", help="also run the side legs (other engine, sibling config, host feed, inference, QHNet / GemNet-OC / eSCN / "
                                                        "EquiformerV2 / PhiSNet records) into gpurun_out/bench_full.json; the stdout line stays the compact record")
    ap.add_argument("--model", choices=sorted(WORKLOADS), default="painn-oc",
                    help="painn-oc: in-tree PaiNN (parity pinned by reference golden vectors); painn-spk: config/painn.yaml (schnetpack PaiNN, unpinned)")
    ap.add_argument("--gemm-variant", type=int, default=None, help="tuning: nq_set_gemm_variant (bit0 8 waves, bit1 prefetch)")
    args = ap.parse_args()

    import nabladft_amd as nq
    from nabladft_amd import _lib, dist as nqdist
    rank, world, local = nqdist.init_from_env()
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    check_ranks(args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    local_dev = local % torch.cuda.device_count()   # (>1 rank per device only in the 1-GPU plumbing test, NQ_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    if args.gemm_variant is not None:
        _lib.load().nq_set_gemm_variant(args.gemm_variant)
    if args.model == "qhnet":
        return bench_qhnet(args, rank, world, local_dev, dev)
    if args.model == "gemnet":
        return bench_gemnet(args, rank, world, local_dev, dev)
    if args.model in ("escn", "equiformer"):
        return bench_escn(args, rank, world, local_dev, dev, which=args.model)
    torch.manual_seed(23)                                       # config/painn-oc.yaml:38 seed
    model, step = build_step(args.model, dev)
    batches = make_batches(1 + rank, 4, args.batch, dev)
    n_atoms = sum(b.num_nodes for b in batches) / len(batches)

    def sync():
        torch.cuda.synchronize()
        if dist_on():
            dist.barrier(device_ids=[local_dev]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(batches[i % len(batches)])
    sync()
    snap = (step._eng.flat().clone(), step.m.clone(), step.v.clone(), step.t) if args.full and world == 1 else None   # for the engine comparison leg
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[i % len(batches)])
    sync()
    dt = time.perf_counter() - t0
    collective = collective_record(step)   # incl. the exposed all-reduce time of the last timed step
    final_loss = float(loss)          # NOW: `loss` is the step's device-side loss buffer, which every later step (roofline pass, side legs) overwrites
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    total_conf = args.batch * world * args.steps
    value = total_conf / dt
    n_edges = model._last_nl.E

    sustained = None
    if world == 1 and not args.no_roofline and args.sustain > 0:
        # the same steps for a few seconds of wall time (untimed by the contract, reported next to `value`): long enough for a coarse GPU-busy sampler to see the
        # run, and a second throughput figure over ~100 steps
        n_s, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.sustain:
            for i in range(10):
                step(batches[(n_s + i) % len(batches)])
            n_s += 10
            torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        sustained = {"value": args.batch * n_s / dts, "unit": "conformer-steps/s", "steps": n_s, "seconds": dts,
                     "what": "the timed region's step repeated for --sustain seconds (synchronised every 10 steps)"}

    roofline, kernels = None, None
    if not args.no_roofline:
        # second, instrumented pass over the same steps: HIP events around every launch, on the launch stream.  EVERY rank runs the steps
        # (a step contains the gradient all-reduce: a rank-0-only pass would deadlock the other ranks); only rank 0 records events.
        if rank == 0:
            _lib.profile_enable(True)
        for i in range(args.steps):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        tot = sum(v[0] for v in prof.values())
        kernels = sorted(((k, v[0] / args.steps, v[1] // args.steps) for k, v in prof.items()), key=lambda x: -x[1])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "kernel_events.txt"), "w") as fh:
            fh.write(f"# HIP-event timing per launcher class, batch={args.batch}, steps={args.steps}: name ms/step launches/step\n")
            for k, ms, n in kernels:
                fh.write(f"{k:40s} {ms:10.4f} {n:5d}\n")
            fh.write(f"{'TOTAL':40s} {tot / args.steps:10.4f}\n")
        dom, dom_ms_step, dom_launches = kernels[0]
        avg_ms = dom_ms_step / max(dom_launches, 1)
        E = n_edges
        roofline = roofline_record(dom, avg_ms, dom_launches, n_atoms, E, args.batch)
        roofline["device_ms_per_step_all_kernels"] = tot / args.steps
        roofline["step"] = step_bounds(kernels, n_atoms, E, args.batch, 1e3 * dt / args.steps, prof, args.steps)
        roofline["kernel_table"] = kernel_table(kernels, prof, args.steps, n_atoms, E)      # full record only (compact_record keeps the dominant kernel)

    gemm_engine = {"engine": gemm_engine_name(),
                   "what": "dense products of >= 192 tiles of 128x128: every f32 operand value is split exactly into three bf16 pieces (x = h + m + l) and the product "
                           "is the sum of six piece products on v_mfma_f32_32x32x16_bf16 with f32 accumulation (csrc/gemm_split.h; error vs float64 measured <= the "
                           "exact-f32 MFMA engine's on every shape, profiles/r03_gemm_lab_split_bf16_engine.txt); smaller products and NQ_GEMM_F32=1: "
                           "v_mfma_f32_32x32x2_f32 (csrc/gemm_tile.h)"}
    if rank == 0 and world == 1 and not args.no_roofline and args.full:
        try:    # both engines against float64 on one forward product of this step's shape class, in every record (never worth losing the record over)
            gemm_engine["max_error_vs_float64_rel_to_largest_entry"] = gemm_accuracy_probe(dev, args.gemm_variant if args.gemm_variant is not None else 1)
        except Exception as e:
            gemm_engine["max_error_vs_float64_rel_to_largest_entry"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_roofline and args.full and gemm_engine["engine"] == "split-bf16":
        # the same steps from the same parameters / optimiser state / batches with every product on the exact-f32 matrix instruction
        var = args.gemm_variant if args.gemm_variant is not None else 1
        end_state = (step._eng.flat().clone(), step.m.clone(), step.v.clone(), step.t)
        _lib.load().nq_set_gemm_variant(var | 32)
        for i in range(2):
            step(batches[i % len(batches)])
        with torch.no_grad():
            step._eng.flat().copy_(snap[0]); step.m.copy_(snap[1]); step.v.copy_(snap[2]); step.t = snap[3]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_x = step(batches[i % len(batches)])
        torch.cuda.synchronize()
        dtx = time.perf_counter() - t0
        _lib.load().nq_set_gemm_variant(var)
        gemm_engine["exact_f32_engine"] = {"value": args.batch * args.steps / dtx, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dtx / args.steps,
                                           "final_loss": float(loss_x), "final_loss_default_engine": final_loss,
                                           "what": "same initial parameters, optimiser state and batches as the timed region"}
        with torch.no_grad():
            step._eng.flat().copy_(end_state[0]); step.m.copy_(end_state[1]); step.v.copy_(end_state[2]); step.t = end_state[3]

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(all_cores=args.full, gemm_variant=args.gemm_variant if args.gemm_variant is not None else 1) if args.model == "painn-oc" else cpu_baseline_spk(args.model)

    host_feed = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full:
        # PCIe-inclusive rate: the same steps fed from a pinned host arena through the overlapped loader (nabladft_amd/data.py);
        # reported next to `value`, never as `value` (inputs of the timed region above are HBM-resident)
        from nabladft_amd import data as nqdata
        hb = [b.to("cpu") for b in batches]
        arena = nqdata.ConformerArena(torch.cat([b.pos for b in hb]), torch.cat([b.z for b in hb]), torch.cat([b.y for b in hb]),
                                      torch.cat([b.forces for b in hb]),
                                      torch.cat([torch.zeros(1, dtype=torch.long)] + [b.ptr[1:] + sum(int(x.ptr[-1]) for x in hb[:i]) for i, b in enumerate(hb)]))
        loader = nqdata.ArenaLoader(arena, args.batch, dev, shuffle=True, seed=1)
        done = 0
        for bt in loader:                      # warm-up epoch (pins the arena, sizes the staging buffers)
            step(bt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while done < args.steps:
            for bt in loader:
                step(bt)
                done += 1
                if done >= args.steps:
                    break
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0
        host_feed = {"value": args.batch * done / dth, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dth / done,
                     "what": "same step, batches collated from a pinned host arena and copied on a side stream (PCIe-inclusive)"}

    inference = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full:
        # serving / geometry-optimisation mode (optimization/calculator.py:124-130 -> model(batch)): energies + forces only, no second-order sweep
        try:
            model.eval()
            with torch.no_grad():
                for i in range(2):
                    model(batches[i % len(batches)])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    model(batches[i % len(batches)])
                torch.cuda.synchronize()
            dti = time.perf_counter() - t0
            inference = {"value": args.batch * args.steps / dti, "unit": "conformers/s", "ms_per_call": 1e3 * dti / args.steps,
                         "what": "model(batch) -> (energy, forces) incl. neighbour list, same batches, eval mode"}
        except (TypeError, KeyError, AttributeError) as exc:      # the schnetpack-style potentials take dict inputs: not measured here
            inference = {"value": None, "what": f"not measured for this model interface ({type(exc).__name__})"}
        finally:
            model.train()

    other, kind2 = None, None
    if rank == 0 and world == 1 and not args.no_roofline:
        # the sibling PaiNN configuration through the same kernels (reported, not `value`): the timing leg always, its kernel table / CPU leg with --full
        kind2 = "painn-spk" if args.model == "painn-oc" else "painn-oc"
        if args.model == "schnet-spk":
            kind2 = None
    small = None
    if rank == 0 and world == 1 and not args.no_roofline and args.batch > 32:
        # the reference's default per-GPU batch (config/painn-oc.yaml:11: 32 conformers), same model and step, next to the throughput batch
        sb = make_batches(77, 4, 32, dev)
        for i in range(5):
            step(sb[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            step(sb[i % 4])
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        small = {"conformers_per_step": 32, "value": 32 * 40 / dts, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dts / 40,
                 "what": "same training step at the reference's default batch size (latency-bound: ~330 small kernels back to back)"}
    if rank == 0 and world == 1 and not args.no_roofline and kind2 is not None:
        del step, model
        torch.cuda.empty_cache()
        model2, step2 = build_step(kind2, dev)
        for i in range(max(args.warmup, 1)):
            step2(batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step2(batches[i % len(batches)])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        other = {"workload": WORKLOADS[kind2], "value": args.batch * args.steps / dt2, "unit": "conformer-steps/s", "ms_per_step": 1e3 * dt2 / args.steps,
                 "parity": "pinned (reference golden vectors)" if kind2 == "painn-oc" else "unpinned (schnetpack is not in the reference tree; restatement oracle/spk_painn_ref.py)"}
        # its own roofline record (same instrumented pass as the headline workload) and accuracy against its CPU restatement
        k2 = []
        if args.full:
            _lib.profile_enable(True)
            for i in range(args.steps):
                step2(batches[i % len(batches)])
            torch.cuda.synchronize()
            prof2 = _lib.profile_read()
            _lib.profile_enable(False)
            k2 = sorted(((k, v[0] / args.steps, v[1] // args.steps) for k, v in prof2.items()), key=lambda x: -x[1])
        if k2:
            d2, ms2, n2 = k2[0]
            other["roofline"] = roofline_record(d2, ms2 / max(n2, 1), n2, n_atoms, model2._last_nl.E if hasattr(model2, "_last_nl") else n_edges, args.batch)
            other["kernel_ms_per_step"] = {k: round(ms, 4) for k, ms, _ in k2[:6]}
        if args.full and not args.no_cpu_baseline and kind2 != "painn-oc":
            cpu2, par2 = cpu_baseline_spk(kind2, seconds_budget=8.0)
            other["cpu_baseline"], other["mae_vs_cpu_reference"] = cpu2, par2

    hamiltonian = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full and args.model == "painn-oc":
        # BASELINE.json configs[3] (QHNet, config/qhnet.yaml) in the same record: a short run at the reference's batch size (2) and at 16
        try:
            del step2, model2
        except NameError:
            pass
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import bench_qhnet as BQ
        h16 = BQ.run(16, 5, 2, kernels=True, device=dev)
        h2 = BQ.run(2, 5, 2, kernels=False, device=dev)
        for h in (h16, h2):
            h.pop("_dt", None)
        hamiltonian = {"workload": h16.pop("workload"), "batch16": h16, "batch2_reference_batch_size": {k: h2[k] for k in ("value", "unit", "ms_per_step", "atoms", "ordered_pairs")},
                       "cpu_baseline": None if args.no_cpu_baseline else BQ.cpu_baseline(seconds_budget=20.0)}

    gemnet = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full and args.model == "painn-oc":
        # BASELINE.json configs[2] (GemNet-OC, config/model/gemnet-oc.yaml, fp32) in the same record
        torch.cuda.empty_cache()
        import bench_gemnet as BG
        g16 = BG.run(16, 5, 2, kernels=True, device=dev)
        g16b = BG.run(16, 5, 4, kernels=True, device=dev, precision="bf16")
        g16.pop("_dt", None)
        gemnet = {"workload": g16.pop("workload"), "batch16": g16,
                  "batch16_bf16_gemms": {k: g16b.get(k) for k in ("value", "unit", "ms_per_step", "dtype", "final_loss", "roofline", "gemm_bf16_ms_per_step")},
                  "cpu_baseline": None if args.no_cpu_baseline else BG.cpu_baseline(seconds_budget=15.0)}

    phisnet = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full and args.model == "painn-oc":
        # PhiSNet (rows a21-a24; north_star "QHNet/PhiSNet SO(3) tensor-product convolutions") at the nablaDFT configuration, 2 molecules per step, in the same record
        torch.cuda.empty_cache()
        import bench_phisnet as BP
        try:
            phisnet = {"eager": BP.run(2, 42, 10, 3, kernels=True, forces=True), "graph_replay": BP.run(2, 42, 10, 3, graph=True)}
        except Exception as e:                      # a failed capture must not cost the record its other legs
            phisnet = {"error": repr(e)[:300]}

    escn = equiformer = None
    if rank == 0 and world == 1 and not args.no_roofline and args.full and args.model == "painn-oc":
        # BASELINE.json configs[4] (eSCN, config/model/escn-oc.yaml, fp32) in the same record
        torch.cuda.empty_cache()
        import bench_escn as BE
        e16 = BE.run(16, 3, 2, kernels=True, device=dev)
        e16.pop("_dt", None)
        escn = {"workload": e16.pop("workload"), "batch16": e16, "cpu_baseline": None if args.no_cpu_baseline else BE.cpu_baseline(seconds_budget=10.0)}
        # ... and its second model, EquiformerV2 (config/model/equiformer_v2_oc20.yaml, fp32, training mode)
        torch.cuda.empty_cache()
        import bench_equiformer as BQ2
        q16 = BQ2.run(16, 3, 2, kernels=True, device=dev)
        q16.pop("_dt", None)
        equiformer = {"workload": q16.pop("workload"), "batch16": q16, "cpu_baseline": None if args.no_cpu_baseline else BQ2.cpu_baseline(seconds_budget=8.0)}

    if rank == 0:
        out = {
            "metric": "conformer-steps/sec (fwd+bwd) + MAE(E,F) vs CPU reference", "value": value, "unit": "conformer-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_string(), "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.model]}; synthetic ~42-atom drug-like conformers, {args.batch} conformers/GPU/step",
                       "conformers_per_gpu": args.batch, "atoms_per_step_per_gpu": n_atoms, "edges_last_step": n_edges,
                       "parallelism": f"dp{world}", "collective": collective},
            "final_loss": final_loss,
            "gemm_engine": gemm_engine,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "mae_vs_cpu_reference": parity,
            "sibling_config": other,
            "hamiltonian": hamiltonian,
            "phisnet": phisnet,
            "gemnet_oc": gemnet,
            "escn": escn,
            "equiformer_v2": equiformer,
            "reference_batch_size_32": small,
            "sustained": sustained,
            "host_feed": host_feed, "inference": inference,
            # end-to-end fraction of the HBM roofline under SURVEY.md 8(d)'s contract figure (17.8 MB / conformer-step)
            "e2e_algorithmic_GBps_per_gpu": 17.8e6 * value / world / 1e9,
            "kernel_ms_per_step": {k: round(ms, 4) for k, ms, _ in (kernels or [])[:8]},
        }
        emit(out)
    finish(local_dev)


if __name__ == "__main__":
    main()
