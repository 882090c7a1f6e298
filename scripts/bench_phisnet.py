"""PhiSNet training-step timing on one MI355X: nabladft_amd.phisnet.NeuralNetwork at the nablaDFT
configuration (phisnet/configs/args_nablaDFT_100k_separate.txt: order 4, F = 128, K = 128, 5 modules, cutoff 15, swish; batch of 2 molecules
as train_batch_size there), forward + MAE loss on the packed full Hamiltonian and overlap + backward.  Synthetic drug-like molecules (H, C, N, O with
def2-SVP-like shells), random-init weights with the zero-initialised layers randomised.  Prints one JSON line.
    python scripts/bench_phisnet.py [--molecules 2] [--atoms 42] [--steps 5] [--warmup 2] [--kernels]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHELLS = {1: (0, 0, 1), 6: (0, 0, 0, 1, 1, 2), 7: (0, 0, 0, 1, 1, 2), 8: (0, 0, 0, 1, 1, 2)}
HP = dict(order=4, num_features=128, num_basis_functions=128, num_modules=5, num_residual_pre_x=1, num_residual_post_x=1, num_residual_pre_vi=1,
          num_residual_pre_vj=1, num_residual_post_v=1, num_residual_output=1, num_residual_pc=1, num_residual_pn=1, num_residual_ii=1, num_residual_ij=1,
          num_residual_full_ii=2, num_residual_full_ij=2, num_residual_core_ii=2, num_residual_core_ij=2, num_residual_over_ij=2,
          basis_functions="exp-bernstein", cutoff=15.0, activation="swish")


def synthetic_batch(molecules, atoms, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    zs, pos, sizes = [], [], []
    for _ in range(molecules):
        z = rng.choice([1, 6, 7, 8], size=atoms, p=[0.5, 0.35, 0.07, 0.08])
        # random points with a minimum separation of ~1 A in a box that keeps the density of an organic molecule
        side = (atoms * 9.0) ** (1 / 3)
        p = []
        while len(p) < atoms:
            c = rng.uniform(0, side, size=3)
            if all(np.linalg.norm(c - q) > 0.95 for q in p):
                p.append(c)
        zs.append(z), pos.append(np.array(p, dtype=np.float32)), sizes.append(atoms)
    z = np.concatenate(zs)
    return dict(z=z, positions=np.concatenate(pos), sizes=np.array(sizes), orbitals=[tuple((int(a), l) for l in SHELLS[int(a)]) for a in z])


def run(molecules=2, atoms=42, steps=20, warmup=3, kernels=False, graph=False, per_tensor_optimizer=False, forces=False):
    """One record (dict) of the PhiSNet training step; ``forces`` adds the inference call with predict_energy + calculate_forces (first-order adjoints)."""
    from types import SimpleNamespace
    return _run(SimpleNamespace(molecules=molecules, atoms=atoms, steps=steps, warmup=warmup, kernels=kernels, graph=graph,
                                per_tensor_optimizer=per_tensor_optimizer, forces=forces))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=2)
    ap.add_argument("--atoms", type=int, default=42)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernels", action="store_true", help="per-kernel HIP-event table (nq profile hooks)")
    ap.add_argument("--graph", action="store_true", help="capture the step into a HIP graph (trainer.GraphedStep) and replay it")
    ap.add_argument("--per-tensor-optimizer", action="store_true", help="torch Adam over the 2.4 k parameter tensors instead of the flat buffer")
    ap.add_argument("--forces", action="store_true", help="also time the energy + forces inference call (predict_energy, calculate_forces, create_graph=False)")
    print(json.dumps(_run(ap.parse_args())))


def _run(a):
    import torch
    from nabladft_amd import _lib
    from nabladft_amd.phisnet import NeuralNetwork
    max_orbitals = tuple(tuple((zz, l) for l in SHELLS[zz]) for zz in (1, 1, 6, 6, 7, 7, 8, 8))
    torch.manual_seed(0)
    m = NeuralNetwork(max_orbitals=max_orbitals, **HP)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.abs().max() == 0:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    m = m.cuda()
    b = synthetic_batch(a.molecules, a.atoms)
    batch = dict(positions=torch.tensor(b["positions"]).view(1, -1, 3).cuda(), atomic_numbers=torch.tensor(b["z"]).cuda(), orbitals=b["orbitals"],
                 molecule_size=torch.tensor(b["sizes"]))
    params = [p for p in m.parameters() if p.requires_grad]
    from nabladft_amd.trainer import FlatParameters
    flat = None if a.per_tensor_optimizer else FlatParameters(params)
    if flat is not None:
        flat.attach(m)
    opt = torch.optim.Adam(params if flat is None else [flat.flat], lr=1e-3, amsgrad=True, capturable=a.graph)
    if a.graph:
        batch["prepared"] = m.prepare(batch)

    def step():
        if flat is None:
            opt.zero_grad(set_to_none=True)
        else:
            flat.zero_grad()
        out = m(batch)
        loss = out["full_hamiltonian_packed"].abs().mean() + out["overlap_matrix_packed"].abs().mean()
        loss.backward()
        if flat is None:
            torch.nn.utils.clip_grad_norm_(params, 1.0)
        else:
            flat.clip_grad_norm_(1.0)
        opt.step()
        return loss
    if a.graph:
        from nabladft_amd.trainer import GraphedStep
        step = GraphedStep(step, warmup=max(a.warmup, 3))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if a.kernels:
        _lib.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    prof = None
    if a.kernels:                       # read the per-kernel table of exactly the timed steps (the wall-clock loop below must not add to it)
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    n_params = sum(p.numel() for p in params)
    P = int(sum(s * (s - 1) for s in b["sizes"]))
    wall0 = __import__("time").perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    wall_ms = 1e3 * (__import__("time").perf_counter() - wall0) / a.steps
    out = {"graph_replay": bool(a.graph), "wall_ms_per_step": wall_ms, "metric": "PhiSNet molecule-steps/sec (fwd + MAE(H,S) + bwd + clip + AMSGrad)", "value": a.molecules / (ms * 1e-3), "unit": "molecule-steps/s",
           "ms_per_step": ms, "molecules": a.molecules, "atoms": int(len(b["z"])), "ordered_pairs": P, "orbitals": int(sum(2 * l + 1 for o in b["orbitals"] for _, l in o)),
           "parameters": n_params, "final_loss": float(loss), "config": {k: v for k, v in HP.items()}, "data": "synthetic", "dtype": "f32"}
    if a.kernels:
        out["device_ms_per_step_nq_kernels"] = sum(v[0] for v in prof.values()) / a.steps
        out["kernel_ms_per_step"] = {k: [round(v[0] / a.steps, 4), int(v[1] // a.steps)] for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:14]}
        dense = {k: v for k, v in prof.items() if v[2] > 0}
        if dense:   # the dense classes record their 2 M N K: TFLOP/s per class and the roofline entry of the largest one
            k, v = max(dense.items(), key=lambda kv: kv[1][0])
            ach = v[2] / (v[0] * 1e-3) / 1e12
            out["roofline"] = {"kernel": k, "bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                               "avg_launch_ms": v[0] / max(v[1], 1), "launches_per_step": int(v[1] // a.steps),
                               "note": "largest dense launcher class of the step by device time; exact-f32 MFMA peak (these launches are below the split engine's 192-tile threshold)"}
    # SURVEY 8(d) "QHNet / PhiSNet work unit" as a contract figure: per molecule and module the node irreps (25 components x F floats per atom) are read and written
    # once and the pair features (per ordered pair, same width) read and written once; the two output matrices are written once; a training step = forward +
    # backward = 3 forward-equivalents (the backward streams every tensor once more as adjoint and once as saved input)
    width = sum(2 * l + 1 for l in range(HP["order"] + 1)) * HP["num_features"] * 4.0
    n_at, n_orb = float(out["atoms"]), float(out["orbitals"])
    fwd_bytes = HP["num_modules"] * (2.0 * n_at * width + 2.0 * P * width) + 2.0 * n_orb * n_orb * 4.0 / max(a.molecules, 1)
    step_bytes = 3.0 * fwd_bytes
    ach_gbs = step_bytes / (ms * 1e-3) / 1e9
    out["roofline_hbm_contract"] = {"bound": "hbm", "achieved": ach_gbs, "peak": 8000.0, "unit": "GB/s", "frac": ach_gbs / 8000.0,
                                    "contract_bytes_per_step": step_bytes,
                                    "formula": "3 x [modules x (2 N w + 2 P w) + 2 Norb^2 4 / molecules], w = 25 F 4 bytes (order 4), N atoms, P ordered pairs of the step"}
    if getattr(a, "forces", False):
        m.predict_energy = m.calculate_forces = True
        m.create_graph = False
        for _ in range(2):
            m(batch)
        torch.cuda.synchronize()
        t0 = __import__("time").perf_counter()
        for _ in range(a.steps):
            m(batch)
        torch.cuda.synchronize()
        out["energy_forces_inference_ms"] = 1e3 * (__import__("time").perf_counter() - t0) / a.steps
        m.predict_energy = m.calculate_forces = False
    return out


if __name__ == "__main__":
    main()
