"""Device-side cost of a training step at the REFERENCE's batch sizes (config/*.yaml: QHNet 2, GemNet-OC 8, eSCN 8, EquiformerV2 2 conformers per step).
At these sizes the eager step of the autograd-driven models is a chain of 900-1500 small launches issued from Python: the host, not the GPU, sets the step
time.  Here the whole step (zero_grad, forward on a PREPARED batch -- ``net.prepare(batch)``: graphs and index lists built ahead, no host read left in the
forward --, loss, backward, clip, AdamW) is captured ONCE into a HIP graph (trainer.GraphedStep) and replayed; the same fixed batch is also stepped eagerly.
What the replay number is: the step's cost once the launches no longer come from Python.  What it is not: a training loop over changing batches (the edge
sets depend on the geometry, so a captured graph is tied to its batch).

    python scripts/bench_graphed.py --model qhnet|gemnet|escn|equiformer [--molecules N] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
REFERENCE_BATCH = {"qhnet": 2, "gemnet": 8, "escn": 8, "equiformer": 2}


def make_step(which, molecules, dev, seed=1):
    """-> (step(), flat, opt, net): one training step on ONE prepared batch, free of host synchronisation."""
    import torch
    from nabladft_amd.trainer import FlatParameters
    if which == "qhnet":
        import bench_qhnet as M
        from nabladft_amd.hamiltonian import HamiltonianLoss
        net = M.build(dev)
        b = M.synthetic_batch(molecules, seed * 100, dev)
        b.prepared = net.prepare(b)
        loss_fn = HamiltonianLoss()
        with torch.no_grad():
            h = net(b, packed=True)
            target = (h + 0.05 * torch.randn(h.shape, generator=torch.Generator().manual_seed(7)).to(dev)).detach()
        flat = FlatParameters(net.parameters())
        opt = torch.optim.AdamW([flat.flat], lr=5e-4, betas=(0.9, 0.95), amsgrad=True, capturable=True)

        def step():
            flat.zero_grad()
            loss = loss_fn(net(b, packed=True), target)
            loss.backward()
            opt.step()
            return loss
        return step, flat, opt, net
    M = {"gemnet": "bench_gemnet", "escn": "bench_escn", "equiformer": "bench_equiformer"}[which]
    M = __import__(M)
    net = M.build(dev)
    if which == "equiformer":
        net.eval()                                     # drop-path / attention dropout draw host-side random numbers per call
    b = M.synthetic_batch(molecules, seed * 100, dev)
    b.prepared = net.prepare(b)
    flat = FlatParameters(net.parameters())
    opt = torch.optim.AdamW([flat.flat], lr=1e-3, betas=(0.9, 0.95), amsgrad=True, weight_decay=0, capturable=True)
    clip = 10.0 if which == "gemnet" else None

    def step():
        flat.zero_grad()
        E, F = net(b)[:2]
        loss = M.loss_fn(E, F, b)
        loss.backward()
        if clip:
            flat.clip_grad_norm_(clip)
        opt.step()
        return loss
    return step, flat, opt, net


def run(which, molecules=None, steps=20, warmup=3, device=None):
    import torch
    from nabladft_amd.trainer import GraphedStep
    dev = device or torch.device("cuda", torch.cuda.current_device())
    molecules = molecules or REFERENCE_BATCH[which]
    step, flat, opt, net = make_step(which, molecules, dev)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()                                         # no reference to the loss is kept: a live autograd graph of an eager step (its AccumulateGrad nodes
    torch.cuda.synchronize()                           # belong to the default stream) would be touched from the capture stream and break the capture
    eager_ms = 1e3 * (time.perf_counter() - t0) / steps
    print(f"# {which}: eager {eager_ms:.3f} ms/step at {molecules} conformers", file=sys.stderr, flush=True)
    g = GraphedStep(step, warmup=2)
    for _ in range(2):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = g()
    torch.cuda.synchronize()
    graph_ms = 1e3 * (time.perf_counter() - t0) / steps
    return {"model": which, "molecules_per_step": molecules, "eager_ms_per_step": eager_ms, "graph_replay_ms_per_step": graph_ms,
            "eager_conformer_steps_per_s": molecules / eager_ms * 1e3, "graph_replay_conformer_steps_per_s": molecules / graph_ms * 1e3,
            "final_loss": float(loss.detach()), "what": "one prepared batch stepped repeatedly: eager = launches issued from Python autograd, graph replay = the same step "
            "captured once into a HIP graph (device-side cost of the step; tied to this batch's edge sets)"}


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=sorted(REFERENCE_BATCH), required=True)
    ap.add_argument("--molecules", type=int, default=None)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    print(json.dumps(run(a.model, a.molecules, a.steps)))
