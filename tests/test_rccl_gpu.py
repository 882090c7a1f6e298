"""RCCL on the one GPU of the test box: a 1-rank process group with backend "nccl" (= RCCL on ROCm) forced through every collective
of a training step (NQ_DIST_FORCE=1), both through torch.distributed and through the C ABI (nq_rccl_* / nq_allreduce, librccl bound with dlopen).  Proves
that librccl loads with HSA_ENABLE_IPC_MODE_LEGACY=0, that the communicator initialises, and that the collectives are stream-ordered between the kernels
that write the gradient and the optimiser kernel that reads it: the result of a step must equal the step of a process without a group.  No scaling number
comes out of this (a second rank needs a second GPU): the driver's 8-GPU run is the first multi-rank execution.  Each case runs in its own process."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_CHILD = r'''
import os, sys, json
sys.path.insert(0, os.environ["NQ_ROOT"])
import torch
import nabladft_amd as nq
from nabladft_amd import dist as nqdist, _lib
from oracle import painn_ref as R

mode = sys.argv[1]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if mode != "single":
    rank, world, local = nqdist.init_from_env()
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and nqdist.active()
    assert (nqdist.native_comm(create=False) is not None) == (mode == "native")

cfg = R.PaiNNConfig(hidden_channels=64, num_layers=3, num_rbf=20, cutoff=5.0)
def model():
    m = nq.PaiNN(cfg.hidden_channels, cfg.num_layers, cfg.num_rbf, cfg.cutoff, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False,
                 False, True, cfg.num_elements)
    m.load_state_dict(R.make_params(cfg, seed=41), strict=False)
    return m.to(dev)
pos, z, batch, y, ft = R.gen_conformers(70, 6, size=(8, 20))
b = nq.Batch(pos, z, batch, y, ft).to(dev)
out = {}
for ov in (True, False):
    step = nq.FusedTrainStep(model(), lr=1e-3, max_grad_norm=5.0)
    step.overlap = ov
    losses = [float(step(b)) for _ in range(3)]
    torch.cuda.synchronize()
    out["overlap" if ov else "plain"] = dict(losses=losses, used_overlap=step._ov is not None, grad=step.grad.cpu().tolist()[:64], exposed_ms=step.allreduce_exposed_ms(),
                                             flat_sum=float(step._eng.flat().double().sum()), grad_norm=float(step.grad.double().norm()))
# autograd-driven models: bucketed all-reduce from post-accumulate hooks on a side stream
from nabladft_amd.trainer import FlatParameters, OverlappedAllReduce
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(7, 50), torch.nn.SiLU(), torch.nn.Linear(50, 19), torch.nn.SiLU(), torch.nn.Linear(19, 1)).to(dev)
flat = FlatParameters(list(net.parameters()))
ovr = OverlappedAllReduce(flat, bucket_bytes=600)
x = torch.randn(32, 7, generator=torch.Generator().manual_seed(3)).to(dev)
flat.zero_grad()
net(x).pow(2).mean().backward()
g = ovr.finish().clone()
out["hooks"] = dict(on=ovr.on, buckets=len(ovr.buckets), grad_norm=float(g.double().norm()), grad=g.cpu().tolist()[:32])
if mode == "native":
    # the C-ABI collectives directly: sum over one rank leaves the buffer unchanged, ordered on the launch stream behind the kernel that fills it
    comm = nqdist.native_comm(create=False)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t = torch.zeros(1 << 22, device=dev)
        t.add_(1.5)                                   # producer kernel on stream s
        comm.allreduce_mean_(t)                       # enqueued on s (current stream)
        t.mul_(2.0)                                   # consumer kernel on s
    s.synchronize()
    out["native_direct"] = dict(min=float(t.min()), max=float(t.max()), world=comm.world)
    out["comm_count"] = comm.ranks_seen()
    lib = _lib.load()
    out["available"] = int(lib.nq_rccl_available())
out["ranks_seen"] = nqdist.ranks_seen()
if mode != "single":
    nqdist.barrier(device=0)
    nqdist.drop_native_comm()
    dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _run(mode):
    env = dict(os.environ, NQ_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NQ_DIST_FORCE", "NQ_RCCL_NATIVE", "NQ_DIST_BACKEND"):
        env.pop(k, None)
    if mode != "single":
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", NQ_DIST_FORCE="1")
    if mode == "native":
        env["NQ_RCCL_NATIVE"] = "1"
    r = subprocess.run([sys.executable, "-c", _CHILD, mode], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def single():
    return _run("single")


@pytest.mark.parametrize("mode", ["torch", "native"])
def test_one_rank_rccl_step_equals_the_step_without_a_group(mode, single):
    got = _run(mode)
    for key in ("overlap", "plain"):
        assert got[key]["losses"] == single[key]["losses"], (mode, key)         # bitwise: a sum over one rank is the identity, and the step is deterministic
        assert got[key]["grad"] == single[key]["grad"] and got[key]["flat_sum"] == single[key]["flat_sum"]
        assert got[key]["grad_norm"] > 0
    assert got["overlap"]["used_overlap"] and not got["plain"]["used_overlap"]   # the per-layer overlapped all-reduce path really ran under the forced group
    assert not single["overlap"]["used_overlap"]
    assert got["hooks"]["on"] and got["hooks"]["buckets"] >= 3 and got["hooks"]["grad"] == single["hooks"]["grad"]
    assert got["ranks_seen"] == 1 and single["ranks_seen"] == 1
    for key in ("overlap", "plain"):                      # the exposed all-reduce time is measured on the step's stream whenever a group is active
        assert got[key]["exposed_ms"] is not None and 0.0 <= got[key]["exposed_ms"] < 50.0, (mode, key, got[key]["exposed_ms"])
        assert single[key]["exposed_ms"] is None
    if mode == "native":
        assert got["available"] == 1 and got["comm_count"] == 1     # ncclCommCount through the C ABI (nq_rccl_comm_count)
        assert got["native_direct"] == {"min": 3.0, "max": 3.0, "world": 1}


def test_bench_one_rank_through_the_nccl_branch():
    """bench.py exactly as the driver launches a rank (RANK / WORLD_SIZE / MASTER_* in the environment), with the 1-rank group forced: process-group
    creation with backend nccl, barriers pinned to the device, max-over-ranks all-reduce of the time, gradient all-reduce inside every step."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               NQ_DIST_FORCE="1")
    env.pop("NQ_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--sustain", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    rec = json.loads(last)
    col = rec["config"]["collective"]
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and col["name"].startswith("rccl via torch.distributed")
    assert col["backend"] == "rccl" and col["ranks_seen"] == 1 and col["path"] == "torch" and col["allreduce_exposed_ms"] is not None
    assert rec["roofline"]["frac"] > 0
