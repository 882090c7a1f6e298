"""CPU, world_size 2 over gloo: the N>1 path (flat-gradient mean all-reduce, parameter broadcast, conformer
sharding).  Each rank computes the ORACLE's gradient on its shard; the all-reduced mean must equal the mean of the
two single-process results -- the data-parallel semantics of the reference (Lightning DDP, utils/pipelines.py:65-68)."""
import os
import sys

import pytest
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nabladft_amd import dist as nqdist
from oracle import painn_ref as R


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_grad(cfg, params, shard_ids, pos, z, batch, y, ft):
    sel = torch.isin(batch, torch.tensor(shard_ids))
    remap = {m: i for i, m in enumerate(shard_ids)}
    b = torch.tensor([remap[int(m)] for m in batch[sel]])
    _, _, loss, g = R.train_step(params, cfg, pos[sel], z[sel], b, y[shard_ids], ft[sel])
    return loss, torch.cat([g[k].reshape(-1) for k, _ in R.param_shapes(cfg)])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = nqdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = R.PaiNNConfig(hidden_channels=64, num_layers=2, num_rbf=12, cutoff=4.0)
    pos, z, batch, y, ft = R.gen_conformers(5, 6, size=(6, 16))
    sizes = torch.bincount(batch).tolist()
    shards = nqdist.shard_by_cost(sizes, world)
    # rank 0's parameters win (broadcast), like DDP at wrap time
    params = R.make_params(cfg, seed=100 + rank)
    flat = torch.cat([params[k].reshape(-1) for k, _ in R.param_shapes(cfg)])
    nqdist.broadcast_(flat, 0)
    o = 0
    for k, shp in R.param_shapes(cfg):
        n = int(np.prod(shp))
        params[k] = flat[o:o + n].view(shp).clone()
        o += n
    _, g = _shard_grad(cfg, params, shards[rank], pos, z, batch, y, ft)
    nqdist.allreduce_mean_(g)
    torch.save({"g": g, "flat": flat, "shards": shards}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["flat"], r1["flat"])
    cfg = R.PaiNNConfig(hidden_channels=64, num_layers=2, num_rbf=12, cutoff=4.0)
    params = R.make_params(cfg, seed=100)
    assert torch.equal(r0["flat"], torch.cat([params[k].reshape(-1) for k, _ in R.param_shapes(cfg)]))
    pos, z, batch, y, ft = R.gen_conformers(5, 6, size=(6, 16))
    gs = [_shard_grad(cfg, params, s, pos, z, batch, y, ft)[1] for s in r0["shards"]]
    expect = (gs[0] + gs[1]) / 2
    assert float((r0["g"] - expect).abs().max()) <= 1e-6 * float(expect.abs().max())


def test_shard_by_cost_balanced_and_complete():
    rng = np.random.Generator(np.random.PCG64(0))
    sizes = rng.integers(10, 91, size=257).tolist()
    for world in (1, 2, 4, 8):
        shards = nqdist.shard_by_cost(sizes, world)
        assert sorted(i for s in shards for i in s) == list(range(len(sizes)))
        load = [sum(sizes[i] ** 2 for i in s) for s in shards]
        assert max(load) <= 1.05 * (sum(load) / world) + 90 * 90
        assert all(s == sorted(s) for s in shards)


def test_cost_proxies_balance_the_10_to_90_atom_mix_for_every_model():
    """BASELINE.json configs[4]: "mixed molecule sizes 10-90 atoms (load-balance stress)".  For every model's cost proxy the greedy partition of a global
    batch over 8 ranks stays within 5 % of the mean from 16 conformers per rank on (pure arithmetic, no GPU needed), and the proxies differ the way the
    graphs do: a neighbour-capped model weighs a 90-atom molecule ~9x a 10-atom one, the pair models ~90x."""
    rng = np.random.Generator(np.random.PCG64(3))
    for model in ("painn", "escn", "equiformer_v2", "gemnet_oc", "qhnet", "phisnet"):
        for per_rank in (16, 32, 64):
            sizes = rng.integers(10, 91, size=8 * per_rank).tolist()
            shards = nqdist.shard_by_cost(sizes, 8, model)
            assert sorted(i for s in shards for i in s) == list(range(len(sizes)))
            assert nqdist.predicted_spread(sizes, 8, model) <= 0.05, (model, per_rank, nqdist.predicted_spread(sizes, 8, model))
    assert 8.0 < nqdist.conformer_cost(90, "escn") / nqdist.conformer_cost(10, "escn") < 35.0
    assert 80.0 < nqdist.conformer_cost(90, "qhnet") / nqdist.conformer_cost(10, "qhnet") < 100.0
    # the n^2 proxy mis-balances a capped model: partition by n^2, evaluate with the capped cost
    sizes = rng.integers(10, 91, size=128).tolist()
    parts = nqdist.shard_by_cost(sizes, 8, "n2")
    loads = [sum(nqdist.conformer_cost(sizes[i], "equiformer_v2") for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) - 1.0 > nqdist.predicted_spread(sizes, 8, "equiformer_v2")


def test_bench_never_calls_a_training_step_on_one_rank_only():
    """bench.py: a training step contains the gradient all-reduce, so inside main() every call of step()/step2() must be reached by ALL
    ranks or be guarded by ``world == 1`` (a rank-0-only instrumented pass once deadlocked every multi-GPU run)."""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "main")
    parents = {}
    for node in ast.walk(main):
        for child in ast.iter_child_nodes(node):
            parents[child] = node
    calls = [n for n in ast.walk(main) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in ("step", "step2")]
    assert len(calls) >= 4
    for call in calls:
        conds, node = [], call
        while node in parents:
            parent = parents[node]
            if isinstance(parent, ast.If) and node in parent.body:
                conds.append(ast.unparse(parent.test))
            node = parent
        gated = [c for c in conds if "rank == 0" in c]
        assert all("world == 1" in c for c in gated), (call.lineno, conds)


# ---- bucketed all-reduce overlapped with backward (trainer.OverlappedAllReduce), world 2 and 3 over gloo ------------------------------------------------
class _OddNet(torch.nn.Module):
    """Parameter sizes chosen against FlatParameters' 16-byte alignment: an odd-sized tensor first (50 floats, as QHNet's radial parameters), unused
    parameters in the MIDDLE (their bucket must still be reduced: zeros), and the LAST parameters receive non-zero gradients -- with ranges taken from
    a running sum of numel() (the round-3 bug) the tail of the buffer was scaled by 1/world without being reduced and buckets cut through parameters."""

    def __init__(self):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.randn(50) * 0.1 + 1.0)
        self.l1 = torch.nn.Linear(7, 50)
        self.unused = torch.nn.Linear(3, 3)
        self.l2 = torch.nn.Linear(50, 19)
        self.odd = torch.nn.Parameter(torch.randn(17))
        self.l3 = torch.nn.Linear(19, 1)
        self.tail = torch.nn.Parameter(torch.randn(21))

    def forward(self, x):
        h = torch.nn.functional.silu(self.l1(x) * self.scale)
        h = torch.nn.functional.silu(self.l2(h))
        return self.l3(h) * self.odd[:16].sum() + (self.tail * self.tail).sum()


def _mlp(seed):
    torch.manual_seed(seed)
    return _OddNet()


def test_overlapped_buckets_tile_the_padded_flat_buffer():
    from nabladft_amd.trainer import FlatParameters, OverlappedAllReduce
    sizes = [(50,), (8, 16), (3,), (5,), (32, 7), (1,), (128,), (17, 4)]             # summed numel 607, padded buffer 612 floats
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in sizes]
    flat = FlatParameters(ps)
    assert flat.flat.numel() > sum(p.numel() for p in ps)                               # padding is present
    for bb in (1, 64, 600, 1 << 20):
        ov = OverlappedAllReduce(flat, bucket_bytes=bb)
        ov.check_tiling()
        assert ov.buckets[0][0] == 0 and ov.buckets[-1][1] == flat.flat.numel()
        for p in ps:
            lo, hi, _ = ov.buckets[ov._bucket_of[id(p)]]
            assert lo <= flat.offset[id(p)] and flat.offset[id(p)] + p.numel() <= hi
    net = _mlp(0)
    flat = FlatParameters(list(net.parameters()))
    assert flat.flat.numel() > sum(p.numel() for p in net.parameters())
    ov = OverlappedAllReduce(flat, bucket_bytes=600)
    ov.check_tiling()
    assert len(ov.buckets) >= 3


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    nqdist.init_from_env(backend="gloo")
    from nabladft_amd.trainer import FlatParameters, OverlappedAllReduce
    net = _mlp(0)
    flat = FlatParameters(list(net.parameters()))
    ov = OverlappedAllReduce(flat, bucket_bytes=600)        # several buckets over ~1.5 k parameters, padded layout
    assert len(ov.buckets) >= 3 and flat.flat.numel() > sum(p.numel() for p in net.parameters())
    net2 = _mlp(0)                                          # the same model without hooks: local gradient -> one flat all-reduce
    flat2 = FlatParameters(list(net2.parameters()))
    g = torch.Generator().manual_seed(10 + rank)
    outs = []
    for step in range(2):                                   # two steps: the counters re-arm
        x = torch.randn(16, 7, generator=g)
        flat.zero_grad()
        net(x).pow(2).mean().backward()                     # buckets start reducing (in place) while autograd is still running
        got = ov.finish().clone()
        flat2.zero_grad()
        net2(x).pow(2).mean().backward()
        ref = flat2.flat.grad.clone()
        nqdist.allreduce_mean_(ref)
        outs.append((got, ref))
    torch.save(outs, os.path.join(out_dir, f"ov{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucketed_allreduce_equals_one_flat_allreduce(tmp_path):
    for world in (2, 3):
        port = _free_port()
        mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        res = [torch.load(tmp_path / f"ov{r}.pt") for r in range(world)]
        for step in range(2):
            for r in range(world):
                got, ref = res[r][step]
                if world == 2:
                    assert torch.equal(got, ref)            # two addends: the sum does not depend on how the collective chunks the buffer
                else:
                    assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())      # ring order differs with the chunking: last-bit differences
                assert torch.equal(got, res[0][step][0])    # every rank holds the same mean
            assert float(res[0][step][0].abs().max()) > 0
            assert float(res[0][step][0][-20:].abs().min()) > 0      # the last parameters (l3.weight, l3.bias) have non-zero gradients and were reduced


# ---- forced 1-rank group (NQ_DIST_FORCE=1): the switch the single-GPU RCCL tests rely on, exercised here over gloo -------------------------------------
def _forced_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", NQ_DIST_FORCE="1")
    torch.set_num_threads(1)
    assert not nqdist.active()
    r, w, _ = nqdist.init_from_env(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and nqdist.forced() and nqdist.active()      # a 1-rank group exists and its collectives run
    from nabladft_amd.trainer import FlatParameters, OverlappedAllReduce
    net = _mlp(0)
    flat = FlatParameters(list(net.parameters()))
    ov = OverlappedAllReduce(flat, bucket_bytes=600)
    assert ov.on and ov.world == 1 and len(ov._handles) == len(flat.params)                      # hooks registered although world == 1
    x = torch.randn(16, 7, generator=torch.Generator().manual_seed(3))
    flat.zero_grad()
    net(x).pow(2).mean().backward()
    got = ov.finish().clone()
    assert all(ov._launched[b] is False for b in range(len(ov.buckets)))                         # re-armed
    net2 = _mlp(0)
    flat2 = FlatParameters(list(net2.parameters()))
    flat2.zero_grad()
    net2(x).pow(2).mean().backward()
    ref = flat2.flat.grad.clone()
    t = ref.clone()
    nqdist.allreduce_mean_(t)                                                                     # runs the collective: sum over one rank, no scaling
    nqdist.allreduce_sum_(t)
    nqdist.broadcast_(t, 0)
    nqdist.barrier()
    torch.save((got, ref, t), os.path.join(out_dir, "forced.pt"))
    dist.destroy_process_group()
    os.environ.pop("NQ_DIST_FORCE")
    assert not nqdist.active()


def test_forced_one_rank_group_runs_every_collective(tmp_path):
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got, ref, t = torch.load(tmp_path / "forced.pt")
    assert torch.equal(got, ref) and torch.equal(t, ref) and float(ref.abs().max()) > 0


def _ranks_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), NQ_DIST_BACKEND="gloo")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec = bench.collective_record()
    codes = []
    for expected in (world, world + 1):
        try:
            bench.check_ranks(expected)
            codes.append(0)
        except SystemExit as e:
            codes.append(e.code)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as fh:
        fh.write(f"{rec['ranks_seen']} {rec['backend']} {rec['path']} {codes[0]} {codes[1]}\n")
    dist.destroy_process_group()


def test_rank_count_is_reported_and_a_mismatch_aborts(tmp_path):
    """the bench record carries the number of ranks the collective library sees (config.collective.ranks_seen) and a job whose communicator
    does not span --gpus ranks exits with a non-zero code instead of printing a record (here: a 2-rank gloo group checked against 2 and against 3)."""
    import bench
    assert bench.collective_record()["ranks_seen"] == 1 and bench.collective_record()["backend"] == "none"
    bench.check_ranks(1)
    with pytest.raises(SystemExit) as e:
        bench.check_ranks(2)
    assert e.value.code == 3
    world = 2
    mp.spawn(_ranks_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        seen, backend, path, ok, bad = open(tmp_path / f"r{r}.txt").read().split()
        assert (int(seen), backend, path, int(ok), int(bad)) == (2, "gloo", "torch", 0, 3)
