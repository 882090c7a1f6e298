"""Data-parallel plumbing: one process per GPU, gradient all-reduce over RCCL (backend "nccl" on ROCm)
or gloo (CPU tests).  Conformers are independent graphs, so the flat parameter-gradient buffer is the
only thing ever exchanged (reference: Lightning DDPStrategy, nablaDFT/utils/pipelines.py:65-68)."""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if a launcher set them. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver supports dmabuf IPC only (RCCL)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("NQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of ONE contiguous buffer (a single collective per step; PaiNN: 5.4 MB)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / w)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


# ---- per-conformer cost proxies (one molecule = one graph; the cost of a step is the sum over its conformers) ------------------------------------
# What a model's step time scales with, from its graph construction (reference yaml files):
#   pairs       all ordered atom pairs, n (n - 1): QHNet's pair blocks and PhiSNet's pair features (qhnet.yaml, phisnet args: full graph)
#   capped K    n * min(n - 1, K) directed edges: neighbour-capped radius graphs -- eSCN K = 40 (config/model/escn-oc.yaml:8-21), EquiformerV2 K = 30
#               (equiformer_v2_oc20.yaml:7-41), GemNet-OC K = 30 on its main graph (gemnet-oc.yaml), PaiNN's 5 A cutoff saturates at ~20 neighbours
# plus `node` edge-equivalents of per-atom work (node MLPs / embeddings).  n^2 (round 1's only proxy) over-weights large molecules by 2-3x on capped graphs.
COST_MODELS = {"qhnet": ("pairs", 0, 0.0), "phisnet": ("pairs", 0, 0.0), "escn": ("capped", 40, 2.0), "equiformer_v2": ("capped", 30, 2.0),
               "gemnet_oc": ("capped", 30, 2.0), "painn": ("capped", 20, 4.0), "schnet": ("capped", 20, 2.0), "n2": ("pairs", 0, 1.0)}


def conformer_cost(n: int, model: str = "n2") -> float:
    kind, K, node = COST_MODELS[model]
    if kind == "pairs":
        return float(n) * max(n - 1, 1) + node * n
    return float(n) * (min(max(n - 1, 1), K) + node)


def shard_by_cost(sizes: Sequence[int], world: int, model: str = "n2", cost: Sequence[float] = None) -> List[List[int]]:
    """Greedy longest-processing-time partition of conformers over ranks (keeps the 10-90 atom mix balanced; each rank's list is returned in
    ascending conformer order).  Cost of a conformer: `cost[i]` if given, else conformer_cost(sizes[i], model)."""
    c = [conformer_cost(int(n), model) for n in sizes] if cost is None else [float(x) for x in cost]
    order = sorted(range(len(c)), key=lambda i: (-c[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += c[i]
    return [sorted(o) for o in out]


def predicted_spread(sizes: Sequence[int], world: int, model: str = "n2") -> float:
    """max over ranks of the summed cost / mean - 1 for the partition shard_by_cost produces (pure arithmetic: what an 8-GPU step would wait for)."""
    parts = shard_by_cost(sizes, world, model)
    loads = [sum(conformer_cost(int(sizes[i]), model) for i in part) for part in parts]
    mean = sum(loads) / float(world)
    return max(loads) / mean - 1.0 if mean > 0 else 0.0


def spread_table(model: str, world: int = 8, per_rank=(2, 8, 16, 64), lo: int = 10, hi: int = 90, seed: int = 0, draws: int = 20) -> dict:
    """Predicted cost spread (max rank / mean - 1, averaged over `draws` global batches) of the cost-balanced partition of a U{lo..hi}-atom conformer mix
    over `world` ranks, per conformers-per-rank: what BASELINE.json configs[4]'s "load-balance stress" costs in waiting time, without an 8-GPU box."""
    import random
    rng = random.Random(seed)
    out = {}
    for pr in per_rank:
        acc = 0.0
        for _ in range(draws):
            sizes = [rng.randint(lo, hi) for _ in range(pr * world)]
            acc += predicted_spread(sizes, world, model)
        out[str(pr)] = round(acc / draws, 4)
    return out
