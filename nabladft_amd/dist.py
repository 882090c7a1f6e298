"""Data-parallel plumbing: one process per GPU, gradient all-reduce over RCCL (backend "nccl" on ROCm)
or gloo (CPU tests).  Conformers are independent graphs, so the flat parameter-gradient buffer is the
only thing ever exchanged (reference: Lightning DDPStrategy, nablaDFT/utils/pipelines.py:65-68)."""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def forced() -> bool:
    """NQ_DIST_FORCE=1: initialise the process group and run every collective even in a 1-rank job -- the whole RCCL path of a step (group
    creation, stream-ordered all-reduce, barriers) then executes on a single GPU (tests/test_dist_gpu.py; a 1-GPU box cannot host a second rank)."""
    return os.environ.get("NQ_DIST_FORCE", "0") not in ("", "0")


def init_from_env(backend: str = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if a launcher set them. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver supports dmabuf IPC only (RCCL)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("NQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get("NQ_RCCL_NATIVE", "0") not in ("", "0") and dist.is_initialized() and torch.cuda.is_available():
        native_comm()
    return rank, world, local


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def active(group=None) -> bool:
    """Do the collectives of a step have to run?  A process group exists and it has more than one rank (or NQ_DIST_FORCE=1)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced())


# ---- RCCL through the C ABI (include/nablaq.h: nq_rccl_*, nq_allreduce) ------------------------------------------------------------------------
class NativeComm:
    """An RCCL communicator created and driven through libnablaq's C ABI (SURVEY 8(b) ``nq_allreduce(buf, n, comm, stream)``): rank 0 makes the
    128-byte unique id, the ranks receive it through the torch.distributed store (any backend), every rank joins with its current HIP device.  The
    collectives run on torch's CURRENT stream (the step's stream), so they need no event plumbing.  Opt-in: ``NQ_RCCL_NATIVE=1`` (default: the same
    collectives through torch.distributed's "nccl" backend, which is the same librccl)."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if not lib.nq_rccl_available():
            raise RuntimeError("librccl.so cannot be loaded: " + lib.nq_last_error().decode(errors="replace"))
        self.world = world_size(group)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.nq_rccl_unique_id(ident))
        if self.world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        self._comm = C.c_void_p()
        _lib.check(lib.nq_rccl_comm_create(ident, self.world, self.rank, C.byref(self._comm)))
        self._lib, self._libmod = lib, _lib

    def allreduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._libmod.check(self._lib.nq_allreduce(self._libmod.ptr(t), t.numel(), self._comm, self._libmod.stream_ptr()))
        return t

    def allreduce_mean_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._libmod.check(self._lib.nq_allreduce_mean(self._libmod.ptr(t), t.numel(), self._comm, self._libmod.stream_ptr()))
        return t

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._libmod.check(self._lib.nq_rccl_broadcast(self._libmod.ptr(t), t.numel(), src, self._comm, self._libmod.stream_ptr()))
        return t

    def ranks_seen(self) -> int:
        """ncclCommCount of this communicator: the number of ranks RCCL itself sees (not what the launcher claims)."""
        import ctypes as C
        n = C.c_int32(0)
        self._libmod.check(self._lib.nq_rccl_comm_count(self._comm, C.byref(n)))
        return int(n.value)

    def destroy(self):
        if self._comm:
            self._libmod.check(self._lib.nq_rccl_comm_destroy(self._comm))
            self._comm = None


_native = None


def native_comm(create: bool = True):
    """The process-wide NativeComm of the default group (created on first use when ``create``)."""
    global _native
    if _native is None and create:
        import atexit
        _native = NativeComm()
        atexit.register(drop_native_comm)   # the communicator is destroyed with the process even when the group is torn down outside bench.py
    return _native


def drop_native_comm():
    global _native
    if _native is not None:
        _native.destroy()
        _native = None


def ranks_seen(group=None) -> int:
    """Ranks the collective library of this job sees: ncclCommCount of the native communicator when it exists, else torch.distributed's world size
    (its "nccl" backend creates the RCCL communicator lazily with exactly that count), 1 without a process group."""
    if _native is not None and group is None:
        return _native.ranks_seen()
    return world_size(group)


def _use_native(t, group):
    return _native is not None and group is None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks on the current stream (slices of the flat gradient in the overlapped reverse sweep)."""
    if active(group):
        if _use_native(t, group):
            return _native.allreduce_sum_(t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean over ranks of ONE contiguous buffer (a single collective per step; PaiNN: 5.4 MB)."""
    if active(group):
        if _use_native(flat, group):
            return _native.allreduce_mean_(flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        w = world_size(group)
        if w > 1:
            flat.mul_(1.0 / w)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    if active(group):
        if _use_native(flat, group):
            return _native.broadcast_(flat, src)
        dist.broadcast(flat, src=src, group=group)
    return flat


def barrier(device=None, group=None):
    """dist.barrier with the device pinned for the nccl backend (an unpinned nccl barrier guesses the device from the rank)."""
    if active(group):
        if dist.get_backend(group) == "nccl" and device is not None:
            dist.barrier(group=group, device_ids=[device])
        else:
            dist.barrier(group=group)


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
