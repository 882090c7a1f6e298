"""The step before the hot path: nablaDFT energy databases -> packed conformer arena ->
device batches.

Reference behaviour mirrored here
  * ``PyGNablaDFT.process`` (nablaDFT/dataset/pyg_datasets.py:101-109): every row of the ASE sqlite file becomes
    ``Data(z=numbers.long(), pos=positions.float(), y=data["energy"].float(), forces=data["forces"].float())``;
  * PyG ``Batch.from_data_list`` collate (used by ``PyGNablaDFTDataModule``, nablaDFT/dataset/nablaDFT_dataset.py:223-286):
    concatenation of pos / z / forces, ``y`` stacked, ``batch`` = graph index per atom, ``ptr`` = first atom of every graph.
The reference does this with ase.db + per-sample Python objects + DataLoader workers.  At >20 k conformers/s per GPU that
path is the bottleneck, so here the whole split lives in ONE packed host arena (pinned when a GPU is present), a batch is a
handful of vectorised gathers into a pinned staging buffer, and the host->device copy of batch k+1 runs on its own HIP
stream while batch k is in the kernels.  The neighbour list is built on the GPU (graph.hip), never on the host.

No ase / apsw needed: the ASE sqlite format is read with the standard library (``systems`` table: ``numbers`` int32 blob,
``positions`` float64 blob, ``data`` = ASE's ``object_to_bytes`` container: int64 offset of a trailing JSON document whose
``{"__ndarray__": [shape, dtype, offset]}`` entries point back into the blob).
"""
import json
import sqlite3
import struct
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from .trainer import Batch


# ---- on-disk format ----------------------------------------------------------------------------------------------------------
def _decode_ase_blob(blob: bytes) -> dict:
    """ASE db ``data`` column (ase.db.core.object_to_bytes): <int64 json_offset> <raw arrays ...> <json>; older files store plain JSON."""
    if blob is None:
        return {}
    if isinstance(blob, str) or blob[:1] == b"{":
        return json.loads(blob)
    (off,) = struct.unpack_from("<q", blob, 0)
    doc = json.loads(blob[off:].decode())

    def resolve(o):
        if isinstance(o, dict):
            if "__ndarray__" in o:
                shape, dtype, offset = o["__ndarray__"]
                count = int(np.prod(shape)) if len(shape) else 1
                return np.frombuffer(blob, dtype=np.dtype(dtype), count=count, offset=offset).reshape(shape)
            return {k: resolve(v) for k, v in o.items()}
        return o

    return resolve(doc)


def read_energy_database(path: str, indices: Optional[Sequence[int]] = None) -> "ConformerArena":
    """Reads an nablaDFT energy database (ASE sqlite) into a packed arena.  ``indices``: 0-based row positions in id order
    (the order ``ase.db.connect(path).select()`` yields, pyg_datasets.py:103)."""
    con = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    try:
        rows = con.execute("select numbers, positions, data from systems order by id").fetchall()
    finally:
        con.close()
    if indices is not None:
        rows = [rows[i] for i in indices]
    z, pos, y, f = [], [], [], []
    for numbers, positions, data in rows:
        zz = np.frombuffer(numbers, dtype=np.int32)
        pp = np.frombuffer(positions, dtype=np.float64).reshape(-1, 3)
        d = _decode_ase_blob(data)
        if "energy" not in d or "forces" not in d:
            raise KeyError(f"{path}: row without data['energy'] / data['forces'] (pyg_datasets.py:106-107)")
        ff = np.asarray(d["forces"], dtype=np.float64).reshape(-1, 3)
        if ff.shape != pp.shape or zz.shape[0] != pp.shape[0]:
            raise ValueError(f"{path}: inconsistent row shapes {zz.shape} {pp.shape} {ff.shape}")
        z.append(zz), pos.append(pp.astype(np.float32)), f.append(ff.astype(np.float32))
        y.append(np.asarray(d["energy"], dtype=np.float64).reshape(-1)[:1].astype(np.float32))
    return ConformerArena.from_lists(pos, z, y, f)


# ---- packed host arena -------------------------------------------------------------------------------------------------------
class ConformerArena:
    """All conformers of a split in five contiguous host tensors: pos f32[N,3], z i64[N], forces f32[N,3], y f32[M], ptr i64[M+1]."""

    def __init__(self, pos, z, y, forces, ptr):
        self.pos, self.z, self.y, self.forces, self.ptr = pos, z, y, forces, ptr
        self.sizes = (ptr[1:] - ptr[:-1])
        assert pos.shape[0] == z.shape[0] == forces.shape[0] == int(ptr[-1]) and y.shape[0] == ptr.shape[0] - 1

    @classmethod
    def from_lists(cls, pos: List[np.ndarray], z: List[np.ndarray], y: List[np.ndarray], forces: List[np.ndarray]) -> "ConformerArena":
        sizes = np.array([len(a) for a in z], dtype=np.int64)
        ptr = np.concatenate([[0], np.cumsum(sizes)])
        cat = lambda xs, shape, dt: torch.from_numpy(np.concatenate(xs).astype(dt)) if xs else torch.zeros(shape, dtype=getattr(torch, np.dtype(dt).name))
        return cls(cat(pos, (0, 3), np.float32), cat(z, (0,), np.int64), cat(y, (0,), np.float32), cat(forces, (0, 3), np.float32), torch.from_numpy(ptr))

    @classmethod
    def from_batch(cls, b: Batch) -> "ConformerArena":
        ptr = b.ptr.cpu().long()
        return cls(b.pos.cpu().float(), b.z.cpu().long(), b.y.cpu().float(), b.forces.cpu().float(), ptr)

    def __len__(self):
        return self.y.shape[0]

    def pin(self):
        """Page-locks the arena (no-op without a GPU) so that batch gathers can be copied asynchronously."""
        if torch.cuda.is_available():
            for name in ("pos", "z", "y", "forces"):
                t = getattr(self, name)
                if not t.is_pinned():
                    setattr(self, name, t.pin_memory())
        return self

    def atom_index(self, conformers: torch.Tensor):
        """(atom rows, new ptr) of the selected conformers in the given order -- vectorised, no per-conformer Python."""
        sz = self.sizes[conformers]
        new_ptr = torch.cat([sz.new_zeros(1), sz.cumsum(0)])
        total = int(new_ptr[-1])
        owner = torch.repeat_interleave(torch.arange(len(conformers)), sz, output_size=total)
        rows = self.ptr[conformers][owner] + (torch.arange(total) - new_ptr[:-1][owner])
        return rows, new_ptr, owner

    def batch(self, conformers, out: Optional["_Staging"] = None) -> Batch:
        """Host-side collate (== PyG Batch.from_data_list on these samples).  With ``out`` the gathers go into pinned staging."""
        conformers = torch.as_tensor(conformers, dtype=torch.long)
        rows, new_ptr, owner = self.atom_index(conformers)
        if out is None:
            return Batch(self.pos[rows], self.z[rows], owner, self.y[conformers], self.forces[rows], new_ptr)
        n, m = rows.shape[0], conformers.shape[0]
        out.reserve(n, m)
        torch.index_select(self.pos, 0, rows, out=out.pos[:n])
        torch.index_select(self.z, 0, rows, out=out.z[:n])
        torch.index_select(self.forces, 0, rows, out=out.forces[:n])
        torch.index_select(self.y, 0, conformers, out=out.y[:m])
        out.batch[:n].copy_(owner)
        out.ptr[:m + 1].copy_(new_ptr)
        return Batch(out.pos[:n], out.z[:n], out.batch[:n], out.y[:m], out.forces[:n], out.ptr[:m + 1])


class _Staging:
    """Grow-only pinned host buffers for one in-flight batch."""

    def __init__(self, pin: bool):
        self.pin, self.cap_n, self.cap_m = pin, 0, 0

    def reserve(self, n, m):
        if n > self.cap_n or m > self.cap_m:
            self.cap_n, self.cap_m = max(n, int(self.cap_n * 1.25)), max(m, int(self.cap_m * 1.25))
            mk = lambda *s, dt: torch.empty(*s, dtype=dt, pin_memory=self.pin)
            self.pos, self.forces = mk(self.cap_n, 3, dt=torch.float32), mk(self.cap_n, 3, dt=torch.float32)
            self.z, self.batch = mk(self.cap_n, dt=torch.long), mk(self.cap_n, dt=torch.long)
            self.y, self.ptr = mk(self.cap_m, dt=torch.float32), mk(self.cap_m + 1, dt=torch.long)


# ---- epoch plan + overlapped feed ---------------------------------------------------------------------------------------------
def epoch_plan(sizes: torch.Tensor, batch_size: int, shuffle: bool, seed: int, epoch: int, rank: int = 0, world: int = 1,
               drop_last: bool = False) -> List[torch.Tensor]:
    """Conformer indices of every step of one epoch for this rank.  All ranks draw the same permutation (seed, epoch), cut it
    into global batches of ``batch_size * world`` and split each by estimated cost (``dist.shard_by_cost``: edges ~ n^2 up to
    the cutoff sphere) so that every rank runs the same number of steps with balanced work -- one molecule = one graph, no
    collective besides the gradient all-reduce."""
    from .dist import shard_by_cost
    m = sizes.shape[0]
    if shuffle:
        g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
        order = torch.randperm(m, generator=g)
    else:
        order = torch.arange(m)
    gb = batch_size * world
    steps = []
    for s in range(0, m, gb):
        chunk = order[s:s + gb]
        if chunk.shape[0] < gb and (drop_last or chunk.shape[0] < world):
            break
        if world == 1:
            steps.append(chunk)
        else:
            parts = shard_by_cost(sizes[chunk].tolist(), world)
            steps.append(chunk[torch.as_tensor(parts[rank], dtype=torch.long)])
    return steps


class ArenaLoader:
    """Iterates device ``Batch``es over an arena.  On a GPU: pinned double-buffered staging, host->device copies on a side
    stream, one event per batch; the consumer stream waits on the event only (no host synchronisation)."""

    def __init__(self, arena: ConformerArena, batch_size: int, device, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1,
                 drop_last: bool = False, depth: int = 2):
        self.arena, self.batch_size, self.device = arena, batch_size, torch.device(device)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.epoch = 0
        self.gpu = self.device.type == "cuda"
        if self.gpu:
            arena.pin()
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self.staging = [_Staging(self.gpu) for _ in range(max(depth, 2))]
        self.staging_free = [None] * len(self.staging)      # event: the copy that last read this staging buffer has finished

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return len(epoch_plan(self.arena.sizes, self.batch_size, False, 0, 0, self.rank, self.world, self.drop_last))

    def _stage(self, conformers, slot):
        if not self.gpu:
            return self.arena.batch(conformers), None
        if self.staging_free[slot] is not None:
            self.staging_free[slot].synchronize()          # the pinned buffer is about to be overwritten by the host
        hb = self.arena.batch(conformers, out=self.staging[slot])
        with torch.cuda.stream(self.copy_stream):
            mv = lambda t: t.to(self.device, non_blocking=True)
            db = Batch(mv(hb.pos), mv(hb.z), mv(hb.batch), mv(hb.y), mv(hb.forces), mv(hb.ptr))
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.staging_free[slot] = ev
        return db, ev

    def __iter__(self) -> Iterator[Batch]:
        plan = epoch_plan(self.arena.sizes, self.batch_size, self.shuffle, self.seed, self.epoch, self.rank, self.world, self.drop_last)
        self.epoch += 1
        if not plan:
            return
        staged = self._stage(plan[0], 0)
        for k in range(len(plan)):
            db, ev = staged
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in (db.pos, db.z, db.batch, db.y, db.forces, db.ptr):
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield db
            # resumed after the consumer has ENQUEUED its kernels for batch k: collate + copy batch k+1 while they run
            if k + 1 < len(plan):
                staged = self._stage(plan[k + 1], (k + 1) % len(self.staging))
