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
               drop_last: bool = False, cost_model: str = "painn") -> List[torch.Tensor]:
    """Conformer indices of every step of one epoch for this rank.  All ranks draw the same permutation (seed, epoch), cut it
    into global batches of ``batch_size * world`` and split each by estimated cost (``dist.shard_by_cost`` with the model's proxy,
    ``dist.COST_MODELS``: n * min(n - 1, K) edges for neighbour-capped graphs, n (n - 1) pairs for the Hamiltonian models) so that every rank runs the same number of steps with balanced work -- one molecule = one graph, no
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
            parts = shard_by_cost(sizes[chunk].tolist(), world, cost_model)
            steps.append(chunk[torch.as_tensor(parts[rank], dtype=torch.long)])
    return steps


class ArenaLoader:
    """Iterates device ``Batch``es over an arena.  On a GPU: pinned double-buffered staging, host->device copies on a side
    stream, one event per batch; the consumer stream waits on the event only (no host synchronisation)."""

    def __init__(self, arena: ConformerArena, batch_size: int, device, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1,
                 drop_last: bool = False, depth: int = 2, cost_model: str = "painn"):
        self.arena, self.batch_size, self.device = arena, batch_size, torch.device(device)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.cost_model = cost_model          # dist.COST_MODELS key: how the per-rank split of a global batch weighs a conformer of n atoms
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
        return len(epoch_plan(self.arena.sizes, self.batch_size, False, 0, 0, self.rank, self.world, self.drop_last, self.cost_model))

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
        plan = epoch_plan(self.arena.sizes, self.batch_size, self.shuffle, self.seed, self.epoch, self.rank, self.world, self.drop_last, self.cost_model)
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


# ---- Hamiltonian databases (second half) ----------------------------------------------------------
class HamiltonianDatabase:
    """nablaDFT Hamiltonian sqlite file, read (and written) with the standard library -- the accessors of
    nablaDFT/dataset/hamiltonian_dataset.py:17-283 (which needs ``apsw``), same names and return values:
      ``len(db)``                       metadata row 0 (:76-78)
      ``db[i]`` / ``db[[i, j, ...]]``     (Z i32[N], R f32[N,3], E f32[1], F f32[N,3], H, S, C f32[Norb,Norb], moses_id, conformer_id) (:80-106)
      ``db.get_orbitals(Z)``            angular momenta of the element's shells, i32 (:173-178)
      ``db.Z``                          elements of the basis-set table, i32 (:277-283)
      ``add_data / add_orbitals / add_Z``  the writers (:108-171, 266-275): float64 -> float32, int64 -> int32 blobs, little endian.
    File format: tables ``data(id, Z, R, E, F, H, S, C)``, ``dataset_ids(id, MOSES_ID, CONFORMER_ID)``, ``basisset(Z, orbitals)``,
    ``nuclear_charges(id, N, Z)``, ``metadata(id, N)`` (:210-257)."""

    def __init__(self, filename: str, readonly: bool = True):
        import os
        self.filename = filename
        new = not os.path.isfile(filename)
        if new and readonly:
            raise FileNotFoundError(filename)
        self._c = sqlite3.connect(("file:" + filename + "?mode=ro") if readonly else filename, uri=readonly, isolation_level=None, check_same_thread=False)
        self._c.execute("PRAGMA busy_timeout=300000")
        if new:
            cur = self._c.cursor()
            cur.execute("CREATE TABLE IF NOT EXISTS dataset_ids (id INTEGER NOT NULL PRIMARY KEY, MOSES_ID INT, CONFORMER_ID INT)")
            cur.execute("CREATE TABLE IF NOT EXISTS data (id INTEGER NOT NULL PRIMARY KEY, Z BLOB, R BLOB, E FLOAT, F BLOB, H BLOB, S BLOB, C BLOB)")
            cur.execute("CREATE TABLE IF NOT EXISTS nuclear_charges (id INTEGER NOT NULL PRIMARY KEY, N INTEGER, Z BLOB)")
            cur.execute("INSERT OR IGNORE INTO nuclear_charges (id, N, Z) VALUES (?,?,?)", (0, 1, self._blob(np.array([0]))))
            cur.execute("CREATE TABLE IF NOT EXISTS basisset (Z INTEGER NOT NULL PRIMARY KEY, orbitals BLOB)")
            cur.execute("CREATE TABLE IF NOT EXISTS metadata (id INTEGER PRIMARY KEY, N INTEGER)")
            cur.execute("INSERT OR IGNORE INTO metadata (id, N) VALUES (?,?)", (0, 0))
        self._orbitals = {}

    @staticmethod
    def _blob(a):
        if a is None:
            return None
        a = np.asarray(a)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        if a.dtype == np.int64:
            a = a.astype(np.int32)
        return memoryview(np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False)))

    @staticmethod
    def _deblob(buf, dtype, shape):
        if buf is None:
            return np.zeros(shape)
        return np.frombuffer(buf, np.dtype(dtype).newbyteorder("<")).astype(dtype, copy=False).reshape(shape)

    def __len__(self):
        return int(self._c.execute("SELECT * FROM metadata WHERE id=0").fetchone()[-1])

    def _unpack(self, d):
        n = len(d[2]) // 12
        norb = int(round((len(d[5]) // 4) ** 0.5))
        return (self._deblob(d[1], np.int32, (n,)), self._deblob(d[2], np.float32, (n, 3)), np.array([0.0 if d[3] is None else d[3]], dtype=np.float32),
                self._deblob(d[4], np.float32, (n, 3)), self._deblob(d[5], np.float32, (norb, norb)), self._deblob(d[6], np.float32, (norb, norb)),
                self._deblob(d[7], np.float32, (norb, norb)))

    def __getitem__(self, idx):
        cur = self._c.cursor()
        if isinstance(idx, (list, tuple, np.ndarray)):       # batched retrieval: rows come back in id order, like the reference's IN (...) query
            ids = [int(i) for i in idx]
            q = ",".join(str(i) for i in ids)
            data = cur.execute("SELECT * FROM data WHERE id IN (" + q + ")").fetchall()
            names = cur.execute("SELECT * FROM dataset_ids WHERE id IN (" + q + ")").fetchall()
            return [(*self._unpack(d), nm[1], nm[2]) for d, nm in zip(data, names)]
        d = cur.execute("SELECT * FROM data WHERE id=" + str(int(idx))).fetchone()
        if d is None:
            raise IndexError(idx)
        nm = cur.execute("SELECT * FROM dataset_ids WHERE id=" + str(int(idx))).fetchone()
        return (*self._unpack(d), nm[1], nm[2])

    def get_orbitals(self, Z):
        Z = int(Z)
        if Z not in self._orbitals:
            d = self._c.execute("SELECT * FROM basisset WHERE Z=" + str(Z)).fetchone()
            if d is None:
                raise KeyError(f"element {Z} has no entry in the basisset table")
            self._orbitals[Z] = self._deblob(d[1], np.int32, (len(d[1]) // 4,))
        return self._orbitals[Z]

    @property
    def Z(self):
        d = self._c.execute("SELECT * FROM nuclear_charges WHERE id=0").fetchone()
        return self._deblob(d[2], np.int32, (d[1],))

    # ---- writers ----
    def add_data(self, Z, R, E, F, H, S, C, moses_id, conformer_id):
        if any(v is not None and np.any(np.isnan(v)) for v in (Z, R, E, F, H, S, C)):
            print("encountered NaN, data is not added")
            return
        cur = self._c.cursor()
        cur.execute("BEGIN EXCLUSIVE")
        try:
            length = len(self)
            rid = None if length > 0 else 0
            cur.execute("INSERT INTO dataset_ids (id, MOSES_ID, CONFORMER_ID) VALUES (?,?,?)", (rid, moses_id, conformer_id))
            cur.execute("INSERT INTO data (id, Z, R, E, F, H, S, C) VALUES (?,?,?,?,?,?,?,?)",
                        (rid, self._blob(Z), self._blob(R), None if E is None else float(E), self._blob(F), self._blob(H), self._blob(S), self._blob(C)))
            cur.execute("INSERT OR REPLACE INTO metadata VALUES (?,?)", (0, length + 1))
            cur.execute("COMMIT")
        except Exception:
            cur.execute("ROLLBACK")
            raise

    def add_orbitals(self, Z, orbitals):
        self._c.execute("INSERT OR REPLACE INTO basisset (Z, orbitals) VALUES (?,?)", (int(Z), self._blob(np.asarray(orbitals))))
        self._orbitals.pop(int(Z), None)

    def add_Z(self, Z):
        self._c.execute("INSERT OR REPLACE INTO nuclear_charges (id, N, Z) VALUES (?,?,?)", (0, len(Z), self._blob(np.asarray(Z))))


class HamiltonianDataset(torch.utils.data.Dataset):
    """nablaDFT/dataset/hamiltonian_dataset.py:286-405: items are row ids, ``collate_fn`` does one batched query and returns the dict PhiSNet's
    ``NeuralNetwork.forward`` takes (``molecule_size``, ``atomic_numbers``, ``orbitals``, ``positions``, ``energy``, ``forces``, block-diagonal
    ``full_hamiltonian`` / ``overlap_matrix`` / ``core_hamiltonian`` / ``mask``), cut at the same three batch limits.  Extra keys for the packed
    engine path: ``*_packed`` = the molecules' matrices flattened and concatenated (what ``IrrepsAssembler`` / ``BlockAssembler`` produce)."""

    def __init__(self, filepath, max_batch_orbitals=1200, max_batch_atoms=150, max_squares=4802, subset=None, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype
        self._database = HamiltonianDatabase(filepath)
        self.max_orbitals = tuple(tuple((int(z), int(l)) for l in self._database.get_orbitals(z)) for z in self._database.Z)
        self.max_batch_orbitals, self.max_batch_atoms, self.max_squares = max_batch_orbitals, max_batch_atoms, max_squares
        self.subset = np.load(subset) if subset else None

    def __len__(self):
        return len(self.subset) if self.subset is not None else len(self._database)

    def __getitem__(self, idx):
        return self.subset[idx] if self.subset is not None else idx

    def collate_fn(self, batch, return_filtered=False):
        rows = self._database[list(batch)]
        Z, R, E, F, H, S, C = [], [], [], [], [], [], []
        orbitals, n_orb, squares = [], 0, 0
        for Z_, R_, E_, F_, H_, S_, C_, _, _ in rows:
            local = [tuple((int(z), int(l)) for l in self._database.get_orbitals(z)) for z in Z_]
            local_n = sum(2 * l + 1 for orbs in local for _, l in orbs)
            if n_orb + local_n > self.max_batch_orbitals or len(local) + len(orbitals) > self.max_batch_atoms or squares + len(local) ** 2 > self.max_squares:
                break
            orbitals += local
            n_orb += local_n
            squares += len(local) ** 2
            Z.append(torch.tensor(Z_, dtype=torch.int64)), R.append(torch.tensor(R_, dtype=self.dtype)), E.append(torch.tensor(E_, dtype=self.dtype))
            F.append(torch.tensor(F_, dtype=self.dtype)), H.append(torch.tensor(H_, dtype=self.dtype)), S.append(torch.tensor(S_, dtype=self.dtype))
            C.append(torch.tensor(C_, dtype=self.dtype))
        out = {"molecule_size": torch.tensor([len(z) for z in Z]), "atomic_numbers": torch.cat(Z), "orbitals": tuple(orbitals), "positions": torch.cat(R),
               "energy": torch.stack(E), "forces": torch.cat(F), "full_hamiltonian": torch.block_diag(*H), "overlap_matrix": torch.block_diag(*S),
               "core_hamiltonian": torch.block_diag(*C), "mask": torch.block_diag(*(torch.ones_like(c) for c in C)),
               "full_hamiltonian_packed": torch.cat([h.reshape(-1) for h in H]), "overlap_matrix_packed": torch.cat([s.reshape(-1) for s in S]),
               "core_hamiltonian_packed": torch.cat([c.reshape(-1) for c in C])}
        if return_filtered:
            out["filtered"] = batch[len(Z):]
        return out


class HamiltonianBatch(Batch):
    """PyG batch of ``PyGHamiltonianNablaDFT`` items (nablaDFT/dataset/pyg_datasets.py:198-222 + Batch.from_data_list): tensors concatenated,
    ``y`` one energy per molecule, the matrices as LISTS of per-molecule numpy arrays (what PyG's collate does with non-tensor attributes and
    what ``QHNetLightning.step`` consumes, qhnet/qhnet.py:366-377; ``hamiltonian.BlockAssembler.pack_targets`` packs them for the HIP loss)."""

    def __init__(self, pos, z, batch, y, forces, ptr, hamiltonian, overlap=None, core=None):
        super().__init__(pos, z, batch, y, forces, ptr)
        self.hamiltonian, self.overlap, self.core = hamiltonian, overlap, core

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return HamiltonianBatch(mv(self.pos), mv(self.z), mv(self.batch), mv(self.y), mv(self.forces), mv(self.ptr), self.hamiltonian, self.overlap, self.core)


def hamiltonian_batch(db: HamiltonianDatabase, indices: Sequence[int], include_overlap: bool = False, include_core: bool = False,
                      dtype=torch.float32) -> HamiltonianBatch:
    """Rows ``indices`` of a Hamiltonian database as one batch, in the given order (one batched query; the reference dataset reads row by row)."""
    idx = [int(i) for i in indices]
    uniq = sorted(set(idx))                                  # the batched query returns each row once, in id order
    by_id = dict(zip(uniq, db[uniq]))
    rows = [by_id[i] for i in idx]                           # repeated indices repeat the row
    sizes = torch.tensor([len(r[0]) for r in rows], dtype=torch.long)
    ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
    return HamiltonianBatch(
        pos=torch.cat([torch.tensor(r[1].copy()).to(dtype) for r in rows]), z=torch.cat([torch.tensor(r[0].copy()).long() for r in rows]),
        batch=torch.repeat_interleave(torch.arange(len(rows)), sizes), y=torch.cat([torch.from_numpy(r[2].copy()).to(dtype) for r in rows]),
        forces=torch.cat([torch.from_numpy(r[3].copy()).to(dtype) for r in rows]), ptr=ptr, hamiltonian=[r[4].copy() for r in rows],
        overlap=[r[5].copy() for r in rows] if include_overlap else None, core=[r[6].copy() for r in rows] if include_core else None)
