"""CPU: the step before the hot path: ASE-sqlite energy database -> packed arena -> batches.
Fixture: tests/golden/energy_db_30.db = 30 verbatim rows of the data file the reference's dataset tests use, with the reference
tests' own known answers (tests/dataset/test_pyg_datasets.py:13-28) and the arrays PyGNablaDFT.process would build."""
import os

import numpy as np
import pytest
import torch

from nabladft_amd import data as D
from tests.helpers import GOLDEN

DB = os.path.join(GOLDEN, "energy_db_30.db")


def test_energy_database_reader_known_answers_and_arrays():
    arena = D.read_energy_database(DB)
    fx = np.load(os.path.join(GOLDEN, "energy_db_30.npz"))
    assert len(arena) == 30
    # the reference's own assertions on this file: sample 0 and the slice 15:30
    b0 = arena.batch([0])
    assert b0.y.shape == (1,) and b0.z.shape == (40,) and b0.pos.shape == b0.forces.shape == (40, 3)
    assert b0.z.dtype == torch.long and b0.pos.dtype == b0.forces.dtype == b0.y.dtype == torch.float32
    bs = arena.batch(range(15, 30))
    assert bs.y.shape == (15,) and bs.pos.shape == bs.forces.shape == (610, 3) and bs.z.shape == (610,)
    for k in ("pos", "z", "y", "forces", "ptr"):
        assert np.array_equal(getattr(arena, k).numpy(), fx[k]), k
    sub = D.read_energy_database(DB, indices=[3, 7])
    assert torch.equal(sub.pos, arena.batch([3, 7]).pos)
    with pytest.raises(sqlite3_error()):
        D.read_energy_database(os.path.join(GOLDEN, "does_not_exist.db"))


def sqlite3_error():
    import sqlite3
    return sqlite3.OperationalError


def test_arena_batch_equals_naive_collate_and_staging():
    arena = D.read_energy_database(DB)
    sel = [5, 0, 29, 12, 12]
    b = arena.batch(sel)
    pos = torch.cat([arena.pos[arena.ptr[i]:arena.ptr[i + 1]] for i in sel])
    z = torch.cat([arena.z[arena.ptr[i]:arena.ptr[i + 1]] for i in sel])
    f = torch.cat([arena.forces[arena.ptr[i]:arena.ptr[i + 1]] for i in sel])
    batch = torch.cat([torch.full((int(arena.sizes[i]),), m) for m, i in enumerate(sel)])
    assert torch.equal(b.pos, pos) and torch.equal(b.z, z) and torch.equal(b.forces, f) and torch.equal(b.batch, batch)
    assert torch.equal(b.y, arena.y[sel]) and torch.equal(b.ptr, torch.tensor([0] + list(np.cumsum([int(arena.sizes[i]) for i in sel]))))
    st = D._Staging(pin=False)
    b2 = arena.batch(sel, out=st)
    for k in ("pos", "z", "batch", "y", "forces", "ptr"):
        assert torch.equal(getattr(b2, k), getattr(b, k)), k
    b3 = arena.batch([1], out=st)                       # staging is reused (grow-only)
    assert b3.pos.shape[0] == int(arena.sizes[1]) and b3.pos.data_ptr() == st.pos.data_ptr()


def test_epoch_plan_covers_split_and_balances_ranks():
    arena = D.read_energy_database(DB)
    for world in (1, 2, 4):
        seen = []
        per_rank_steps = set()
        for rank in range(world):
            plan = D.epoch_plan(arena.sizes, 4, True, 7, 0, rank, world)
            per_rank_steps.add(len(plan))
            seen += [int(i) for s in plan for i in s]
        assert len(per_rank_steps) == 1                                   # same number of steps on every rank (collective safety)
        assert len(seen) == len(set(seen))                                # disjoint
        assert len(seen) >= 30 - 30 % world - (4 * world - 1) or len(seen) == 30
    p0 = D.epoch_plan(arena.sizes, 8, True, 7, 0)
    p1 = D.epoch_plan(arena.sizes, 8, True, 7, 1)
    assert sorted(int(i) for s in p0 for i in s) == list(range(30)) and any(not torch.equal(a, b) for a, b in zip(p0, p1))
    # cost balance at world 2: |cost_0 - cost_1| is at most one largest molecule
    a = D.epoch_plan(arena.sizes, 15, False, 0, 0, 0, 2)[0]
    b = D.epoch_plan(arena.sizes, 15, False, 0, 0, 1, 2)[0]
    ca, cb = (arena.sizes[a] ** 2).sum(), (arena.sizes[b] ** 2).sum()
    assert abs(int(ca) - int(cb)) <= int(arena.sizes.max()) ** 2


def test_loader_on_cpu_yields_the_plan():
    arena = D.read_energy_database(DB)
    ld = D.ArenaLoader(arena, 7, "cpu", shuffle=True, seed=3)
    plan = D.epoch_plan(arena.sizes, 7, True, 3, 0)
    got = list(ld)
    assert len(got) == len(plan) == len(ld) == 5
    for bt, sel in zip(got, plan):
        ref = arena.batch(sel)
        assert torch.equal(bt.pos, ref.pos) and torch.equal(bt.y, ref.y) and torch.equal(bt.ptr, ref.ptr)
    assert not torch.equal(next(iter(ld)).y, got[0].y)                    # next epoch: new permutation


# ---- Hamiltonian databases (row f2) ------------------------------------------------------------------------------------------
def _hamdb():
    return os.path.join(GOLDEN, "hamiltonian_db_6.db"), np.load(os.path.join(GOLDEN, "hamiltonian_db_6.npz"))


def test_hamiltonian_database_reads_like_reference():
    """Against what the REAL HamiltonianDatabase returned for the same file (oracle/make_golden_hamdb.py): bit-exact."""
    from nabladft_amd.data import HamiltonianDatabase
    path, fx = _hamdb()
    db = HamiltonianDatabase(path)
    assert len(db) == int(fx["len"]) == 6
    assert np.array_equal(db.Z, fx["Z_table"]) and db.Z.dtype == np.int32
    for zz in (1, 6, 7, 8):
        assert np.array_equal(db.get_orbitals(zz), fx[f"orbitals_{zz}"])
    for i in range(6):
        row = db[i]
        for name, v in zip(("Z", "R", "E", "F", "H", "S", "C", "moses_id", "conformer_id"), row):
            ref = fx[f"row{i}:{name}"]
            assert np.array_equal(np.asarray(v), ref), (i, name)
            if name in "ZRFHSC" and name != "E":
                assert np.asarray(v).dtype == ref.dtype, (i, name)
    some = db[[int(v) for v in fx["list_rows"]]]
    for j, row in enumerate(some):
        assert np.array_equal(row[4], fx[f"list{j}:H"]) and row[7] == int(fx[f"list{j}:moses_id"])
    with pytest.raises(KeyError):
        db.get_orbitals(35)
    with pytest.raises(FileNotFoundError):
        HamiltonianDatabase(path + ".missing")


def test_hamiltonian_dataset_collate_like_reference():
    from nabladft_amd.data import HamiltonianDataset
    path, fx = _hamdb()
    ds = HamiltonianDataset(path)
    mo = fx["max_orbitals"]
    assert ds.max_orbitals == tuple(tuple((int(a), int(b)) for a, b in orbs if a >= 0) for orbs in mo)
    assert len(ds) == 6 and ds[3] == 3
    b = ds.collate_fn([0, 1, 3])
    for k in ("molecule_size", "atomic_numbers", "positions", "energy", "forces", "full_hamiltonian", "overlap_matrix", "core_hamiltonian", "mask"):
        ref = fx["batch:" + k]
        assert tuple(b[k].shape) == ref.shape and np.array_equal(b[k].numpy(), ref), k
        assert str(b[k].dtype).split(".")[-1] == str(ref.dtype), k
    flat = [t for orbs in b["orbitals"] for t in orbs]
    assert np.array_equal(np.array(flat), fx["batch:orbitals_flat"]) and [len(o) for o in b["orbitals"]] == list(fx["batch:orbitals_count"])
    # packed targets = the diagonal blocks, molecule by molecule
    o, chunks = 0, []
    for n in (24, 34, 38):
        chunks.append(b["full_hamiltonian"][o:o + n, o:o + n].reshape(-1))
        o += n
    assert o == b["full_hamiltonian"].shape[0] and torch.equal(torch.cat(chunks), b["full_hamiltonian_packed"])
    tight = HamiltonianDataset(path, max_batch_atoms=8)
    b2 = tight.collate_fn([1, 3, 0], return_filtered=True)
    assert np.array_equal(b2["molecule_size"].numpy(), fx["tight:molecule_size"]) and list(b2["filtered"]) == list(fx["tight:filtered"])


def test_hamiltonian_database_write_read_roundtrip(tmp_path):
    from nabladft_amd.data import HamiltonianDatabase
    p = str(tmp_path / "w.db")
    db = HamiltonianDatabase(p, readonly=False)
    db.add_orbitals(1, np.array([0, 0, 1]))
    db.add_Z(np.array([1]))
    rng = np.random.default_rng(0)
    H = rng.normal(size=(10, 10))
    db.add_data(np.array([1, 1]), rng.normal(size=(2, 3)), -1.25, rng.normal(size=(2, 3)), H, H * 2, H * 3, 5, 9)
    db.add_data(np.array([1, 1]), np.full((2, 3), np.nan), -1.0, rng.normal(size=(2, 3)), H, H, H, 6, 10)      # NaN rows are refused (:122-125)
    rd = HamiltonianDatabase(p)
    assert len(rd) == 1
    Z, R, E, F, Hh, S, C, mid, cid = rd[0]
    assert Z.dtype == np.int32 and Hh.dtype == np.float32 and np.array_equal(Hh, H.astype(np.float32)) and np.array_equal(C, (H * 3).astype(np.float32))
    assert E[0] == np.float32(-1.25) and (mid, cid) == (5, 9) and list(rd.get_orbitals(1)) == [0, 0, 1]


def test_hamiltonian_batch_matches_the_database_rows():
    """PyG-style batch for the QHNet side (pyg_datasets.py:198-222): concatenated tensors, matrices as per-molecule lists, any row order."""
    from nabladft_amd.data import HamiltonianDatabase, hamiltonian_batch
    path, fx = _hamdb()
    db = HamiltonianDatabase(path)
    order = [3, 0, 5]
    b = hamiltonian_batch(db, order, include_overlap=True)
    sizes = [len(fx[f"row{i}:Z"]) for i in order]
    assert b.ptr.tolist() == [0] + list(np.cumsum(sizes)) and b.num_nodes == sum(sizes)
    assert b.z.dtype == torch.long and b.pos.dtype == torch.float32 and b.y.shape == (3,)
    o = 0
    for j, i in enumerate(order):
        n = sizes[j]
        assert np.array_equal(b.z[o:o + n].numpy(), fx[f"row{i}:Z"]) and np.array_equal(b.pos[o:o + n].numpy(), fx[f"row{i}:R"])
        assert np.array_equal(b.forces[o:o + n].numpy(), fx[f"row{i}:F"]) and float(b.y[j]) == float(fx[f"row{i}:E"][0])
        assert np.array_equal(b.hamiltonian[j], fx[f"row{i}:H"]) and np.array_equal(b.overlap[j], fx[f"row{i}:S"])
        assert (b.batch[o:o + n] == j).all()
        o += n
    assert b.core is None
    # the list is what BlockAssembler.pack_targets flattens: per-molecule row-major blocks
    flat = np.concatenate([h.reshape(-1) for h in b.hamiltonian])
    assert flat.size == sum(h.shape[0] ** 2 for h in b.hamiltonian)


def test_hamiltonian_batch_with_repeated_and_unordered_indices():
    """A repeated index repeats the row (the batched query returns every id once, in id order)."""
    from nabladft_amd.data import HamiltonianDatabase, hamiltonian_batch
    db = HamiltonianDatabase(os.path.join(GOLDEN, "hamiltonian_db_6.db"))
    order = [3, 3, 5, 0, 3]
    b = hamiltonian_batch(db, order)
    rows = [db[i] for i in order]
    assert [int(x) for x in (b.ptr[1:] - b.ptr[:-1])] == [len(r[0]) for r in rows]
    for k, r in enumerate(rows):
        assert np.array_equal(b.z[b.ptr[k]:b.ptr[k + 1]].numpy(), r[0]) and np.array_equal(b.hamiltonian[k], r[4])
