"""TEST INFRASTRUCTURE (container-only): a small Hamiltonian database written and read back by the REAL reference classes
nablaDFT/dataset/hamiltonian_dataset.py:{HamiltonianDatabase, HamiltonianDataset}.  The reference talks to sqlite through `apsw`, which is not
installed here; the module is loaded with a 20-line adapter that maps the five apsw calls it makes onto the standard library's sqlite3
(same SQL, same file format -- the adapter contains no database logic).  Writes
  tests/golden/hamiltonian_db_6.db     the database file (data: 6 synthetic molecules, H/C/N/O, def2-SVP-like shells)
  tests/golden/hamiltonian_db_6.npz    what the reference classes return for it (rows, orbitals, Z, one collated batch)
    python oracle/make_golden_hamdb.py"""
import importlib.util
import os
import sqlite3
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SRC = "/root/reference/nablaDFT/dataset/hamiltonian_dataset.py"


def _apsw_adapter():
    m = types.ModuleType("apsw")
    m.SQLITE_OPEN_READONLY, m.SQLITE_OPEN_READWRITE, m.SQLITE_OPEN_CREATE = 1, 2, 4

    class Connection:
        def __init__(self, filename, flags=6):
            if flags == m.SQLITE_OPEN_READONLY:
                self._c = sqlite3.connect("file:" + filename + "?mode=ro", uri=True, isolation_level=None)
            else:
                self._c = sqlite3.connect(filename, isolation_level=None)        # autocommit, like apsw

        def cursor(self):
            return self._c.cursor()

        def setbusytimeout(self, ms):
            self._c.execute("PRAGMA busy_timeout=%d" % ms)
    m.Connection = Connection
    return m


def main():
    sys.modules["apsw"] = _apsw_adapter()
    spec = importlib.util.spec_from_file_location("ref_hamiltonian_dataset", SRC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = os.path.join(OUT, "hamiltonian_db_6.db")
    if os.path.exists(path):
        os.remove(path)
    apsw = sys.modules["apsw"]
    db = mod.HamiltonianDatabase(path, flags=apsw.SQLITE_OPEN_READWRITE | apsw.SQLITE_OPEN_CREATE)
    shells = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 7: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2]}
    flags = apsw.SQLITE_OPEN_READWRITE
    for zz, ls in shells.items():
        db.add_orbitals(zz, np.array(ls, dtype=np.int32), flags=flags)
    db.add_Z(np.array(sorted(shells), dtype=np.int32), flags=flags)
    rng = np.random.Generator(np.random.PCG64(77))
    mols = [[8, 1, 1], [6, 1, 1, 1, 1], [7, 1, 1, 1], [6, 8, 1, 1], [1, 1], [6, 6, 1, 1, 1, 1, 8]]
    for i, zs in enumerate(mols):
        n = len(zs)
        norb = sum(2 * l + 1 for a in zs for l in shells[a])
        sym = lambda: (lambda a: (a + a.T) / 2)(rng.normal(size=(norb, norb)))
        db.add_data(np.array(zs, dtype=np.int64), rng.normal(0, 1.5, size=(n, 3)), np.float64(rng.normal(-50, 10)), rng.normal(size=(n, 3)), sym(), sym() * 0.1 + np.eye(norb),
                    sym(), moses_id=100 + i, conformer_id=7 * i, flags=flags)
    del db
    rd = mod.HamiltonianDatabase(path)
    fx = {"len": np.int64(len(rd)), "Z_table": rd.Z}
    for zz in shells:
        fx[f"orbitals_{zz}"] = rd.get_orbitals(zz)
    for i in range(len(mols)):
        for name, v in zip(("Z", "R", "E", "F", "H", "S", "C", "moses_id", "conformer_id"), rd[i]):
            fx[f"row{i}:{name}"] = np.asarray(v)
    some = rd[[1, 4, 5]]
    fx["list_rows"] = np.array([1, 4, 5])
    for j, row in enumerate(some):
        fx[f"list{j}:H"], fx[f"list{j}:moses_id"] = np.asarray(row[4]), np.asarray(row[7])
    ds = mod.HamiltonianDataset(path)
    fx["max_orbitals"] = np.array([[list(t) for t in orbs] + [[-1, -1]] * (8 - len(orbs)) for orbs in ds.max_orbitals])
    batch = ds.collate_fn([0, 1, 3])
    for k, v in batch.items():
        if k == "orbitals":
            fx["batch:orbitals_flat"] = np.array([list(t) for orbs in v for t in orbs])
            fx["batch:orbitals_count"] = np.array([len(orbs) for orbs in v])
        else:
            fx["batch:" + k] = v.numpy()
    tight = mod.HamiltonianDataset(path, max_batch_atoms=8)
    b2 = tight.collate_fn([1, 3, 0], return_filtered=True)
    fx["tight:molecule_size"], fx["tight:filtered"] = b2["molecule_size"].numpy(), np.array(b2["filtered"])
    np.savez_compressed(os.path.join(OUT, "hamiltonian_db_6.npz"), **fx)
    print("hamiltonian_db_6:", os.path.getsize(path), "bytes;", len(fx), "arrays; batch orbitals", batch["full_hamiltonian"].shape)


if __name__ == "__main__":
    main()
