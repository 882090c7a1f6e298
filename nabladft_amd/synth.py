"""Seeded synthetic drug-like conformers (numpy only).

A valence-limited, self-avoiding random tree: heavy atoms at 1.45 +- 0.06 A from a parent (<= 3 heavy
bonds per atom, non-bonded heavy atoms >= 2.1 A apart, 50 % chain growth from the last three atoms),
then hydrogens at 1.09 A (<= 4 bonds per heavy atom, >= 1.65 A from everything but the parent).
Element frequencies follow the reference's test database (tests/data/raw/test_database.db:
H .470, C .384, N .068, O .059, S .009, F .007, Cl .003).  Measured statistics of the generator
(128 conformers): mean degree at 5 A = 18.8 (real 19.6), diameter 9.2-18.0 A, mean 13.0 (real 8.9-17.5,
13.7), minimum distance 1.09 A (real 0.97), 790 directed edges per conformer (real 833).
Used by bench.py (workload) and by the oracle/tests (same inputs on both sides)."""
import numpy as np
import torch

_ELEMENTS = np.array([1, 6, 7, 8, 16, 9, 17])
_EL_P = np.array([0.470, 0.384, 0.068, 0.059, 0.009, 0.007, 0.003])
_P_CHAIN, _SEP_HEAVY, _SEP_H = 0.5, 2.1, 1.65


def _one_molecule(rng, n):
    n_heavy = max(1, int(round(n * 0.53)))
    pts, val = [np.zeros(3)], [0]
    tries = 0
    while len(pts) < n_heavy:
        tries += 1
        if rng.random() < _P_CHAIN:
            b = len(pts) - 1 - int(rng.integers(0, min(3, len(pts))))
        else:
            b = int(rng.integers(len(pts)))
        if val[b] >= 3 and tries < 5000:
            continue
        v = rng.normal(size=3)
        cand = pts[b] + v / np.linalg.norm(v) * rng.normal(1.45, 0.06)
        d = np.linalg.norm(np.array(pts) - cand, axis=1)
        d[b] = 9.0
        if d.min() >= (_SEP_HEAVY if tries < 5000 else 1.2):
            pts.append(cand)
            val.append(1)
            val[b] += 1
    heavy = len(pts)
    tries = 0
    while len(pts) < n:
        tries += 1
        b = int(rng.integers(heavy))
        if val[b] >= 4 and tries < 5000:
            continue
        v = rng.normal(size=3)
        cand = pts[b] + v / np.linalg.norm(v) * 1.09
        d = np.linalg.norm(np.array(pts) - cand, axis=1)
        d[b] = 9.0
        if d.min() >= (_SEP_H if tries < 5000 else 0.95):
            pts.append(cand)
            val[b] += 1
    return np.array(pts), heavy


def gen_conformers(seed: int, n_mol: int, size="drug", dtype=torch.float32):
    """Returns (pos [N,3], z int64 [N], batch int64 [N], y [n_mol], forces [N,3]).
    size: 'drug' -> n ~ clip(round(N(42,5)), 29, 54); (lo, hi) -> U{lo..hi}; int -> fixed.
    Targets: y ~ N(0,1), forces ~ N(0, 0.05)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pos_all, z_all, batch_all = [], [], []
    for m in range(n_mol):
        if size == "drug":
            n = int(np.clip(np.rint(rng.normal(42, 5)), 29, 54))
        elif isinstance(size, tuple):
            n = int(rng.integers(size[0], size[1] + 1))
        else:
            n = int(size)
        pts, heavy = _one_molecule(rng, n)
        zs = rng.choice(_ELEMENTS[1:], size=heavy, p=_EL_P[1:] / _EL_P[1:].sum())
        z = np.concatenate([zs, np.ones(n - heavy, dtype=zs.dtype)])
        perm = rng.permutation(n)
        pos_all.append(pts[perm])
        z_all.append(z[perm])
        batch_all.append(np.full(n, m))
    pos = torch.tensor(np.concatenate(pos_all).astype(np.float32)).to(dtype)
    z = torch.tensor(np.concatenate(z_all).astype(np.int64))
    batch = torch.tensor(np.concatenate(batch_all).astype(np.int64))
    y = torch.tensor(rng.normal(0, 1, size=n_mol).astype(np.float32)).to(dtype)
    f = torch.tensor(rng.normal(0, 0.05, size=(pos.shape[0], 3)).astype(np.float32)).to(dtype)
    return pos, z, batch, y, f


def take_conformers(pos, z, batch, y, f, idx):
    """The conformers `idx` (ascending list) of a batch, renumbered 0..len(idx)-1."""
    idx_t = torch.as_tensor(list(idx), dtype=torch.long)
    n_mol = int(y.shape[0])
    new_id = torch.full((n_mol,), -1, dtype=torch.long)
    new_id[idx_t] = torch.arange(idx_t.shape[0])
    keep = new_id[batch] >= 0
    return pos[keep], z[keep], new_id[batch[keep]], y[idx_t], f[keep]


def gen_rank_conformers(seed: int, per_rank: int, world: int = 1, rank: int = 0, size="drug", cost_model: str = "n2"):
    """What one rank of a data-parallel job trains on: every rank draws the SAME global batch of per_rank * world conformers (same seed) and keeps its
    share of the cost-balanced partition (dist.shard_by_cost with the model's proxy) -- one molecule = one graph, nothing else is exchanged.
    Returns ((pos, z, batch, y, forces), predicted spread of the per-rank cost)."""
    from .dist import predicted_spread, shard_by_cost
    data = gen_conformers(seed, per_rank * world, size)
    if world == 1:
        return data, 0.0
    sizes = torch.bincount(data[2], minlength=per_rank * world).tolist()
    parts = shard_by_cost(sizes, world, cost_model)
    return take_conformers(*data, parts[rank]), predicted_spread(sizes, world, cost_model)
