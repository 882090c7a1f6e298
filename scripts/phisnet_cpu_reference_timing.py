"""CONTAINER-ONLY measurement (needs /root/reference): wall time of one training step (forward + MAE loss on H and S + backward) of the REAL
reference PhiSNet NeuralNetwork on CPU, at the nablaDFT configuration (phisnet/configs/args_nablaDFT_100k_separate.txt: order 4, F = 128,
K = 128, 5 modules, batch of 2 molecules), on the synthetic batch scripts/bench_phisnet.py uses.  The missing pindex_dict.npy is answered
with the inferred table as in oracle/make_golden_phisnet.py.  Prints one JSON line; the number is quoted in DESIGN.md section 5."""
import importlib
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.bench_phisnet import HP, SHELLS, synthetic_batch  # noqa: E402
from nabladft_amd.phisnet import inferred_pair_of_pairs  # noqa: E402


def main():
    atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    pkg = types.ModuleType("ref_phisnet_nn5")
    pkg.__path__ = ["/root/reference/nablaDFT/phisnet/nn"]
    sys.modules["ref_phisnet_nn5"] = pkg
    real_load = np.load

    def load(path, *a, **k):
        if str(path).endswith("pindex_dict.npy"):
            return np.array({n: tuple(t.numpy() for t in inferred_pair_of_pairs(n)) for n in {atoms}}, dtype=object)
        return real_load(path, *a, **k)
    np.load = load
    try:
        nnmod = importlib.import_module("ref_phisnet_nn5.neural_network")
        max_orbitals = tuple(tuple((zz, l) for l in SHELLS[zz]) for zz in (1, 1, 6, 6, 7, 7, 8, 8))
        torch.manual_seed(0)
        m = nnmod.NeuralNetwork(max_orbitals=max_orbitals, **HP).float()
    finally:
        np.load = real_load
    b = synthetic_batch(2, atoms, seed=0)
    batch = dict(positions=torch.tensor(b["positions"]).view(1, -1, 3), atomic_numbers=torch.tensor(b["z"]), orbitals=b["orbitals"],
                 molecule_size=torch.tensor(b["sizes"]))
    times = []
    for it in range(2):
        t0 = time.perf_counter()
        out = m(batch)
        loss = out["full_hamiltonian"].abs().mean() + out["overlap_matrix"].abs().mean()
        loss.backward()
        times.append(time.perf_counter() - t0)
        m.zero_grad()
    print(json.dumps({"what": "real reference PhiSNet train step on CPU (container)", "atoms_per_molecule": atoms, "molecules": 2, "threads": torch.get_num_threads(),
                      "seconds_per_step": min(times), "molecule_steps_per_s": 2 / min(times)}))


if __name__ == "__main__":
    main()
