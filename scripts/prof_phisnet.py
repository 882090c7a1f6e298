"""Development: cProfile of the PhiSNet training step (host-side overhead), model construction excluded."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from scripts.bench_phisnet import HP, SHELLS, synthetic_batch  # noqa: E402
from nabladft_amd.phisnet import NeuralNetwork  # noqa: E402

max_orbitals = tuple(tuple((zz, l) for l in SHELLS[zz]) for zz in (1, 1, 6, 6, 7, 7, 8, 8))
torch.manual_seed(0)
m = NeuralNetwork(max_orbitals=max_orbitals, **HP).cuda()
b = synthetic_batch(2, 42)
batch = dict(positions=torch.tensor(b["positions"]).view(1, -1, 3).cuda(), atomic_numbers=torch.tensor(b["z"]).cuda(), orbitals=b["orbitals"],
             molecule_size=torch.tensor(b["sizes"]))
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-3, amsgrad=True)


def step():
    opt.zero_grad(set_to_none=True)
    out = m(batch)
    loss = out["full_hamiltonian_packed"].abs().mean() + out["overlap_matrix_packed"].abs().mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print(s.getvalue()[:4500])
