"""Stage-by-stage errors of EquiformerV2 on the GPU against the small fixture (development aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_equiformer_gpu import SMALL, Data, _loss, build, rel  # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "equiformer_small.npz"))
dev = torch.device("cuda:0")
net = build(SMALL, d, dev)
data = Data(d, dev)
E, F, rec, G = net(data, edge_rot_mat=torch.tensor(d["edge_rot_mat"]), return_intermediates=True)
print("graph equal", np.array_equal(np.stack([G.src.cpu().numpy(), G.dst.cpu().numpy()]), d["edge_index"]))
for k in ("embed", "norm1", "ga", "block0", "block1"):
    print(k, rel(rec[k].detach().cpu().numpy(), d["f64:" + k].reshape(G.N, -1)), "own f32", rel(d["f32:" + k], d["f64:" + k]))
print("E", rel(E.detach().cpu().numpy(), d["f64:E"]), "F", rel(F.detach().cpu().numpy(), d["f64:F"]))
loss = _loss(E, F, data)
loss.backward()
print("loss", float(loss), float(d["f64:loss"]))
rows = []
for name, p in net.named_parameters():
    ref64 = d["f64:grad:" + name]
    g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref64)
    scale = max(np.abs(ref64).max(), 1e-30)
    rows.append((np.abs(g - ref64).max() / scale, np.abs(d["f32:grad:" + name] - ref64).max() / scale, name))
rows.sort(reverse=True)
for r in rows[:25]:
    print("grad %.2e (f32 ref %.2e) %s" % r)
print("params with err > 5e-5:", sum(1 for r in rows if r[0] > max(5e-5, 3 * r[1])), "of", len(rows))
