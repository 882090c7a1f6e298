"""Which part of the QHNet step cannot be captured into a HIP graph?  Runs one variant per process (a failed capture may take the process down).
Round 5 found the cause with the ``only:<substring>`` variants (tracked tensors parked on the batch kept default-stream AccumulateGrad nodes alive: qhnet.py forward)
and every variant captures since; kept as the regression probe (profiles/r05_qhnet_capture_bisect_and_fix.txt)."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
faulthandler.enable()
import torch  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda:0")
import bench_qhnet as M  # noqa: E402
from nabladft_amd.hamiltonian import HamiltonianLoss  # noqa: E402
net = M.build(dev)
b = M.synthetic_batch(2, 100, dev)
b.prepared = net.prepare(b)
loss_fn = HamiltonianLoss()
with torch.no_grad():
    target = net(b, packed=True).clone()


if what.startswith("only:"):      # backward pruned to the parameters whose name contains the substring: bisects the autograd graph by depth
    keep = what[5:]
    n_on = 0
    for name, p in net.named_parameters():
        p.requires_grad_(keep in name)
        n_on += int(keep in name)
    print("parameters that require grad:", n_on, flush=True)


def fn():
    if what == "fwd_blocks":
        with torch.no_grad():
            return net(b, keep_blocks=True)["hamiltonian_diagonal_blocks"].sum()
    if what == "fwd_packed":
        with torch.no_grad():
            return net(b, packed=True).sum()
    if what == "fwd_loss":
        with torch.no_grad():
            return loss_fn(net(b, packed=True), target)
    if what == "bwd_sum":
        for p in net.parameters():
            p.grad = None
        out = net(b, packed=True).sum()
        out.backward()
        return out.detach()
    if what == "bwd_blocks":
        for p in net.parameters():
            p.grad = None
        o = net(b, keep_blocks=True)
        out = o["hamiltonian_diagonal_blocks"].sum() + o["hamiltonian_non_diagonal_blocks"].sum()
        out.backward()
        return out.detach()
    if what == "bwd_loss":
        for p in net.parameters():
            p.grad = None
        out = loss_fn(net(b, packed=True), target)
        out.backward()
        return out.detach()
    if what.startswith("only:"):
        for p in net.parameters():
            p.grad = None
        out = net(b, packed=True).sum()
        out.backward()
        return out.detach()
    raise SystemExit("unknown variant")


for _ in range(2):
    fn()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
g.replay()
torch.cuda.synchronize()
print(what, "captured and replayed OK", float(out))
