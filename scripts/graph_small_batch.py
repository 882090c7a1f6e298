"""What a HIP-graph replay of the PaiNN training step buys at the reference's default batch size (config/painn-oc.yaml:11, 32 conformers): the step with the
neighbour list of ONE fixed batch (the list build synchronises to size its arrays, so it stays outside the capture), eager vs captured."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nabladft_amd as nq
from nabladft_amd import painn as P, trainer as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
CUTOFF, MAX_NEIGHBORS = 5.0, 100                                            # config/model/painn-oc.yaml
model = nq.PaiNN(128, 6, 100, CUTOFF, MAX_NEIGHBORS, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100).to(dev)
from nabladft_amd.synth import gen_conformers
pos, z, batch, y, f = gen_conformers(7, B)
b = nq.Batch(pos, z, batch, y, f).to(dev)
step = nq.FusedTrainStep(model, lr=5e-4, max_grad_norm=5.0)
for _ in range(5):
    step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step(b)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 50
nl = P.build_neighbor_list(b.pos, b.batch, b.z, CUTOFF, MAX_NEIGHBORS, b.ptr)
orig = T.build_neighbor_list
T.build_neighbor_list = lambda *a, **k: nl
try:
    for _ in range(3):
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(b)
    torch.cuda.synchronize()
    fixed = (time.perf_counter() - t0) / 50
    g = T.GraphedStep(lambda: step(b), warmup=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 200
finally:
    T.build_neighbor_list = orig
print(json.dumps({"conformers": B, "eager_ms": 1e3 * eager, "eager_fixed_graph_ms": 1e3 * fixed, "hip_graph_replay_ms": 1e3 * graphed}))
