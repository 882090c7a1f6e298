"""Where does the eSCN smoke configuration lose its force accuracy?  HIP model vs the oracle in float64 and in float32, stage by stage."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nabladft_amd.escn import eSCN, _RowFn, _MatmulFn
from nabladft_amd.synth import gen_conformers
from oracle import escn_ref as R
from oracle.escn_params import make_state
dev = torch.device("cuda:0")
cfg = dict(num_targets=1, use_pbc=False, regress_forces=True, otf_graph=True, use_grid=True, distance_function="gaussian", basis_width_scalar=1.0,
           show_timing_info=False, max_neighbors=6, cutoff=4.5, max_num_elements=20, num_layers=2, lmax_list=[4], mmax_list=[2], sphere_channels=16,
           hidden_channels=32, edge_channels=16, num_sphere_samples=32, distance_resolution=0.25)
rng = np.random.Generator(np.random.PCG64(13))
sizes = [7, 10]
pos = torch.cat([gen_conformers(13 + i, 1, size=n)[0] for i, n in enumerate(sizes)])
z = torch.tensor(rng.choice([1, 6, 7, 8], size=sum(sizes)))
net = eSCN(**cfg)
train = [(k, tuple(p.shape)) for k, p in net.named_parameters() if p.requires_grad]
net.load_state_dict(make_state(train, 3), strict=False)
P = {k: v.detach().double() for k, v in net.state_dict().items()}
if "--own-constants" not in sys.argv:
    P["sphere_points"], P["sphharm_weights.0"] = R.sphere_constants(cfg, torch.float64)      # round 2's choice: constants rebuilt in float64 for the oracle
P["distance_expansion.offset"] = torch.linspace(0.0, cfg["cutoff"], int(cfg["cutoff"] / cfg["distance_resolution"]), dtype=torch.float64)
net.to(dev)
class B: pass
b = B()
b.pos, b.z, b.batch = pos.to(dev), z.to(dev), torch.repeat_interleave(torch.arange(2), torch.tensor(sizes)).to(dev)
with torch.no_grad():
    G = net.build_graph(b)
    E, F, layers, _ = net(b, edge_rot_mat=G.rot, return_layers=True)
    x = layers[0]
    for out in layers[1:]:
        x = x + out
    nf, Cc, Pn = 25, 16, 32
    x_pt = _RowFn.apply(x, net.sphharm_weights[0], 0, Pn, nf, Cc, False, None, None, G.N).view(-1, Cc)
    f = net.force_block(x_pt).view(G.N, Pn)
    p64, p32 = {}, {}
    E64, F64 = R.forward(P, cfg, pos.double(), z, sizes, rot=G.rot.cpu().double(), probe=p64)
    P32 = {k: v.float() for k, v in P.items()}
    E32, F32 = R.forward(P32, cfg, pos.float(), z, sizes, rot=G.rot.cpu().float(), probe=p32)
rel = lambda a, r: float((a.detach().cpu().double().reshape(-1) - r.double().reshape(-1)).abs().max() / r.double().abs().max())
print("final embedding x : hip", rel(x, p64["x"]), " oracle32", rel(p32["x"], p64["x"]))
print("sphere features   : hip", rel(x_pt, p64["x_pt"]), " oracle32", rel(p32["x_pt"], p64["x_pt"]))
print("point forces f    : hip", rel(f, p64["f"]), " oracle32", rel(p32["f"], p64["f"]), " |f|max", float(p64["f"].abs().max()), " |F|max", float(F64.abs().max()))
print("forces            : hip", rel(F, F64), " oracle32", rel(F32, F64))
# the last contraction alone, from the float64 f: float32 GEMM vs float32 torch
f64 = p64["f"]
sp = P["sphere_points"]
Fg = _MatmulFn.apply(f64.float().to(dev), (sp.float() / Pn).contiguous().to(dev))
Ft = (f64.float().unsqueeze(-1) * sp.float().view(1, Pn, 3)).sum(1) / Pn
print("last contraction from exact f: hip gemm", rel(Fg, F64), " torch32", rel(Ft, F64))
for i, lay in enumerate(layers):
    pass
