"""Development aid: nabladft_amd.escn pieces against plain torch on the GPU."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_escn_gpu as T
from nabladft_amd import escn as ES
d = np.load(os.path.join(ROOT, "tests/golden/escn_small.npz"))
dev = torch.device("cuda:0")
net = T.build(T.SMALL, d, dev)
data = T.Data(d, dev)
with torch.no_grad():
    G = net.build_graph(data, torch.tensor(d["edge_rot_mat"]))
    K = net._constants(dev)
    o = K.order; C = K.C
    x = torch.randn(G.N, o.n_full * C, device=dev)
    xs = ES._RowFn.apply(x, G.wigner, o.n_red * o.n_full, o.n_red, o.n_full, C, False, G.src, G.src_inverse, G.E)
    W = G.wigner.view(G.E, o.n_red, o.n_full)
    ref = torch.bmm(W, x.view(G.N, o.n_full, C)[G.src.long()]).reshape(G.E, -1)
    print("rotate", (xs - ref).abs().max().item())
    blk = net.layer_blocks[1].message_block.so2_block_source
    xe = torch.randn(G.E, T.SMALL["edge_channels"], device=dev)
    y = blk(xs, xe, o)
    # torch restatement of SO2Block on the m-primary tensor
    act = torch.nn.functional.silu
    E = G.E
    emb = xs.view(E, o.n_red, C).clone()
    n0 = o.m_size[0]
    x0 = emb[:, :n0].reshape(E, -1)
    g0 = act(xe @ blk.fc1_dist0.weight.T + blk.fc1_dist0.bias)
    out = [((x0 @ blk.fc1_m0.weight.T) * g0) @ blk.fc2_m0.weight.T]
    off = n0
    for m, conv in enumerate(blk.so2_conv, start=1):
        nm = o.m_size[m]
        xm = emb[:, off:off + 2 * nm].reshape(E, 2, -1)
        g = act(xe @ conv.fc1_dist.weight.T + conv.fc1_dist.bias).view(E, 2, -1)
        xr = ((xm @ conv.fc1_r.weight.T) * g[:, 0:1]) @ conv.fc2_r.weight.T
        xi = ((xm @ conv.fc1_i.weight.T) * g[:, 1:2]) @ conv.fc2_i.weight.T
        out += [xr[:, 0] - xi[:, 1], xr[:, 1] + xi[:, 0]]
        off += 2 * nm
    ref = torch.cat(out, dim=1)
    print("so2block", (y - ref).abs().max().item(), ref.abs().max().item())
    # node-level grids
    Tf, Ff = K.to_grid_full, K.from_grid_full
    gx = ES._RowFn.apply(x, Tf, 0, Tf.shape[0], o.n_full, C, False, None, None, G.N)
    ref = torch.einsum("gi,nic->ngc", Tf, x.view(G.N, o.n_full, C)).reshape(G.N, -1)
    print("to_grid", (gx - ref).abs().max().item())
    back = ES._RowFn.apply(gx, Ff, 0, Ff.shape[0], o.n_full, C, True, None, None, G.N)
    print("from(to(x)) - x", (back - x).abs().max().item())
    E_, F_, layers, _ = net(data, edge_rot_mat=torch.tensor(d["edge_rot_mat"]), return_layers=True)
    for i, xl in enumerate(layers):
        ref64 = d[f"f64:layer{i}"].reshape(G.N, o.n_full, C)
        got = xl.cpu().numpy().reshape(G.N, o.n_full, C)
        per_l = [np.abs(got[:, l * l:(l + 1) ** 2] - ref64[:, l * l:(l + 1) ** 2]).max() / np.abs(ref64).max() for l in range(o.lmax + 1)]
        print("layer", i, ["%.2e" % v for v in per_l])
