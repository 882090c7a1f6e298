"""TEST INFRASTRUCTURE (container-only): golden vectors of the QHNet network, produced by the REAL reference classes
(nablaDFT/qhnet/qhnet.py: QHNet; qhnet/layers.py: ConvNetLayer, SelfNetLayer, PairNetLayer, Expansion, NormGate, InnerProduct;
qhnet/loss.py: HamiltonianLoss) imported through oracle/qhnet_ref_import.py ON TOP OF oracle/e3nn_mini.py.

Pinned by these fixtures: every QHNet-specific line.  NOT pinned: e3nn 0.5.1's own arithmetic (restated in e3nn_mini.py, parity unpinned).

  tests/golden/qhnet_small.npz   hidden 32 / bottleneck 16 / 16 radial functions / 4 layers / cutoff 3.0: 3 molecules (incl. a single atom),
                                 every intermediate (per-layer node features, fii, fij, padded blocks), H in fp32 and fp64, loss, all gradients
  tests/golden/qhnet_full.npz    config/model/qhnet.yaml sizes (hidden 128 / bottleneck 32 / 32 radial / 5 layers / cutoff 12): 2 molecules;
                                 H in fp32 and fp64, loss, per-tensor gradient summaries (norm and projection on a fixed direction)
Run:  python oracle/make_golden_qhnet_model.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.qhnet_params import make_state, probe_direction  # noqa: E402
from oracle.qhnet_ref_import import Data, load_qhnet_full  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ORBITALS = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 7: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2], 9: [0, 0, 0, 1, 1, 2],
            16: [0, 0, 0, 0, 1, 1, 1, 2], 17: [0, 0, 0, 0, 1, 1, 1, 2], 35: [0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]}   # config/model/qhnet.yaml:14-22


def molecules(rng, sizes, spread):
    pos, z = [], []
    for n in sizes:
        p = [np.zeros(3)]
        while len(p) < n:                                          # random tree, bond length 1.1-1.6, no two atoms closer than 0.9
            c = p[rng.integers(len(p))] + rng.normal(size=3) / np.sqrt(3) * spread
            if min(np.linalg.norm(c - q) for q in p) > 0.9:
                p.append(c)
        pos.append(np.array(p))
        z.append(rng.choice([1, 1, 1, 6, 6, 7, 8, 9, 16, 17, 35], size=n))
    return np.concatenate(pos).astype(np.float32), np.concatenate(z)


def run(ref, cfg, pos, z, sizes, seed, dtype):
    QHNet = ref["qhnet"].QHNet
    torch.manual_seed(0)
    net = QHNet(**cfg, orbitals=ORBITALS)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters()]
    state = make_state(names, seed)
    missing = net.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys
    assert all(("output_mask" in k) or k.endswith("tp.weight") or k.endswith("tp_node.weight") or k.endswith("tp_node_pair.weight")
               or k.endswith("mul.weight") or k.endswith(".bias") or "distance_expansion" in k for k in missing.missing_keys), missing.missing_keys
    net = net.to(dtype)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    data = Data(torch.tensor(pos, dtype=dtype), torch.tensor(z, dtype=torch.long), batch, ptr)
    inter = {}

    def hook(name):
        def fn(mod, inp, out):
            inter[name] = out.detach().clone()
        return fn

    for i, m in enumerate(net.e3_gnn_layer):
        m.register_forward_hook(hook(f"conv{i}"))
    for i, m in enumerate(net.e3_gnn_node_layer):
        m.register_forward_hook(hook(f"self{i}"))
    for i, m in enumerate(net.e3_gnn_node_pair_layer):
        m.register_forward_hook(hook(f"pair{i}"))
    net.expand_ii["hamiltonian"].register_forward_hook(hook("diag_blocks"))
    net.expand_ij["hamiltonian"].register_forward_hook(hook("nondiag_blocks"))
    H = net(data)
    norb = [len(net.orbital_mask[int(a)]) for a in z]
    mol_orb = [sum(norb[int(ptr[b]):int(ptr[b + 1])]) for b in range(len(sizes))]
    trng = np.random.Generator(np.random.PCG64(seed + 100))
    tblocks = []
    for m in mol_orb:
        t = trng.normal(size=(m, m)) * 0.05
        tblocks.append(torch.tensor(t + t.T, dtype=dtype))
    target = torch.block_diag(*tblocks)
    mask = torch.block_diag(*[torch.ones_like(t) for t in tblocks])
    loss = ref["loss"].HamiltonianLoss()(H, target, mask)
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    sd = net.state_dict()

    def instr(tp):
        return np.array([[tp.irreps_in1[i.i_in1].ir.l, tp.irreps_in2[i.i_in2].ir.l, tp.irreps_out[i.i_out].ir.l, i.path_weight] for i in tp.instructions])
    extra = dict(instr_conv0=instr(net.e3_gnn_layer[0].conv.tp_node), instr_conv1=instr(net.e3_gnn_layer[1].conv.tp_node),
                 instr_pair=instr(net.e3_gnn_node_pair_layer[0].tp_node_pair), instr_self=instr(net.e3_gnn_node_layer[0].tp), state_keys=np.array(list(sd.keys())), state_shapes=np.array([",".join(str(d) for d in v.shape) for v in sd.values()]), edge_index=data.edge_index, full_edge_index=data.full_edge_index, edge_attr=data.edge_attr.detach(), edge_sh=data.edge_sh.detach(),
                 node_attr=data.node_attr.detach())
    return net, names, H.detach(), target, loss.detach(), grads, inter, extra, ptr


def main():
    ref = load_qhnet_full()
    # ---- small: everything stored ------------------------------------------------------------------------------------------------------
    cfg = dict(in_node_features=1, sh_lmax=4, hidden_size=32, bottle_hidden_size=16, num_gnn_layers=4, max_radius=3.0, num_nodes=40, radius_embed_dim=16)
    rng = np.random.Generator(np.random.PCG64(5))
    sizes = [6, 1, 4]
    pos, z = molecules(rng, sizes, 1.4)
    z[0], z[7] = 35, 1
    net, names, H32, target, loss32, g32, inter32, extra, ptr = run(ref, cfg, pos, z, sizes, 3, torch.float32)
    _, _, H64, _, loss64, g64, inter64, _, _ = run(ref, cfg, pos, z, sizes, 3, torch.float64)
    out = dict(pos=pos, z=z, sizes=np.array(sizes), seed=np.int64(3), cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()]),
               H32=H32.numpy(), H64=H64.numpy(), target=target.numpy(), loss32=np.float64(loss32), loss64=np.float64(loss64),
               edge_index=extra["edge_index"].numpy(), full_edge_index=extra["full_edge_index"].numpy(), edge_attr=extra["edge_attr"].numpy(),
               edge_sh=extra["edge_sh"].numpy(), node_attr=extra["node_attr"].numpy(), param_names=np.array([n for n, _ in names]),
               state_keys=extra["state_keys"], state_shapes=extra["state_shapes"], instr_conv0=extra["instr_conv0"], instr_conv1=extra["instr_conv1"],
               instr_pair=extra["instr_pair"], instr_self=extra["instr_self"])
    for k, v in inter64.items():
        out["inter64_" + k] = v.numpy().astype(np.float32)
    for k, v in g64.items():
        if v is not None:
            out["grad64_" + k] = v.numpy().astype(np.float32)
    out["grad32_relerr"] = np.array([float((g32[k].double() - g64[k]).abs().max() / g64[k].abs().max().clamp_min(1e-300)) for k, _ in names if g64[k] is not None])
    out["unused_params"] = np.array([k for k, _ in names if g64[k] is None])
    np.savez_compressed(os.path.join(OUT, "qhnet_small.npz"), **out)
    print("qhnet_small.npz: N", len(z), "E", extra["edge_index"].shape[1], "P", extra["full_edge_index"].shape[1], "H", tuple(H32.shape),
          "loss", float(loss32), float(loss64), "max|H32-H64|/max|H|", float((H32.double() - H64).abs().max() / H64.abs().max()),
          "params", sum(int(np.prod(s)) for _, s in names), "grad32 vs 64 worst", out["grad32_relerr"].max())

    # ---- full configuration: summaries ---------------------------------------------------------------------------------------------------
    cfg = dict(in_node_features=1, sh_lmax=4, hidden_size=128, bottle_hidden_size=32, num_gnn_layers=5, max_radius=12, num_nodes=83, radius_embed_dim=32)
    rng = np.random.Generator(np.random.PCG64(8))
    sizes = [7, 5]
    pos, z = molecules(rng, sizes, 5.5)               # bohr-like distances; the 12-bohr cutoff excludes some pairs
    z[0], z[3] = 35, 16
    net, names, H32, target, loss32, g32, _, extra, ptr = run(ref, cfg, pos, z, sizes, 4, torch.float32)
    _, _, H64, _, loss64, g64, inter64, _, _ = run(ref, cfg, pos, z, sizes, 4, torch.float64)
    gnames = [k for k, _ in names if g64[k] is not None]
    out = dict(pos=pos, z=z, sizes=np.array(sizes), seed=np.int64(4), cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()]),
               H32=H32.numpy(), H64=H64.numpy(), target=target.numpy(), loss32=np.float64(loss32), loss64=np.float64(loss64),
               edge_index=extra["edge_index"].numpy(), full_edge_index=extra["full_edge_index"].numpy(),
               grad_names=np.array(gnames), grad64_norm=np.array([float(g64[k].norm()) for k in gnames]),
               grad64_probe=np.array([float((g64[k] * probe_direction(k, g64[k].shape, 4)).sum()) for k in gnames]),
               grad32_probe=np.array([float((g32[k].double() * probe_direction(k, g64[k].shape, 4)).sum()) for k in gnames]),
               unused_params=np.array([k for k, _ in names if g64[k] is None]), state_keys=extra["state_keys"], state_shapes=extra["state_shapes"],
               diag_blocks=inter64["diag_blocks"].numpy().astype(np.float32), nondiag_blocks=inter64["nondiag_blocks"].numpy().astype(np.float32),
               conv4=inter64["conv4"].numpy().astype(np.float32), self1=inter64["self1"].numpy().astype(np.float32),
               pair1=inter64["pair1"].numpy().astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "qhnet_full.npz"), **out)
    print("qhnet_full.npz: N", len(z), "E", extra["edge_index"].shape[1], "P", extra["full_edge_index"].shape[1], "H", tuple(H32.shape),
          "loss", float(loss32), float(loss64), "max|H32-H64|/max|H|", float((H32.double() - H64).abs().max() / H64.abs().max()),
          "params", sum(int(np.prod(s)) for _, s in names))


if __name__ == "__main__":
    main()
