"""TEST INFRASTRUCTURE (container-only): golden vectors of EquiformerV2 produced by the REAL reference classes (nablaDFT/equiformer_v2/*.py) imported through
oracle/equiformer_import.py on top of oracle/e3nn_mini.py (four e3nn symbols restated, PARITY UNPINNED for those; the reference's own Jd.pt is used as is).
The model runs in eval() mode (attention dropout and drop-path are random in training mode; they are identities here).

  tests/golden/equiformer_small.npz  lmax 3 / mmax 2, 2 blocks, 16 channels, 2 heads, cutoff 4.0, max_neighbors 5 (binds), 3 molecules: graph, the edge
                                     rotation matrices the run drew, Wigner matrices, grid matrices, the embedding after every stage, E, F, loss, all
                                     gradients (fp32 and fp64 runs with the same rotation matrices)
  tests/golden/equiformer_full.npz   config/model/equiformer_v2_oc20.yaml: 2 molecules; graph, E, F, loss, gradient norms / projections
Run:  python oracle/make_golden_equiformer.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.equiformer_import import load_equiformer  # noqa: E402
from oracle.equiformer_params import make_state, probe_direction  # noqa: E402
from oracle.make_golden_gemnet import Data, molecules  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
COMMON = dict(use_pbc=False, regress_forces=True, otf_graph=True, norm_type="layer_norm_sh", use_atom_edge_embedding=True, share_atom_edge_embedding=False,
              distance_function="gaussian", num_distance_basis=512, attn_activation="silu", use_s2_act_attn=False, use_attn_renorm=True, ffn_activation="silu",
              use_gate_act=False, use_grid_mlp=True, use_sep_s2_act=True, alpha_drop=0.1, drop_path_rate=0.05, proj_drop=0.0, weight_init="uniform")
SMALL = dict(COMMON, max_neighbors=5, max_radius=4.0, max_num_elements=40, num_layers=2, sphere_channels=16, attn_hidden_channels=8, num_heads=2,
             attn_alpha_channels=8, attn_value_channels=4, ffn_hidden_channels=16, lmax_list=[3], mmax_list=[2], num_sphere_samples=32, edge_channels=16)
FULL = dict(COMMON, max_neighbors=30, max_radius=12.0, max_num_elements=65, num_layers=12, sphere_channels=128, attn_hidden_channels=64, num_heads=8,
            attn_alpha_channels=64, attn_value_channels=16, ffn_hidden_channels=128, lmax_list=[6], mmax_list=[2], num_sphere_samples=128,
            edge_channels=128)                                                              # config/model/equiformer_v2_oc20.yaml:5-41
LOSS_COEFS = (2.0, 100.0)                                                                   # :62-64


def run(ref, cfg, pos, z, sizes, seed, dtype, rot=None, record=True):
    import logging
    logging.disable(logging.WARNING)
    torch.set_default_dtype(dtype)
    torch.manual_seed(0)
    net = ref["model"].EquiformerV2_OC20(**cfg)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters() if v.requires_grad]
    missing = net.load_state_dict(make_state(names, seed), strict=False)
    assert not missing.unexpected_keys
    net = net.to(dtype).eval()
    rec = {}
    orig = net._init_edge_rot_mat

    def rot_hook(data, edge_index, vec):
        r = orig(data, edge_index, vec) if rot is None else torch.tensor(rot, dtype=dtype)
        rec["edge_rot_mat"], rec["edge_index"], rec["edge_vec"] = r.detach().clone(), edge_index.clone(), vec.detach().clone()
        return r

    net._init_edge_rot_mat = rot_hook
    if record:
        net.blocks[0].register_forward_pre_hook(lambda m, inp: rec.__setitem__("embed", inp[0].embedding.detach().clone()))
        net.blocks[0].norm_1.register_forward_hook(lambda m, inp, out: rec.__setitem__("norm1", out.detach().clone()))
        net.blocks[0].ga.register_forward_hook(lambda m, inp, out: rec.__setitem__("ga", out.embedding.detach().clone()))
        for i, blk in enumerate(net.blocks):
            blk.register_forward_hook(lambda m, inp, out, i=i: rec.__setitem__(f"block{i}", out.embedding.detach().clone()))
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    data = Data(torch.tensor(pos, dtype=dtype), torch.tensor(z, dtype=torch.long), batch)
    torch.manual_seed(seed)
    E, F = net(data)
    trng = np.random.Generator(np.random.PCG64(seed + 100))
    y = torch.tensor(trng.normal(size=len(sizes)) * 0.1, dtype=dtype)
    ft = torch.tensor(trng.normal(size=(len(z), 3)) * 0.05, dtype=dtype)
    loss = LOSS_COEFS[0] * torch.nn.functional.l1_loss(E, y) + LOSS_COEFS[1] * torch.linalg.vector_norm(F - ft, dim=-1).mean()
    loss.backward()
    torch.set_default_dtype(torch.float32)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    return net, rec, E.detach(), F.detach(), y, ft, loss.detach(), grads, names


def npy(v):
    return v.detach().cpu().numpy()


def main():
    ref = load_equiformer()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(31))
    sizes = [9, 4, 11]
    pos, z = molecules(rng, sizes, 1.5)
    z = np.minimum(z, 35)
    out = {"pos": pos, "z": z, "sizes": np.array(sizes), "seed": np.array(8)}
    net, rec, E, F, y, ft, loss, grads, names = run(ref, SMALL, pos, z, sizes, 8, torch.float32)
    rot = npy(rec["edge_rot_mat"])
    out.update({"edge_index": npy(rec["edge_index"]), "edge_rot_mat": rot, "y": npy(y), "f_target": npy(ft)})
    out["state_keys"] = np.array(list(net.state_dict().keys()))
    out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])
    out["param_names"] = np.array([n for n, _ in names])
    pn = set(out["param_names"].tolist())
    for k, v in net.state_dict().items():
        if k in pn or k.count(".") <= 3 and not k.startswith("blocks.1"):                  # parameters + the top-level buffers (the nested copies are the same objects)
            out[f"state:{k}"] = npy(v)
    out["wigner"] = npy(net.SO3_rotation[0].wigner)
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        if tag == "f64":
            net, rec, E, F, y, ft, loss, grads, names = run(ref, SMALL, pos, z, sizes, 8, dtype, rot=rot)
        for k in ["embed", "norm1", "ga"] + [f"block{i}" for i in range(SMALL["num_layers"])]:
            out[f"{tag}:{k}"] = npy(rec[k])
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = npy(E), npy(F), npy(loss)
        for k, gr in grads.items():
            out[f"{tag}:grad:{k}"] = npy(gr)
    print("small: E", out["f32:E"], "loss", float(out["f32:loss"]), "edges", out["edge_index"].shape, "f32-f64 E", np.abs(out["f32:E"] - out["f64:E"]).max(),
          "F", np.abs(out["f32:F"] - out["f64:F"]).max() / np.abs(out["f64:F"]).max())
    np.savez_compressed(os.path.join(OUT, "equiformer_small.npz"), **out)
    # ---- yaml configuration
    rng = np.random.Generator(np.random.PCG64(32))
    sizes = [20, 46]
    pos, z = molecules(rng, sizes, 1.5)
    z = np.minimum(z, 35)
    out = {"pos": pos, "z": z, "sizes": np.array(sizes), "seed": np.array(9)}
    rot = None
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        net, rec, E, F, y, ft, loss, grads, names = run(ref, FULL, pos, z, sizes, 9, dtype, rot=rot, record=(tag == "f32"))
        if tag == "f32":
            rot = npy(rec["edge_rot_mat"])
            out.update({"edge_index": npy(rec["edge_index"]), "edge_rot_mat": rot, "y": npy(y), "f_target": npy(ft)})
            out["param_names"] = np.array([n for n, _ in names])
            out["state_keys"] = np.array(list(net.state_dict().keys()))
            out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])
            out["f32:embed"], out["f32:block0"], out["f32:block11"] = (npy(rec[k])[::5, :, ::8] for k in ("embed", "block0", "block11"))
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = npy(E), npy(F), npy(loss)
        out[f"{tag}:grad_norm"] = np.array([float(g.double().norm()) for g in grads.values()])
        out[f"{tag}:grad_probe"] = np.array([float((g.double() * probe_direction(k, g.shape, 9)).sum()) for k, g in grads.items()])
    print("full: params", net.num_params, "E", out["f32:E"], "loss", float(out["f32:loss"]), "edges", out["edge_index"].shape,
          "f32-f64 F", np.abs(out["f32:F"] - out["f64:F"]).max() / np.abs(out["f64:F"]).max())
    np.savez_compressed(os.path.join(OUT, "equiformer_full.npz"), **out)
    for f in ("equiformer_small.npz", "equiformer_full.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "kB")


if __name__ == "__main__":
    main()
