"""TEST INFRASTRUCTURE (container-only): golden vectors of eSCN produced by the REAL reference classes (nablaDFT/escn/escn.py, so3.py) imported through
oracle/escn_import.py on top of oracle/e3nn_mini.py (five e3nn symbols restated, PARITY UNPINNED for those; the reference's own Jd.pt is used as is).

  tests/golden/escn_small.npz  lmax 3 / mmax 2, 3 layers, channels 16 / 32 / 16, 32 sphere samples, cutoff 4.0, max_neighbors 5 (binds), 3 molecules:
                               graph, the edge rotation matrices the run drew, Wigner matrices, grid matrices, per-layer embeddings, E, F, loss, all gradients
                               (fp32 and fp64 runs with the same rotation matrices), the J matrices of Jd.pt for l <= 6 (data)
  tests/golden/escn_full.npz   config/model/escn-oc.yaml (34.3 M parameters): 2 molecules; graph, E, F, loss, gradient norms / projections
Run:  python oracle/make_golden_escn.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.escn_import import load_escn  # noqa: E402
from oracle.escn_params import make_state, probe_direction  # noqa: E402
from oracle.make_golden_gemnet import Data, molecules  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
COMMON = dict(num_targets=1, use_pbc=False, regress_forces=True, otf_graph=True, use_grid=True, distance_function="gaussian", basis_width_scalar=1.0,
              show_timing_info=False)
SMALL = dict(COMMON, max_neighbors=5, cutoff=4.0, max_num_elements=40, num_layers=3, lmax_list=[3], mmax_list=[2], sphere_channels=16, hidden_channels=32,
             edge_channels=16, num_sphere_samples=32, distance_resolution=0.25)
FULL = dict(COMMON, max_neighbors=40, cutoff=8.0, max_num_elements=65, num_layers=8, lmax_list=[6], mmax_list=[2], sphere_channels=128, hidden_channels=256,
            edge_channels=128, num_sphere_samples=128, distance_resolution=0.02)          # config/model/escn-oc.yaml:5-25


def run(ref, cfg, pos, z, sizes, seed, dtype, rot=None, record=True):
    import logging
    logging.disable(logging.WARNING)
    torch.set_default_dtype(dtype)
    torch.manual_seed(0)
    net = ref["escn"].eSCN(**cfg)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters() if v.requires_grad]
    missing = net.load_state_dict(make_state(names, seed), strict=False)
    assert not missing.unexpected_keys
    net = net.to(dtype)
    rec = {}
    orig = net._init_edge_rot_mat

    def rot_hook(data, edge_index, vec):
        r = orig(data, edge_index, vec) if rot is None else torch.tensor(rot, dtype=dtype)
        rec["edge_rot_mat"], rec["edge_index"], rec["edge_vec"] = r.detach().clone(), edge_index.clone(), vec.detach().clone()
        return r

    net._init_edge_rot_mat = rot_hook
    if record:
        for i, blk in enumerate(net.layer_blocks):
            blk.register_forward_hook(lambda m, inp, out, i=i: rec.__setitem__(f"layer{i}", out.embedding.detach().clone()))
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    data = Data(torch.tensor(pos, dtype=dtype), torch.tensor(z, dtype=torch.long), batch)
    torch.manual_seed(seed)
    E, F = net(data)
    trng = np.random.Generator(np.random.PCG64(seed + 100))
    y = torch.tensor(trng.normal(size=len(sizes)) * 0.01, dtype=dtype)
    ft = torch.tensor(trng.normal(size=(len(z), 3)) * 0.05, dtype=dtype)
    loss = torch.nn.functional.l1_loss(E, y) + 100.0 * torch.linalg.vector_norm(F - ft, dim=-1).mean()       # config/model/escn-oc.yaml:41-48
    loss.backward()
    torch.set_default_dtype(torch.float32)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    return net, rec, E.detach(), F.detach(), y, ft, loss.detach(), grads, names


def npy(v):
    return v.detach().cpu().numpy()


def main():
    ref = load_escn()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(21))
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
    for k, v in net.state_dict().items():
        out[f"state:{k}"] = npy(v)
    out["wigner"] = npy(net.SO3_edge_rot[0].wigner)
    g = net.SO3_grid[3]
    out["to_grid_3_3"], out["from_grid_3_3"] = npy(g[3].get_to_grid_mat("cpu")), npy(g[3].get_from_grid_mat("cpu"))
    out["to_grid_3_2"], out["from_grid_3_2"] = npy(g[2].get_to_grid_mat("cpu")), npy(g[2].get_from_grid_mat("cpu"))
    for l, J in enumerate(ref["so3"]._Jd[:7]):
        out[f"Jd:{l}"] = npy(J)
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        if tag == "f64":
            net, rec, E, F, y, ft, loss, grads, names = run(ref, SMALL, pos, z, sizes, 8, dtype, rot=rot)
        for i in range(SMALL["num_layers"]):
            out[f"{tag}:layer{i}"] = npy(rec[f"layer{i}"])
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = npy(E), npy(F), npy(loss)
        for k, gr in grads.items():
            out[f"{tag}:grad:{k}"] = npy(gr)
    print("small: E", out["f32:E"], "loss", float(out["f32:loss"]), "edges", out["edge_index"].shape, "f32-f64 E", np.abs(out["f32:E"] - out["f64:E"]).max())
    np.savez_compressed(os.path.join(OUT, "escn_small.npz"), **out)
    # ---- yaml configuration
    rng = np.random.Generator(np.random.PCG64(22))
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
            out["f32:layer0"], out["f32:layer7"] = npy(rec["layer0"])[::5, :, ::8], npy(rec["layer7"])[::5, :, ::8]
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = npy(E), npy(F), npy(loss)
        out[f"{tag}:grad_norm"] = np.array([float(g.double().norm()) for g in grads.values()])
        out[f"{tag}:grad_probe"] = np.array([float((g.double() * probe_direction(k, g.shape, 9)).sum()) for k, g in grads.items()])
    print("full: params", net.num_params, "E", out["f32:E"], "loss", float(out["f32:loss"]), "edges", out["edge_index"].shape,
          "f32-f64 F", np.abs(out["f32:F"] - out["f64:F"]).max() / np.abs(out["f64:F"]).max())
    np.savez_compressed(os.path.join(OUT, "escn_full.npz"), **out)
    for f in ("escn_small.npz", "escn_full.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "kB")


if __name__ == "__main__":
    main()
