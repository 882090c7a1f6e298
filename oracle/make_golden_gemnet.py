"""TEST INFRASTRUCTURE (container-only): golden vectors of GemNet-OC produced by the REAL reference classes (nablaDFT/gemnet_oc/gemnet_oc.py: GemNetOC;
gemnet_oc/utils.py, interaction_indices.py, layers/*.py) imported through oracle/gemnet_import.py (stand-ins only for the wheels that are not installed:
torch_scatter, torch_cluster.radius_graph, torch_sparse.SparseTensor -- documented semantics, parity unpinned upstream).

  tests/golden/gemnet_small.npz  2 blocks, atom 32 / edge 48 / 24 radial / 5 spherical, FOUR DIFFERENT cutoffs (5.5 / 5.0 / 4.5 / 4.0) and neighbour caps
                                 (1000 / 6 / 4 / 3) that all bind, fitted (non-zero) ScaleFactors, one atom-embedding residual layer: 3 molecules;
                                 all graphs and interaction indices, all bases, per-block features, E, F, loss, all gradients; fp32 and fp64 runs
  tests/golden/gemnet_full.npz   config/model/gemnet-oc.yaml (37.8 M parameters, 12 A cutoffs, caps 1000 / 30 / 20 / 8, ScaleFactors unfitted as with
                                 scale_file: null): 2 molecules of 26 and 38 atoms (the cap of 30 binds); graphs and index counts, E, F, loss, per-tensor gradient
                                 norms and projections on fixed directions (fp32 and fp64)
Run:  python oracle/make_golden_gemnet.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gemnet_import import load_gemnet  # noqa: E402
from oracle.gemnet_params import make_state, probe_direction  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

COMMON = dict(num_targets=1, num_before_skip=2, num_after_skip=2, num_concat=1, num_atom=3, num_output_afteratom=3, num_global_out_layers=2,
              regress_forces=True, direct_forces=True, use_pbc=False, scale_backprop_forces=False, enforce_max_neighbors_strictly=True,
              rbf={"name": "gaussian"}, rbf_spherical=None, envelope={"name": "polynomial", "exponent": 5}, cbf={"name": "spherical_harmonics"},
              sbf={"name": "legendre_outer"}, extensive=True, forces_coupled=True, output_init="HeOrthogonal", activation="silu", scale_file=None,
              quad_interaction=True, atom_edge_interaction=True, edge_atom_interaction=True, atom_interaction=True, scale_basis=True)
SMALL = dict(COMMON, num_spherical=5, num_radial=24, num_blocks=2, emb_size_atom=32, emb_size_edge=48, emb_size_trip_in=16, emb_size_trip_out=16,
             emb_size_quad_in=8, emb_size_quad_out=8, emb_size_aint_in=16, emb_size_aint_out=16, emb_size_rbf=8, emb_size_cbf=8, emb_size_sbf=16,
             num_atom_emb_layers=1, cutoff=5.0, cutoff_qint=4.0, cutoff_aeaint=4.5, cutoff_aint=5.5, max_neighbors=6, max_neighbors_qint=3,
             max_neighbors_aeaint=4, max_neighbors_aint=1000)
FULL = dict(COMMON, num_spherical=7, num_radial=128, num_blocks=4, emb_size_atom=256, emb_size_edge=512, emb_size_trip_in=64, emb_size_trip_out=64,
            emb_size_quad_in=32, emb_size_quad_out=32, emb_size_aint_in=64, emb_size_aint_out=64, emb_size_rbf=16, emb_size_cbf=16, emb_size_sbf=32,
            num_atom_emb_layers=0, cutoff=12.0, cutoff_qint=12.0, cutoff_aeaint=12.0, cutoff_aint=12.0, max_neighbors=30, max_neighbors_qint=8,
            max_neighbors_aeaint=20, max_neighbors_aint=1000)            # config/model/gemnet-oc.yaml:5-60


class Data:
    def __init__(self, pos, z, batch):
        self.pos, self.z, self.batch = pos, z, batch


def molecules(rng, sizes, spread):
    pos, z = [], []
    for n in sizes:
        p = [np.zeros(3)]
        while len(p) < n:
            c = p[rng.integers(len(p))] + rng.normal(size=3) / np.sqrt(3) * spread
            if min(np.linalg.norm(c - q) for q in p) > 0.9:
                p.append(c)
        pos.append(np.array(p) + rng.normal(size=3) * 3.0)
        z.append(rng.choice([1, 1, 1, 6, 6, 7, 8, 9, 16, 17, 35], size=n))
    return np.concatenate(pos).astype(np.float32), np.concatenate(z)


def run(ref, cfg, pos, z, sizes, seed, dtype, fit_scales, full_record):
    import logging
    logging.disable(logging.WARNING)
    torch.manual_seed(0)
    torch.set_default_dtype(dtype)        # utils.py:456 builds its distance table in the default dtype (index_copy_ needs it equal to the distances' dtype)
    net = ref["gemnet"].GemNetOC(**cfg)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters()]
    state = make_state(names, seed, fit_scales)
    missing = net.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys                       # missing: buffers (rbf.offset) and state_dict ALIASES of shared modules (seq_energy_pre = layers, the
    # shared radial_basis_spherical instance) -- named_parameters() lists every shared tensor once, so all of them are set
    net = net.to(dtype)
    net.out_energy.float()                # gemnet_oc.py:1204-1207 casts the inputs of the two final projections with .float(): these stay fp32 in the fp64 run
    net.out_forces.float()
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    data = Data(torch.tensor(pos, dtype=dtype), torch.tensor(z, dtype=torch.long), batch)
    rec = {}

    def keep(name, v):
        rec[name] = v.detach().clone()

    if full_record:
        graphs = net.get_graphs_and_indices(data)
        main, a2a, aea, qint, id_swap, t_e2e, t_a2e, t_e2a, quad = graphs
        for gname, g in (("main", main), ("a2a", a2a), ("a2ee2a", aea), ("qint", qint)):
            for k in ("edge_index", "distance", "vector", "num_neighbors", "target_neighbor_idx"):
                if k in g:
                    keep(f"graph:{gname}:{k}", g[k])
        keep("graph:id_swap", id_swap)
        for tname, t in (("e2e", t_e2e), ("a2e", t_a2e), ("e2a", t_e2a)):
            for k in ("in", "out", "out_agg"):
                keep(f"trip:{tname}:{k}", t[k])
        for k in ("out", "out_agg", "trip_in_to_quad", "trip_out_to_quad"):
            keep(f"quad:{k}", quad[k])
        for k in ("in", "out"):
            keep(f"quad:triplet_in:{k}", quad["triplet_in"][k])
            keep(f"quad:triplet_out:{k}", quad["triplet_out"][k])
        bases = net.get_bases(main_graph=main, a2a_graph=a2a, a2ee2a_graph=aea, qint_graph=qint, trip_idx_e2e=t_e2e, trip_idx_a2e=t_a2e,
                              trip_idx_e2a=t_e2a, quad_idx=quad, num_atoms=len(z))
        rad_raw, b_h, b_out, b_qint, b_e2e, b_a2e, b_e2a, b_a2a = bases
        keep("basis:rad_main_raw", rad_raw)
        keep("basis:atom_update", b_h)
        keep("basis:output", b_out)
        keep("basis:a2a_rad", b_a2a)
        for bname, b in (("qint", b_qint), ("e2e", b_e2e), ("a2e", b_a2e), ("e2a", b_e2a)):
            keep(f"basis:{bname}:rad", b["rad"])
            if isinstance(b["cir"], tuple):
                keep(f"basis:{bname}:cir:rad_W1", b["cir"][0])
                keep(f"basis:{bname}:cir:sph", b["cir"][1])
            else:
                keep(f"basis:{bname}:cir", b["cir"])
            if "sph" in b:
                keep(f"basis:{bname}:sph:rad_W1", b["sph"][0])
                keep(f"basis:{bname}:sph:sph", b["sph"][1])

        def hook(name):
            def fn(mod, inp, out):
                if isinstance(out, tuple):
                    for i, o in enumerate(out):
                        keep(f"{name}:{i}", o)
                else:
                    keep(name, out)
            return fn

        net.atom_emb.register_forward_hook(hook("atom_emb"))
        net.edge_emb.register_forward_hook(hook("edge_emb"))
        for i, b in enumerate(net.int_blocks):
            b.register_forward_hook(hook(f"int{i}"))
            b.trip_interaction.register_forward_hook(hook(f"int{i}:e2e"))
            b.quad_interaction.register_forward_hook(hook(f"int{i}:qint"))
            b.atom_edge_interaction.register_forward_hook(hook(f"int{i}:a2e"))
            b.edge_atom_interaction.register_forward_hook(hook(f"int{i}:e2a"))
            b.atom_interaction.register_forward_hook(hook(f"int{i}:a2a"))
            b.atom_update.register_forward_hook(hook(f"int{i}:atom_update"))
        for i, b in enumerate(net.out_blocks):
            b.register_forward_hook(hook(f"out{i}"))
    E, F = net(data)
    trng = np.random.Generator(np.random.PCG64(seed + 100))
    y = torch.tensor(trng.normal(size=len(sizes)), dtype=dtype)
    ft = torch.tensor(trng.normal(size=(len(z), 3)) * 0.3, dtype=dtype)
    loss = 1.0 * torch.nn.functional.l1_loss(E, y) + 100.0 * ref["loss"].L2Loss()(F, ft)     # config/model/gemnet-oc.yaml:78-85
    loss.backward()
    torch.set_default_dtype(torch.float32)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters() if p.requires_grad}
    return net, rec, E.detach(), F.detach(), y, ft, loss.detach(), grads, names


def to_np(v):
    v = v.detach().cpu()
    return v.numpy()


def main():
    ref = load_gemnet()
    os.makedirs(OUT, exist_ok=True)
    # ---- small ----------------------------------------------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(11))
    sizes = [9, 5, 12]
    pos, z = molecules(rng, sizes, 1.55)
    out = {"pos": pos, "z": z, "sizes": np.array(sizes), "seed": np.array(5)}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net, rec, E, F, y, ft, loss, grads, names = run(ref, SMALL, pos, z, sizes, 5, dtype, True, True)
        for k, v in rec.items():
            if tag == "f32" or k.startswith(("int", "out", "atom_emb", "edge_emb")):
                out[f"{tag}:{k}"] = to_np(v)
            elif v.dtype == torch.long:
                assert np.array_equal(out[f"f32:{k}"], to_np(v)), k      # the fp64 run sees the same graphs
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = to_np(E), to_np(F), to_np(loss)
        for k, g in grads.items():
            out[f"{tag}:grad:{k}"] = to_np(g)
        if tag == "f32":
            out["y"], out["f_target"] = to_np(y), to_np(ft)
            out["state_keys"] = np.array(list(net.state_dict().keys()))
            out["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])
            out["param_names"] = np.array([n for n, _ in names])
            for k, v in net.state_dict().items():
                out[f"state:{k}"] = to_np(v)
            print("small: E", E.numpy(), "loss", float(loss), "edges", {k.split(':')[1]: v.shape[1] for k, v in rec.items() if k.endswith(':edge_index')},
                  "triplets", rec["trip:e2e:in"].numel(), rec["trip:a2e:in"].numel(), rec["trip:e2a:in"].numel(), "quads", rec["quad:out"].numel())
    np.savez_compressed(os.path.join(OUT, "gemnet_small.npz"), **out)
    # ---- full -----------------------------------------------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(12))
    sizes = [26, 38]
    pos, z = molecules(rng, sizes, 1.5)
    out = {"pos": pos, "z": z, "sizes": np.array(sizes), "seed": np.array(6)}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        net, rec, E, F, y, ft, loss, grads, names = run(ref, FULL, pos, z, sizes, 6, dtype, False, tag == "f32")
        out[f"{tag}:E"], out[f"{tag}:F"], out[f"{tag}:loss"] = to_np(E), to_np(F), to_np(loss)
        out[f"{tag}:grad_norm"] = np.array([float(g.double().norm()) for g in grads.values()])
        out[f"{tag}:grad_probe"] = np.array([float((g.double() * probe_direction(k, g.shape, 6)).sum()) for k, g in grads.items()])
        if tag == "f32":
            out["y"], out["f_target"] = to_np(y), to_np(ft)
            out["param_names"] = np.array([n for n, _ in names])
            out["state_keys"] = np.array(list(net.state_dict().keys()))
            for k, v in rec.items():
                if k.startswith("graph:") or k in ("trip:e2e:out", "trip:e2e:in", "trip:a2e:out", "trip:e2a:out", "quad:out"):
                    out[f"f32:{k}"] = to_np(v)
            for k in ("int0:0", "int0:1", "int3:0", "int3:1", "out0:0", "out0:1", "out4:0", "out4:1"):
                out[f"f32:{k}"] = to_np(rec[k])[::9]              # every 9th atom / edge row
            print("full: params", sum(p.numel() for p in net.parameters()), "E", E.numpy(), "loss", float(loss),
                  "edges", {k.split(':')[1]: v.shape[1] for k, v in rec.items() if k.endswith(':edge_index')}, "quads", rec["quad:out"].numel())
    np.savez_compressed(os.path.join(OUT, "gemnet_full.npz"), **out)
    for f in ("gemnet_small.npz", "gemnet_full.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "kB")


if __name__ == "__main__":
    main()
