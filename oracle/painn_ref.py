"""TEST INFRASTRUCTURE -- CPU restatement (oracle) of the reference PaiNN hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the product (``nabladft_amd``) never does.

Pure torch (fp32 or fp64, CPU), written from the reference's algorithm, each function
citing the reference lines it restates (paths relative to /root/reference/):

* graph:       nablaDFT/painn_pyg/painn.py:351-432 (_generate_graph, non-PBC branch),
               :306-349 (generate_graph_values), :168-304 (symmetrize_edges, live branch
               :233-282), painn_pyg/utils.py:469-481 (compute_neighbors)
* radial:      painn_pyg/layers.py:14-33 (PolynomialEnvelope), :129-185 (RadialBasis),
               PyG GaussianSmearing(0, 1, R)
* model:       painn.py:89-148 (PaiNN.forward), :449-512 (PaiNNMessage), :515-548 (PaiNNUpdate),
               layers.py:198-222 (AtomEmbedding)
* loss:        painn.py:741-745 (_calculate_loss) with torch.nn.L1Loss + gemnet_oc/loss.py:5-22
               (L2Loss), config/model/painn-oc.yaml:36-43

Pinned by ``tests/golden/*.npz`` (produced by ``oracle/make_golden.py`` from the imported
reference in the build container) -- see tests/test_oracle_golden.py.
The third-party pieces (torch_cluster.radius_graph, torch_scatter.scatter) have no
reference-side numeric test: their semantics are the documented ones.
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as Fn


@dataclass
class PaiNNConfig:
    hidden_channels: int = 128
    num_layers: int = 6
    num_rbf: int = 100
    cutoff: float = 5.0
    max_neighbors: int = 100
    envelope_exponent: int = 5
    num_elements: int = 100
    # "pyg": rbf_proj(envelope * gauss(d/rc)) + bias (nablaDFT/painn_pyg).  "spk": cosine_cutoff(d) * (filter(gauss(d)) + bias)
    # (schnetpack PaiNN, config/model/painn.yaml) -- see oracle/spk_painn_ref.py
    filter_mode: str = "pyg"
    # radial basis of RadialBasis (layers.py:168-179): "gaussian" | "spherical_bessel" (learnable frequencies [R]) | "bernstein" (learnable pregamma)
    rbf: str = "gaussian"
    # direct_forces=True: forces from the PaiNNOutput head (painn.py:551-620) instead of -dE/dpos
    direct_forces: bool = False


# ----------------------------------------------------------------------------------------
# deterministic inputs (numpy PCG64: stream is stable across numpy versions)
# ----------------------------------------------------------------------------------------
def param_shapes(cfg: PaiNNConfig):
    """state_dict layout of the reference PaiNN (painn.py:63-87)."""
    F, R = cfg.hidden_channels, cfg.num_rbf
    shapes = [("atom_emb.embeddings.weight", (cfg.num_elements, F))]
    if cfg.rbf == "spherical_bessel":
        shapes += [("radial_basis.rbf.frequencies", (R,))]
    elif cfg.rbf == "bernstein":
        shapes += [("radial_basis.rbf.pregamma", ())]
    for i in range(cfg.num_layers):
        p = f"message_layers.{i}."
        shapes += [(p + "x_proj.0.weight", (F, F)), (p + "x_proj.0.bias", (F,)),
                   (p + "x_proj.2.weight", (3 * F, F)), (p + "x_proj.2.bias", (3 * F,)),
                   (p + "rbf_proj.weight", (3 * F, R)), (p + "rbf_proj.bias", (3 * F,))]
    for i in range(cfg.num_layers):
        p = f"update_layers.{i}."
        shapes += [(p + "vec_proj.weight", (2 * F, F)),
                   (p + "xvec_proj.0.weight", (F, 2 * F)), (p + "xvec_proj.0.bias", (F,)),
                   (p + "xvec_proj.2.weight", (3 * F, F)), (p + "xvec_proj.2.bias", (3 * F,))]
    shapes += [("out_energy.0.weight", (F // 2, F)), ("out_energy.0.bias", (F // 2,)),
               ("out_energy.2.weight", (1, F // 2)), ("out_energy.2.bias", (1,))]
    if cfg.direct_forces:
        for i, (h, o) in enumerate(((F, F // 2), (F // 2, 1))):
            p = f"out_forces.output_network.{i}."
            shapes += [(p + "vec1_proj.weight", (h, h)), (p + "vec2_proj.weight", (o, h)),
                       (p + "update_net.0.weight", (h, 2 * h)), (p + "update_net.0.bias", (h,)),
                       (p + "update_net.2.weight", (2 * o, h)), (p + "update_net.2.bias", (2 * o,))]
    return shapes


def make_params(cfg: PaiNNConfig, seed: int, dtype=torch.float32):
    """Seeded weights: xavier-uniform ranges for matrices (reference initialiser family,
    painn.py:150-154,467-473,528-533), U(-sqrt3, sqrt3) embedding (layers.py:213), and small
    non-zero biases (std 0.02: keeps the 6-layer net O(1)) so that every bias path is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("rbf.frequencies"):
            a = np.pi * np.arange(1, shape[0] + 1) + rng.normal(0, 0.05, size=shape)       # layers.py:71-74 canonical positions, perturbed
        elif name.endswith("rbf.pregamma"):
            a = np.array(0.45264 + rng.normal(0, 0.05))                                    # layers.py:104
        elif name.endswith("embeddings.weight"):
            a = rng.uniform(-np.sqrt(3.0), np.sqrt(3.0), size=shape)
        elif name.endswith("weight"):
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-bound, bound, size=shape)
        else:
            a = rng.normal(0.0, 0.02, size=shape)
        out[name] = torch.tensor(a.astype(np.float32)).to(dtype)
    return out


from nabladft_amd.synth import gen_conformers  # noqa: E402,F401  (data generator shared with bench.py)


# ----------------------------------------------------------------------------------------
# graph (integer work: bit-exact contract)
# ----------------------------------------------------------------------------------------
def build_graph(pos, batch, cutoff, max_neighbors):
    """radius_graph (torch_cluster semantics: strict d^2 < r^2, no self loops, first K
    neighbours per centre in ascending index) followed by the reference's symmetrisation:
    keep j<i, then per graph [kept edges (i asc, j asc)] ++ [their flips] (painn.py:233-282).
    Returns edge_index int64 [2,E] (row0=source j, row1=target i), neighbors int64 [B],
    id_swap int64 [E] (painn.py:290-295)."""
    pos = pos.detach()
    B = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=B)
    ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    src, dst, nbrs, swap = [], [], [], []
    off = 0
    r2 = torch.tensor(cutoff * cutoff, dtype=pos.dtype)
    for g in range(B):
        a, b = int(ptr[g]), int(ptr[g + 1])
        p = pos[a:b]
        n = b - a
        d2 = (p[:, None, :] - p[None, :, :]).pow(2).sum(-1)
        adj = (d2 < r2) & ~torch.eye(n, dtype=torch.bool)
        rank = torch.cumsum(adj.long(), dim=1) - 1
        adj &= rank < max_neighbors
        lower = torch.tril(adj, diagonal=-1)  # [centre i, neighbour j<i]
        i_loc, j_loc = lower.nonzero(as_tuple=True)
        k = i_loc.numel()
        src += [j_loc + a, i_loc + a]
        dst += [i_loc + a, j_loc + a]
        nbrs.append(2 * k)
        swap += [torch.arange(k) + off + k, torch.arange(k) + off]
        off += 2 * k
    if not src:
        z = torch.zeros(0, dtype=torch.long)
        return torch.stack([z, z]), torch.zeros(B, dtype=torch.long), z
    edge_index = torch.stack([torch.cat(src), torch.cat(dst)])
    return edge_index, torch.tensor(nbrs, dtype=torch.long), torch.cat(swap)


def edge_geometry(pos, edge_index):
    """painn.py:418-420 + :319-321."""
    j, i = edge_index
    distance_vec = pos[j] - pos[i]
    edge_dist = (pos[i] - pos[j]).pow(2).sum(dim=-1).sqrt()
    mask_zero = torch.isclose(edge_dist, torch.zeros((), dtype=pos.dtype), atol=1e-6).to(pos.dtype) * 1e-6
    edge_vector = distance_vec / (edge_dist + mask_zero)[:, None]
    return edge_dist, edge_vector


def radial_basis(cfg: PaiNNConfig, d, P=None):
    """layers.py:181-185 with PolynomialEnvelope (:23-33) / ExponentialEnvelope (:36-48) and GaussianSmearing(0,1,R) /
    SphericalBesselBasis (:51-80) / BernsteinBasis (:83-126; needs the parameter dict P for the learnable basis parameters)."""
    p = float(cfg.envelope_exponent)
    a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
    ds = d * (1 / cfg.cutoff)
    if cfg.envelope_exponent == 0:                    # ExponentialEnvelope (layers.py:36-48)
        env = torch.exp(-(ds**2) / ((1 - ds) * (1 + ds)))
    else:
        env = 1 + a * ds**p + b * ds ** (p + 1) + c * ds ** (p + 2)
    env = torch.where(ds < 1, env, torch.zeros_like(ds))
    if cfg.rbf == "spherical_bessel":
        import math
        norm_const = math.sqrt(2 / (cfg.cutoff ** 3))
        return env[:, None] * (norm_const / ds[:, None] * torch.sin(P["radial_basis.rbf.frequencies"] * ds[:, None]))
    if cfg.rbf == "bernstein":
        from scipy.special import binom
        R_ = cfg.num_rbf
        prefactor = torch.tensor(binom(R_ - 1, np.arange(R_)), dtype=torch.float).to(d.dtype)
        gamma = Fn.softplus(P["radial_basis.rbf.pregamma"])
        exp_d = torch.exp(-gamma * ds)[:, None]
        exp1 = torch.arange(R_)[None, :]
        return env[:, None] * (prefactor * (exp_d ** exp1) * ((1 - exp_d) ** (R_ - 1 - exp1)))
    offset = torch.linspace(0.0, 1.0, cfg.num_rbf).to(d.dtype)
    coeff = -0.5 / (torch.linspace(0.0, 1.0, cfg.num_rbf)[1] - 0.0).item() ** 2
    g = torch.exp(coeff * (ds.view(-1, 1) - offset.view(1, -1)).pow(2))
    return env[:, None] * g


# ----------------------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------------------
def _scatter_sum(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def spk_radial(cfg, d):
    """schnetpack GaussianRBF(n_rbf, cutoff) on the unscaled distance and CosineCutoff(cutoff)."""
    import math
    offsets = torch.linspace(0.0, cfg.cutoff, cfg.num_rbf).to(d.dtype)
    width = (torch.linspace(0.0, cfg.cutoff, cfg.num_rbf)[1]).item()
    g = torch.exp((-0.5 / width ** 2) * (d[:, None] - offsets[None, :]) ** 2)
    fcut = 0.5 * (torch.cos(d * math.pi / cfg.cutoff) + 1.0) * (d < cfg.cutoff).to(d.dtype)
    return g, fcut


def message_layer(P, pre, F, x, vec, edge_index, edge_rbf, edge_vector, bias_scale=None):
    """painn.py:475-509.  bias_scale (spk filter mode): filter = fcut * (W g + b) = W (fcut g) + fcut b."""
    xh = Fn.linear(Fn.silu(Fn.linear(x, P[pre + "x_proj.0.weight"], P[pre + "x_proj.0.bias"])),
                   P[pre + "x_proj.2.weight"], P[pre + "x_proj.2.bias"])
    if bias_scale is None:
        rbfh = Fn.linear(edge_rbf, P[pre + "rbf_proj.weight"], P[pre + "rbf_proj.bias"])
    else:
        rbfh = Fn.linear(edge_rbf, P[pre + "rbf_proj.weight"]) + bias_scale[:, None] * P[pre + "rbf_proj.bias"]
    j, i = edge_index
    m = xh[j] * rbfh
    xa, xh2, xh3 = torch.split(m, F, dim=-1)
    mvec = vec[j] * xh2.unsqueeze(1) + xh3.unsqueeze(1) * edge_vector.unsqueeze(2)
    return _scatter_sum(xa, i, x.shape[0]), _scatter_sum(mvec, i, x.shape[0])


def update_layer(P, pre, F, x, vec):
    """painn.py:535-548."""
    vec1, vec2 = torch.split(Fn.linear(vec, P[pre + "vec_proj.weight"]), F, dim=-1)
    vec_dot = (vec1 * vec2).sum(dim=1)
    cat = torch.cat([x, torch.sqrt(torch.sum(vec2**2, dim=-2) + 1e-8)], dim=-1)
    h = Fn.linear(Fn.silu(Fn.linear(cat, P[pre + "xvec_proj.0.weight"], P[pre + "xvec_proj.0.bias"])),
                  P[pre + "xvec_proj.2.weight"], P[pre + "xvec_proj.2.bias"])
    xvec1, xvec2, xvec3 = torch.split(h, F, dim=-1)
    return xvec1 + xvec2 * vec_dot, xvec3.unsqueeze(1) * vec1


def scaled_silu(x):
    return Fn.silu(x) * (1 / 0.6)                       # layers.py:188-195


def painn_output_head(P, cfg: PaiNNConfig, x, vec):
    """PaiNNOutput = two GatedEquivariantBlocks (painn.py:551-620): returns forces [N, 3]."""
    F = cfg.hidden_channels
    for i, (h, o) in enumerate(((F, F // 2), (F // 2, 1))):
        p = f"out_forces.output_network.{i}."
        vec1 = torch.norm(Fn.linear(vec, P[p + "vec1_proj.weight"]), dim=-2)
        vec2 = Fn.linear(vec, P[p + "vec2_proj.weight"])
        u = scaled_silu(Fn.linear(torch.cat([x, vec1], dim=-1), P[p + "update_net.0.weight"], P[p + "update_net.0.bias"]))
        o2 = Fn.linear(u, P[p + "update_net.2.weight"], P[p + "update_net.2.bias"])
        xo, gate = torch.split(o2, o, dim=-1)
        vec = gate.unsqueeze(1) * vec2
        x = scaled_silu(xo)
    return vec.squeeze(-1)


def painn_energy(P, cfg: PaiNNConfig, pos, z, batch, edge_index, trace=None):
    """painn.py:89-128 (energy only; forces are taken by autograd in energy_forces)."""
    F = cfg.hidden_channels
    B = int(batch.max()) + 1
    edge_dist, edge_vector = edge_geometry(pos, edge_index)
    bias_scale = None
    if cfg.filter_mode == "spk":
        g, fcut = spk_radial(cfg, edge_dist)
        edge_rbf, bias_scale = g * fcut[:, None], fcut
    else:
        edge_rbf = radial_basis(cfg, edge_dist, P)
    x = P["atom_emb.embeddings.weight"][z - 1]
    vec = torch.zeros(x.size(0), 3, F, dtype=x.dtype)
    for l in range(cfg.num_layers):
        dx, dvec = message_layer(P, f"message_layers.{l}.", F, x, vec, edge_index, edge_rbf, edge_vector, bias_scale)
        x, vec = x + dx, vec + dvec
        if trace is not None:
            trace[f"x_msg{l}"], trace[f"vec_msg{l}"] = x.detach(), vec.detach()
        dx, dvec = update_layer(P, f"update_layers.{l}.", F, x, vec)
        x, vec = x + dx, vec + dvec
        if trace is not None:
            trace[f"x_upd{l}"], trace[f"vec_upd{l}"] = x.detach(), vec.detach()
    h = Fn.silu(Fn.linear(x, P["out_energy.0.weight"], P["out_energy.0.bias"]))
    per_atom = Fn.linear(h, P["out_energy.2.weight"], P["out_energy.2.bias"]).squeeze(1)
    if trace is not None:
        trace["edge_dist"], trace["edge_vector"] = edge_dist.detach(), edge_vector.detach()
        trace["edge_rbf"] = edge_rbf.detach()
    if cfg.direct_forces:
        return _scatter_sum(per_atom, batch, B), painn_output_head(P, cfg, x, vec)
    return _scatter_sum(per_atom, batch, B)


def energy_forces(P, cfg, pos, z, batch, edge_index=None, create_graph=False, trace=None):
    """painn.py:130-146: forces = -dE/dpos with grad_outputs=ones."""
    pos = pos.detach().clone().requires_grad_(True)
    if edge_index is None:
        edge_index, _, _ = build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    if cfg.direct_forces:                                 # painn.py:131-133: no autograd forces
        with torch.enable_grad():
            energy, forces = painn_energy(P, cfg, pos.detach(), z, batch, edge_index, trace)
        return (energy, forces) if create_graph else (energy.detach(), forces.detach())
    with torch.enable_grad():
        energy = painn_energy(P, cfg, pos, z, batch, edge_index, trace)
        forces = -torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy),
                                      create_graph=create_graph)[0]
    if not create_graph:
        energy, forces = energy.detach(), forces.detach()
    return energy, forces


def loss_fn(energy, forces, y, f_target, coef_e=1.0, coef_f=1.0):
    """painn.py:741-745 with L1Loss (energy) + L2Loss (forces)."""
    le = (energy - y).abs().mean()
    lf = torch.linalg.vector_norm(forces - f_target, dim=-1).mean()
    return coef_e * le + coef_f * lf


def train_step(P, cfg, pos, z, batch, y, f_target, edge_index=None):
    """One reference training step without the optimizer: forward, autograd forces with
    create_graph=True, loss, backward to parameter gradients (painn.py:642-668)."""
    names = list(P.keys())
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    energy, forces = energy_forces(Pg, cfg, pos, z, batch, edge_index, create_graph=True)
    loss = loss_fn(energy, forces, y, f_target)
    grads = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(names, grads)}
    return energy.detach(), forces.detach(), loss.detach(), grads
