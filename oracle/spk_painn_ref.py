"""TEST INFRASTRUCTURE -- CPU restatement of schnetpack 2.0.4's PaiNN potential as nablaDFT configures it
(config/model/painn.yaml: representation.PaiNN(n_atom_basis=128, n_interactions=6, GaussianRBF(100, 5.0),
CosineCutoff(5.0)) + Atomwise(n_in=128) + Forces, wrapped by nablaDFT/ase_model/task.py:9-31).

PARITY UNPINNED: schnetpack is a third-party dependency pinned in the reference's setup.py (schnetpack==2.0.4); its
source is not under /root/reference and it is not installed here, and the reference's own tests at this boundary assert
shapes only (tests/model/test_torch_models.py:30-40).  This file restates the published architecture (Schuett et al.
2021 and the schnetpack 2.0 layer definitions) -- it is the only check available for the spk path
and says so.  Parameter names follow schnetpack's module tree as recalled (unverified).

The second half maps an spk parameter set onto the engine's (painn_pyg) layout: the two PaiNNs differ only by
  * the order of the three message parts  (spk: dq | dmuR | dmumu;  pyg: x | vec_j* | r*)          -> row permutation
  * which half of the mixing projection feeds the norm (spk: first = mu_V; pyg: second = vec2)      -> row permutation
  * the order of the update outputs (spk: dq | dmu | dqmu;  pyg: xvec1 | xvec2(*dot) | xvec3(*vec)) -> row permutation
  * one shared filter network for all layers                                                       -> row slices
  * the radial filter: W_ij = fcut(d) * (filter_net(gauss(d)))  -- cosine cutoff applied AFTER the bias, Gaussians on the
    unscaled distance -- instead of  rbf_proj(envelope(d/rc) * gauss(d/rc)) + bias
  * embedding indexed by Z (padding_idx 0) instead of Z-1.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as Fn


@dataclass
class SpkPaiNNConfig:
    n_atom_basis: int = 128
    n_interactions: int = 6
    n_rbf: int = 100
    cutoff: float = 5.0
    max_z: int = 101
    epsilon: float = 1e-8


def spk_param_shapes(cfg: SpkPaiNNConfig):
    F, L, R = cfg.n_atom_basis, cfg.n_interactions, cfg.n_rbf
    s = [("representation.embedding.weight", (cfg.max_z, F)),
         ("representation.filter_net.weight", (L * 3 * F, R)), ("representation.filter_net.bias", (L * 3 * F,))]
    for l in range(L):
        p = f"representation.interactions.{l}.interatomic_context_net."
        s += [(p + "0.weight", (F, F)), (p + "0.bias", (F,)), (p + "1.weight", (3 * F, F)), (p + "1.bias", (3 * F,))]
    for l in range(L):
        p = f"representation.mixing.{l}."
        s += [(p + "intraatomic_context_net.0.weight", (F, 2 * F)), (p + "intraatomic_context_net.0.bias", (F,)),
              (p + "intraatomic_context_net.1.weight", (3 * F, F)), (p + "intraatomic_context_net.1.bias", (3 * F,)),
              (p + "mu_channel_mix.weight", (2 * F, F))]
    s += [("output_modules.0.outnet.0.weight", (F // 2, F)), ("output_modules.0.outnet.0.bias", (F // 2,)),
          ("output_modules.0.outnet.1.weight", (1, F // 2)), ("output_modules.0.outnet.1.bias", (1,))]
    return s


def make_spk_params(cfg: SpkPaiNNConfig, seed: int, dtype=torch.float32):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in spk_param_shapes(cfg):
        if name.endswith("embedding.weight"):
            a = rng.normal(0.0, 1.0, size=shape)
            a[0] = 0.0                                   # padding_idx = 0
        elif name.endswith("weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-bound, bound, size=shape)
        else:
            a = rng.normal(0.0, 0.02, size=shape)
        out[name] = torch.tensor(a.astype(np.float32)).to(dtype)
    return out


def full_neighbor_list(pos, batch, cutoff):
    """ASE-style list used by schnetpack (config/datamodule/nablaDFT_ase.yaml:12-14): all ordered pairs with d < cutoff."""
    idx_i, idx_j = [], []
    B = int(batch.max()) + 1
    for g in range(B):
        sel = (batch == g).nonzero().flatten()
        p = pos[sel].detach()
        d = (p[:, None] - p[None]).norm(dim=-1)
        m = (d < cutoff) & ~torch.eye(len(sel), dtype=torch.bool)
        i, j = m.nonzero(as_tuple=True)
        idx_i.append(sel[i]), idx_j.append(sel[j])
    return torch.cat(idx_i), torch.cat(idx_j)


def spk_painn_energy(P, cfg: SpkPaiNNConfig, pos, z, idx_i, idx_j, idx_m):
    F, L = cfg.n_atom_basis, cfg.n_interactions
    n = pos.shape[0]
    r_ij = pos[idx_j] - pos[idx_i]
    d = torch.linalg.norm(r_ij, dim=1)
    dir_ij = r_ij / d[:, None]
    offsets = torch.linspace(0.0, cfg.cutoff, cfg.n_rbf).to(pos.dtype)
    width = (torch.linspace(0.0, cfg.cutoff, cfg.n_rbf)[1]).item()
    phi = torch.exp((-0.5 / width ** 2) * (d[:, None] - offsets[None, :]) ** 2)
    fcut = 0.5 * (torch.cos(d * math.pi / cfg.cutoff) + 1.0) * (d < cfg.cutoff).to(pos.dtype)
    filters = Fn.linear(phi, P["representation.filter_net.weight"], P["representation.filter_net.bias"]) * fcut[:, None]
    q = P["representation.embedding.weight"][z]
    mu = torch.zeros(n, 3, F, dtype=pos.dtype)
    for l in range(L):
        pi = f"representation.interactions.{l}.interatomic_context_net."
        x = Fn.linear(Fn.silu(Fn.linear(q, P[pi + "0.weight"], P[pi + "0.bias"])), P[pi + "1.weight"], P[pi + "1.bias"])
        W = filters[:, l * 3 * F:(l + 1) * 3 * F]
        xj = W * x[idx_j]
        dq, dmuR, dmumu = torch.split(xj, F, dim=-1)
        dmu = dmuR[:, None, :] * dir_ij[:, :, None] + dmumu[:, None, :] * mu[idx_j]
        q = q + torch.zeros_like(q).index_add_(0, idx_i, dq)
        mu = mu + torch.zeros_like(mu).index_add_(0, idx_i, dmu)
        pm = f"representation.mixing.{l}."
        mu_mix = Fn.linear(mu, P[pm + "mu_channel_mix.weight"])
        mu_V, mu_W = torch.split(mu_mix, F, dim=-1)
        mu_Vn = torch.sqrt(torch.sum(mu_V ** 2, dim=-2) + cfg.epsilon)
        ctx = torch.cat([q, mu_Vn], dim=-1)
        y = Fn.linear(Fn.silu(Fn.linear(ctx, P[pm + "intraatomic_context_net.0.weight"], P[pm + "intraatomic_context_net.0.bias"])),
                      P[pm + "intraatomic_context_net.1.weight"], P[pm + "intraatomic_context_net.1.bias"])
        dq_i, dmu_i, dqmu_i = torch.split(y, F, dim=-1)
        q = q + dq_i + dqmu_i * torch.sum(mu_V * mu_W, dim=1)
        mu = mu + dmu_i[:, None, :] * mu_W
    h = Fn.silu(Fn.linear(q, P["output_modules.0.outnet.0.weight"], P["output_modules.0.outnet.0.bias"]))
    yi = Fn.linear(h, P["output_modules.0.outnet.1.weight"], P["output_modules.0.outnet.1.bias"]).squeeze(1)
    B = int(idx_m.max()) + 1
    return torch.zeros(B, dtype=pos.dtype).index_add_(0, idx_m, yi)


def spk_train_step(P, cfg, pos, z, batch, y, f_target, w_e=1.0, w_f=1.0):
    """energy, forces (= -dE/dR, create_graph), MSE losses as config/model/painn.yaml:30-46, parameter gradients."""
    names = list(P.keys())
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    R_ = pos.detach().clone().requires_grad_(True)
    idx_i, idx_j = full_neighbor_list(R_, batch, cfg.cutoff)
    energy = spk_painn_energy(Pg, cfg, R_, z, idx_i, idx_j, batch)
    forces = -torch.autograd.grad(energy, R_, torch.ones_like(energy), create_graph=True)[0]
    loss = w_e * Fn.mse_loss(energy, y) + w_f * Fn.mse_loss(forces, f_target)
    grads = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(names, grads)}
    return energy.detach(), forces.detach(), loss.detach(), grads


# ----------------------------------------------------------------------------------------------------------------------
# mapping spk parameters <-> engine (painn_pyg state_dict order) layout
# ----------------------------------------------------------------------------------------------------------------------
def _perm3(F, order):
    return torch.cat([torch.arange(F) + o * F for o in order])


def spk_to_engine_index(cfg: SpkPaiNNConfig):
    """Returns (spk_flat_index int64[P_engine], names) such that engine_flat = spk_flat[spk_flat_index] where spk_flat is the
    concatenation of the spk tensors in spk_param_shapes order (row-major).  Pure gather: every engine element is one spk
    element (embedding row 0 = padding row is dropped)."""
    F, L, R = cfg.n_atom_basis, cfg.n_interactions, cfg.n_rbf
    shapes = spk_param_shapes(cfg)
    off, o = {}, 0
    for name, shp in shapes:
        off[name] = (o, shp)
        o += int(np.prod(shp))

    def rows(name, row_idx):
        o0, shp = off[name]
        cols = shp[1] if len(shp) == 2 else 1
        return (o0 + row_idx[:, None] * cols + torch.arange(cols)[None, :]).reshape(-1)

    def whole(name):
        o0, shp = off[name]
        return torch.arange(o0, o0 + int(np.prod(shp)))

    msg_perm = _perm3(F, [0, 2, 1])      # pyg (x | vec_j* | r*)  <- spk (dq | dmumu | dmuR)
    upd_perm = _perm3(F, [0, 2, 1])      # pyg (xvec1 | xvec2 | xvec3) <- spk (dq | dqmu | dmu)
    mix_perm = torch.cat([torch.arange(F) + F, torch.arange(F)])   # pyg (vec1 | vec2) <- spk (mu_W | mu_V)
    idx = [rows("representation.embedding.weight", torch.arange(1, cfg.max_z))]
    for l in range(L):
        pi = f"representation.interactions.{l}.interatomic_context_net."
        idx += [whole(pi + "0.weight"), whole(pi + "0.bias"), rows(pi + "1.weight", msg_perm), rows(pi + "1.bias", msg_perm),
                rows("representation.filter_net.weight", l * 3 * F + msg_perm), rows("representation.filter_net.bias", l * 3 * F + msg_perm)]
    for l in range(L):
        pm = f"representation.mixing.{l}."
        idx += [rows(pm + "mu_channel_mix.weight", mix_perm), whole(pm + "intraatomic_context_net.0.weight"),
                whole(pm + "intraatomic_context_net.0.bias"), rows(pm + "intraatomic_context_net.1.weight", upd_perm),
                rows(pm + "intraatomic_context_net.1.bias", upd_perm)]
    idx += [whole("output_modules.0.outnet.0.weight"), whole("output_modules.0.outnet.0.bias"),
            whole("output_modules.0.outnet.1.weight"), whole("output_modules.0.outnet.1.bias")]
    return torch.cat(idx)
