"""TEST INFRASTRUCTURE -- CPU restatement of schnetpack 2.0.4's SchNet potential as nablaDFT configures it
(config/model/schnet.yaml:4-28: representation.SchNet(n_atom_basis=128, n_interactions=6, GaussianRBF(100, 5.0),
CosineCutoff(5.0)) + Atomwise(n_in=128) + Forces, wrapped by nablaDFT/ase_model/task.py:9-31; BASELINE.json configs[0]).

PARITY UNPINNED, exactly as oracle/spk_painn_ref.py: schnetpack is a pip dependency of the reference (setup.py,
schnetpack==2.0.4), its source is not under /root/reference and it is not installed here; the reference's tests at this
boundary assert shapes only (tests/model/test_torch_models.py:30-40).  Restated from the published architecture (Schuett et
al. 2018) and SURVEY.md Appendix C; parameter names follow schnetpack's module tree as recalled (unverified).

Two halves:
  * `schnet_energy` / `schnet_train_step`: plain autograd restatement (the check of last resort for the spk-SchNet path);
  * `SchNetSweeps`: the same training step as four hand-derived sweeps without autograd (forward, force adjoint, tangent,
    dual reverse -- see oracle/painn_sweeps.py for the scheme), with the buffer names of the HIP engine; verified against
    the autograd half in fp64 by tests/test_schnet_cpu.py.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as Fn

from oracle.spk_painn_ref import full_neighbor_list

LN2 = math.log(2.0)


@dataclass
class SchNetConfig:
    n_atom_basis: int = 128
    n_interactions: int = 6
    n_rbf: int = 100
    cutoff: float = 5.0
    max_z: int = 101


def ssp(x):
    return Fn.softplus(x) - LN2


def schnet_param_shapes(cfg: SchNetConfig):
    F, L, R = cfg.n_atom_basis, cfg.n_interactions, cfg.n_rbf
    s = [("representation.embedding.weight", (cfg.max_z, F))]
    for l in range(L):
        p = f"representation.interactions.{l}."
        s += [(p + "in2f.weight", (F, F)),
              (p + "filter_network.0.weight", (F, R)), (p + "filter_network.0.bias", (F,)),
              (p + "filter_network.1.weight", (F, F)), (p + "filter_network.1.bias", (F,)),
              (p + "f2out.0.weight", (F, F)), (p + "f2out.0.bias", (F,)),
              (p + "f2out.1.weight", (F, F)), (p + "f2out.1.bias", (F,))]
    s += [("output_modules.0.outnet.0.weight", (F // 2, F)), ("output_modules.0.outnet.0.bias", (F // 2,)),
          ("output_modules.0.outnet.1.weight", (1, F // 2)), ("output_modules.0.outnet.1.bias", (1,))]
    return s


def make_schnet_params(cfg: SchNetConfig, seed: int, dtype=torch.float32):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in schnet_param_shapes(cfg):
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


def gauss_and_cutoff(cfg: SchNetConfig, d):
    """g[e,k] = exp(-0.5/w^2 (d-mu_k)^2) on the unscaled distance, its d-derivative, fcut(d) and fcut'(d)."""
    mu = torch.linspace(0.0, cfg.cutoff, cfg.n_rbf).to(d.dtype)
    width = (torch.linspace(0.0, cfg.cutoff, cfg.n_rbf)[1]).item()
    coeff = -0.5 / width ** 2
    diff = d[:, None] - mu[None, :]
    g = torch.exp(coeff * diff * diff)
    dg = g * (2 * coeff) * diff
    inside = (d < cfg.cutoff).to(d.dtype)
    rc = 0.5 * (torch.cos(d * math.pi / cfg.cutoff) + 1.0) * inside
    drc = -0.5 * math.pi / cfg.cutoff * torch.sin(d * math.pi / cfg.cutoff) * inside
    return g, dg, rc, drc


def schnet_energy(P, cfg: SchNetConfig, pos, z, idx_i, idx_j, idx_m):
    L = cfg.n_interactions
    r_ij = pos[idx_j] - pos[idx_i]
    d = torch.linalg.norm(r_ij, dim=1)
    g, _, rc, _ = gauss_and_cutoff(cfg, d)
    x = P["representation.embedding.weight"][z]
    for l in range(L):
        p = f"representation.interactions.{l}."
        y = Fn.linear(x, P[p + "in2f.weight"])
        W = Fn.linear(ssp(Fn.linear(g, P[p + "filter_network.0.weight"], P[p + "filter_network.0.bias"])),
                      P[p + "filter_network.1.weight"], P[p + "filter_network.1.bias"]) * rc[:, None]
        m = torch.zeros_like(x).index_add_(0, idx_i, y[idx_j] * W)
        v = Fn.linear(ssp(Fn.linear(m, P[p + "f2out.0.weight"], P[p + "f2out.0.bias"])), P[p + "f2out.1.weight"], P[p + "f2out.1.bias"])
        x = x + v
    h = Fn.silu(Fn.linear(x, P["output_modules.0.outnet.0.weight"], P["output_modules.0.outnet.0.bias"]))
    yi = Fn.linear(h, P["output_modules.0.outnet.1.weight"], P["output_modules.0.outnet.1.bias"]).squeeze(1)
    B = int(idx_m.max()) + 1
    return torch.zeros(B, dtype=pos.dtype).index_add_(0, idx_m, yi)


def schnet_train_step(P, cfg, pos, z, batch, y, f_target, w_e=1.0, w_f=1.0):
    """energy, forces (= -dE/dR, create_graph), MSE losses as config/model/schnet.yaml:30-46, parameter gradients."""
    names = list(P.keys())
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    R_ = pos.detach().clone().requires_grad_(True)
    idx_i, idx_j = full_neighbor_list(R_, batch, cfg.cutoff)
    energy = schnet_energy(Pg, cfg, R_, z, idx_i, idx_j, batch)
    forces = -torch.autograd.grad(energy, R_, torch.ones_like(energy), create_graph=True)[0]
    loss = w_e * Fn.mse_loss(energy, y) + w_f * Fn.mse_loss(forces, f_target)
    grads = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(names, grads)}
    return energy.detach(), forces.detach(), loss.detach(), grads


# ----------------------------------------------------------------------------------------------------------------------
# the four sweeps without autograd (what the HIP engine computes)
# ----------------------------------------------------------------------------------------------------------------------
def sig(x):
    return torch.sigmoid(x)


def dsig(x):
    s = torch.sigmoid(x)
    return s * (1 - s)


def silu(x):
    return x * torch.sigmoid(x)


def dsilu(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def d2silu(x):
    s = torch.sigmoid(x)
    return s * (1 - s) * (2 + x * (1 - 2 * s))


class SchNetSweeps:
    """Edges e = (i <- j): idx_i[e] = centre (CSR row), idx_j[e] = neighbour.  d_e = |r_j - r_i|, u_e = (r_j - r_i)/d_e.
    All per-edge filter quantities depend on d_e only, hence are identical for e and its reverse edge: every reverse-mode
    scatter over the source index is evaluated as a gather over the atom's own row (no atomics on the GPU)."""

    def __init__(self, P, cfg: SchNetConfig, pos, z, batch, idx_i, idx_j):
        self.P, self.cfg, self.pos, self.z, self.batch, self.i, self.j = P, cfg, pos, z, batch, idx_i, idx_j
        self.N, self.B = pos.shape[0], int(batch.max()) + 1
        r = pos[idx_j] - pos[idx_i]
        self.d = torch.linalg.norm(r, dim=1)
        self.u = r / self.d[:, None]
        self.g, self.dg, self.rc, self.drc = gauss_and_cutoff(cfg, self.d)
        self.S = {}

    def _w(self, l):
        p = f"representation.interactions.{l}."
        P = self.P
        return (P[p + "in2f.weight"], P[p + "filter_network.0.weight"], P[p + "filter_network.0.bias"], P[p + "filter_network.1.weight"],
                P[p + "filter_network.1.bias"], P[p + "f2out.0.weight"], P[p + "f2out.0.bias"], P[p + "f2out.1.weight"], P[p + "f2out.1.bias"])

    def _gather_sum(self, node_vals, edge_vals):
        """out_i = sum_{e in row i} node_vals[j_e] * edge_vals[e]"""
        return torch.zeros_like(node_vals).index_add_(0, self.i, node_vals[self.j] * edge_vals)

    # 1 ---------------------------------------------------------------------------------------------------------------
    def forward(self):
        S, P = self.S, self.P
        x = P["representation.embedding.weight"][self.z]
        S["x", 0] = x
        for l in range(self.cfg.n_interactions):
            Win, W1, b1, W2, b2, Wo1, bo1, Wo2, bo2 = self._w(l)
            S["z1", l] = self.g @ W1.T + b1
            S["a1", l] = ssp(S["z1", l])
            S["h2", l] = S["a1", l] @ W2.T + b2
            S["y", l] = x @ Win.T
            S["m", l] = self._gather_sum(S["y", l], S["h2", l] * self.rc[:, None])
            S["t1", l] = S["m", l] @ Wo1.T + bo1
            S["u", l] = ssp(S["t1", l])
            x = x + S["u", l] @ Wo2.T + bo2
            S["x", l + 1] = x
        S["r1"] = x @ P["output_modules.0.outnet.0.weight"].T + P["output_modules.0.outnet.0.bias"]
        eps = silu(S["r1"]) @ P["output_modules.0.outnet.1.weight"].T + P["output_modules.0.outnet.1.bias"]
        return torch.zeros(self.B, dtype=x.dtype).index_add_(0, self.batch, eps.squeeze(1))

    # 2 ---------------------------------------------------------------------------------------------------------------
    def force_adjoint(self):
        """seeds dE_tot/d eps_i = 1  ->  gd[e] = dE_tot/d d_e (per directed edge)  ->  forces"""
        S, P = self.S, self.P
        L = self.cfg.n_interactions
        gx = (torch.ones(self.N, 1, dtype=self.pos.dtype) @ P["output_modules.0.outnet.1.weight"]) * dsilu(S["r1"])
        gx = gx @ P["output_modules.0.outnet.0.weight"]
        gd = torch.zeros_like(self.d)
        for l in reversed(range(L)):
            Win, W1, b1, W2, b2, Wo1, bo1, Wo2, bo2 = self._w(l)
            gm = ((gx @ Wo2) * sig(S["t1", l])) @ Wo1
            gW = gm[self.i] * S["y", l][self.j]                                  # adjoint of the (cutoff-scaled) filter, per edge
            gd = gd + self.drc * (gW * S["h2", l]).sum(1)
            gz1 = ((gW * self.rc[:, None]) @ W2) * sig(S["z1", l])
            gd = gd + ((gz1 @ W1) * self.dg).sum(1)
            gy = self._gather_sum(gm, S["h2", l] * self.rc[:, None])             # own-row gather (symmetric filter)
            gx = gx + gy @ Win
        self.S["gd"] = gd
        f = torch.zeros_like(self.pos)
        f.index_add_(0, self.i, self.u * gd[:, None])                           # F_i = -dE/dr_i, dd_e/dr_i = -u_e
        f.index_add_(0, self.j, -self.u * gd[:, None])
        return f

    # 3 ---------------------------------------------------------------------------------------------------------------
    def tangent(self, pos_dot):
        S, P = self.S, self.P
        td = (self.u * (pos_dot[self.j] - pos_dot[self.i])).sum(1)
        S["td"] = td
        tx = torch.zeros_like(S["x", 0])
        for l in range(self.cfg.n_interactions):
            Win, W1, b1, W2, b2, Wo1, bo1, Wo2, bo2 = self._w(l)
            S["tz1", l] = td[:, None] * (self.dg @ W1.T)
            S["ta1", l] = sig(S["z1", l]) * S["tz1", l]
            S["th2", l] = S["ta1", l] @ W2.T
            S["tx", l] = tx
            S["ty", l] = tx @ Win.T
            S["wt", l] = S["th2", l] * self.rc[:, None] + S["h2", l] * (self.drc * td)[:, None]     # tangent of the scaled filter
            S["tm", l] = self._gather_sum(S["ty", l], S["h2", l] * self.rc[:, None]) + self._gather_sum(S["y", l], S["wt", l])
            S["tt1", l] = S["tm", l] @ Wo1.T
            S["tu", l] = sig(S["t1", l]) * S["tt1", l]
            tx = tx + S["tu", l] @ Wo2.T
        S["tx", self.cfg.n_interactions] = tx
        S["tr1"] = tx @ P["output_modules.0.outnet.0.weight"].T
        teps = (dsilu(S["r1"]) * S["tr1"]) @ P["output_modules.0.outnet.1.weight"].T
        return torch.zeros(self.B, dtype=tx.dtype).index_add_(0, self.batch, teps.squeeze(1))

    # 4 ---------------------------------------------------------------------------------------------------------------
    def dual_reverse(self, ge_mol):
        """seeds (a_b, 1) on (E_b, Edot) -> parameter gradients (dict by name)."""
        S, P = self.S, self.P
        L = self.cfg.n_interactions
        G = {k: torch.zeros_like(v) for k, v in P.items()}
        Wr1, Wr2 = P["output_modules.0.outnet.0.weight"], P["output_modules.0.outnet.1.weight"]
        ge = ge_mol[self.batch][:, None]                               # adjoint of eps_i
        gte = torch.ones(self.N, 1, dtype=self.pos.dtype)              # adjoint of teps_i
        hr, thr = silu(S["r1"]), dsilu(S["r1"]) * S["tr1"]
        G["output_modules.0.outnet.1.weight"] = ge.T @ hr + gte.T @ thr
        G["output_modules.0.outnet.1.bias"] = ge.sum(0)
        ghr, gthr = ge @ Wr2, gte @ Wr2
        gr1 = ghr * dsilu(S["r1"]) + gthr * d2silu(S["r1"]) * S["tr1"]
        gtr1 = gthr * dsilu(S["r1"])
        G["output_modules.0.outnet.0.weight"] = gr1.T @ S["x", L] + gtr1.T @ S["tx", L]
        G["output_modules.0.outnet.0.bias"] = gr1.sum(0)
        gx, gtx = gr1 @ Wr1, gtr1 @ Wr1
        for l in reversed(range(L)):
            p = f"representation.interactions.{l}."
            Win, W1, b1, W2, b2, Wo1, bo1, Wo2, bo2 = self._w(l)
            rc, td, drc = self.rc[:, None], S["td"][:, None], self.drc[:, None]
            # f2out
            G[p + "f2out.1.weight"] = gx.T @ S["u", l] + gtx.T @ S["tu", l]
            G[p + "f2out.1.bias"] = gx.sum(0)
            gu, gtu = gx @ Wo2, gtx @ Wo2
            gt1 = gu * sig(S["t1", l]) + gtu * dsig(S["t1", l]) * S["tt1", l]
            gtt1 = gtu * sig(S["t1", l])
            G[p + "f2out.0.weight"] = gt1.T @ S["m", l] + gtt1.T @ S["tm", l]
            G[p + "f2out.0.bias"] = gt1.sum(0)
            gm, gtm = gt1 @ Wo1, gtt1 @ Wo1
            # continuous-filter convolution
            yj, tyj = S["y", l][self.j], S["ty", l][self.j]
            gh2 = gm[self.i] * yj * rc + gtm[self.i] * (tyj * rc + yj * drc * td)
            gth2 = gtm[self.i] * yj * rc
            gy = self._gather_sum(gm, S["h2", l] * rc) + self._gather_sum(gtm, S["wt", l])
            gty = self._gather_sum(gtm, S["h2", l] * rc)
            # filter network
            G[p + "filter_network.1.weight"] = gh2.T @ S["a1", l] + gth2.T @ S["ta1", l]
            G[p + "filter_network.1.bias"] = gh2.sum(0)
            ga1, gta1 = gh2 @ W2, gth2 @ W2
            gz1 = ga1 * sig(S["z1", l]) + gta1 * dsig(S["z1", l]) * S["tz1", l]
            gtz1 = gta1 * sig(S["z1", l])
            G[p + "filter_network.0.weight"] = gz1.T @ self.g + (gtz1 * td).T @ self.dg
            G[p + "filter_network.0.bias"] = gz1.sum(0)
            # in2f and the residual stream
            G[p + "in2f.weight"] = gy.T @ S["x", l] + gty.T @ S["tx", l]
            gx, gtx = gx + gy @ Win, gtx + gty @ Win
        G["representation.embedding.weight"] = torch.zeros_like(P["representation.embedding.weight"]).index_add_(0, self.z, gx)
        return G

    def train_step(self, y, f_target, w_e=1.0, w_f=1.0):
        energy = self.forward()
        forces = self.force_adjoint()
        gE = 2 * w_e * (energy - y) / energy.numel()
        gF = 2 * w_f * (forces - f_target) / forces.numel()
        loss = w_e * ((energy - y) ** 2).mean() + w_f * ((forces - f_target) ** 2).mean()
        self.tangent(-gF)
        return energy, forces, loss, self.dual_reverse(gE)
