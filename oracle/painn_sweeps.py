"""TEST INFRASTRUCTURE -- explicit-sweep restatement of one PaiNN training step.

The reference obtains forces and parameter gradients from torch autograd (painn.py:135-146
with create_graph=True, then loss.backward()).  The HIP engine cannot call autograd, so it
runs four hand-derived sweeps.  This file states those sweeps in plain torch ops (no
autograd anywhere) with the SAME buffer names the engine uses for its workspace, so that

  * the derivation is verified here, in fp64, against the autograd oracle (painn_ref.py)
    -- tests/test_sweeps_cpu.py, and
  * every engine buffer can be compared one-by-one on the GPU (tests/test_engine_gpu.py).

Sweeps (E_tot = sum_b E_b; a_b = dL/dE_b; g_i = dL/dF_i supplied by the loss):
  1. forward            x, vec, ... -> E_b
  2. force adjoint      seeds dE_tot/de_i = 1 -> gd[e], gr[e,3] -> F = -dE_tot/dpos
  3. tangent forward    direction pos_dot = -g  ->  t_* (JVP of every activation), Edot
  4. dual reverse       seeds (a_b, 1) on (E_b, Edot) -> dL/dtheta
     because dL/dtheta = sum_b a_b dE_b/dtheta + sum_i g_i dF_i/dtheta
                       = d/dtheta [ sum_b a_b E_b + Edot ],   Edot = d/deps E_tot(pos - eps g).

Notation in the code: ``g_X`` adjoint of primal buffer X, ``gt_X`` adjoint of tangent buffer
t_X.  sig = SiLU.
"""
import torch

from oracle.painn_ref import PaiNNConfig


def silu(z):
    return z * torch.sigmoid(z)


def dsilu(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def d2silu(z):
    s = torch.sigmoid(z)
    ds = s * (1 - s)
    return ds * (2 + z * (1 - 2 * s))


def rbf_and_derivative(cfg: PaiNNConfig, d):
    """rho[e,k] = env(d/rc) * exp(coeff (d/rc - mu_k)^2) and d rho / d d."""
    p = float(cfg.envelope_exponent)
    a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
    inv = 1.0 / cfg.cutoff
    ds = d * inv
    inside = (ds < 1).to(d.dtype)
    if cfg.envelope_exponent == 0:                    # ExponentialEnvelope
        dsc = torch.where(ds < 1, ds, torch.zeros_like(ds))
        om = (1 - dsc) * (1 + dsc)
        env = torch.exp(-(dsc**2) / om) * inside
        denv = -env * 2 * dsc / (om * om)
    else:
        env = (1 + a * ds**p + b * ds ** (p + 1) + c * ds ** (p + 2)) * inside
        denv = (a * p * ds ** (p - 1) + b * (p + 1) * ds**p + c * (p + 2) * ds ** (p + 1)) * inside
    mu = torch.linspace(0.0, 1.0, cfg.num_rbf).to(d.dtype)
    coeff = -0.5 / (torch.linspace(0.0, 1.0, cfg.num_rbf)[1]).item() ** 2
    diff = ds[:, None] - mu[None, :]
    g = torch.exp(coeff * diff * diff)
    rho = env[:, None] * g
    drho = inv * g * (denv[:, None] + env[:, None] * (2 * coeff) * diff)
    return rho, drho


def spk_rbf_and_derivative(cfg: PaiNNConfig, d):
    """spk filter mode: rho = fcut(d) * gauss(d) (unscaled Gaussians, cosine cutoff), beta = fcut multiplies the bias."""
    import math
    mu = torch.linspace(0.0, cfg.cutoff, cfg.num_rbf).to(d.dtype)
    width = (torch.linspace(0.0, cfg.cutoff, cfg.num_rbf)[1]).item()
    coeff = -0.5 / width ** 2
    diff = d[:, None] - mu[None, :]
    g = torch.exp(coeff * diff * diff)
    inside = (d < cfg.cutoff).to(d.dtype)
    fcut = 0.5 * (torch.cos(d * math.pi / cfg.cutoff) + 1.0) * inside
    dfcut = -0.5 * math.pi / cfg.cutoff * torch.sin(d * math.pi / cfg.cutoff) * inside
    rho = fcut[:, None] * g
    drho = dfcut[:, None] * g + fcut[:, None] * g * (2 * coeff) * diff
    return rho, drho, fcut, dfcut


class Sweeps:
    def __init__(self, P, cfg: PaiNNConfig, pos, z, batch, edge_index):
        self.P, self.cfg = P, cfg
        self.F, self.L = cfg.hidden_channels, cfg.num_layers
        self.pos, self.z, self.batch = pos, z, batch
        self.src, self.dst = edge_index[0], edge_index[1]
        self.N, self.E, self.B = pos.shape[0], edge_index.shape[1], int(batch.max()) + 1
        self.ws = {}

    # ------------------------------------------------------------------ helpers
    def _w(self, l, kind):
        P = self.P
        m, u = f"message_layers.{l}.", f"update_layers.{l}."
        return {
            "W1": P[m + "x_proj.0.weight"], "b1": P[m + "x_proj.0.bias"],
            "W2": P[m + "x_proj.2.weight"], "b2": P[m + "x_proj.2.bias"],
            "Wr": P[m + "rbf_proj.weight"], "br": P[m + "rbf_proj.bias"],
            "U": P[u + "vec_proj.weight"],
            "V1": P[u + "xvec_proj.0.weight"], "c1": P[u + "xvec_proj.0.bias"],
            "V2": P[u + "xvec_proj.2.weight"], "c2": P[u + "xvec_proj.2.bias"],
        }[kind]

    def _scatter(self, src, index, n):
        return torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)

    # ------------------------------------------------------------------ 1. forward
    def forward(self):
        ws, F = self.ws, self.F
        j, i = self.src, self.dst
        w = self.pos[j] - self.pos[i]
        d = (self.pos[i] - self.pos[j]).pow(2).sum(-1).sqrt()
        c0 = torch.isclose(d, torch.zeros((), dtype=d.dtype), atol=1e-6).to(d.dtype) * 1e-6
        ws["d"], ws["r"] = d, w / (d + c0)[:, None]
        if self.cfg.filter_mode == "spk":
            ws["rho"], ws["drho"], ws["beta"], ws["dbeta"] = spk_rbf_and_derivative(self.cfg, d)
        else:
            ws["rho"], ws["drho"] = rbf_and_derivative(self.cfg, d)
            ws["beta"], ws["dbeta"] = torch.ones_like(d), torch.zeros_like(d)
        x = self.P["atom_emb.embeddings.weight"][self.z - 1]
        vec = torch.zeros(self.N, 3, F, dtype=x.dtype)
        ws["x_in0"], ws["vec_in0"] = x, vec
        for l in range(self.L):
            z1 = x @ self._w(l, "W1").T + self._w(l, "b1")
            h = silu(z1)
            xh = h @ self._w(l, "W2").T + self._w(l, "b2")
            phi = ws["rho"] @ self._w(l, "Wr").T + ws["beta"][:, None] * self._w(l, "br")
            psi = ws["drho"] @ self._w(l, "Wr").T + ws["dbeta"][:, None] * self._w(l, "br")
            m = xh[j] * phi
            ma, mb, mc = m[:, :F], m[:, F:2 * F], m[:, 2 * F:]
            dx = self._scatter(ma, i, self.N)
            dvec = self._scatter(vec[j] * mb[:, None, :] + mc[:, None, :] * ws["r"][:, :, None], i, self.N)
            xm, vm = x + dx, vec + dvec
            u = vm @ self._w(l, "U").T
            v1, v2 = u[..., :F], u[..., F:]
            s = (v1 * v2).sum(1)
            n = torch.sqrt((v2 * v2).sum(1) + 1e-8)
            cat = torch.cat([xm, n], -1)
            zq = cat @ self._w(l, "V1").T + self._w(l, "c1")
            q = silu(zq)
            y = q @ self._w(l, "V2").T + self._w(l, "c2")
            ya, yb, yc = y[:, :F], y[:, F:2 * F], y[:, 2 * F:]
            xu = xm + ya + yb * s
            vu = vm + yc[:, None, :] * v1
            ws.update({f"z1_{l}": z1, f"h_{l}": h, f"xh_{l}": xh, f"phi_{l}": phi, f"psi_{l}": psi,
                       f"x_msg{l}": xm, f"vec_msg{l}": vm, f"u_{l}": u, f"s_{l}": s, f"cat_{l}": cat,
                       f"zq_{l}": zq, f"q_{l}": q, f"y_{l}": y, f"x_upd{l}": xu, f"vec_upd{l}": vu,
                       f"x_in{l + 1}": xu, f"vec_in{l + 1}": vu})
            x, vec = xu, vu
        zo = x @ self.P["out_energy.0.weight"].T + self.P["out_energy.0.bias"]
        e_atom = silu(zo) @ self.P["out_energy.2.weight"][0] + self.P["out_energy.2.bias"][0]
        ws["zo"], ws["e_atom"] = zo, e_atom
        ws["energy"] = self._scatter(e_atom, self.batch, self.B)
        return ws["energy"]

    # ------------------------------------------------------------------ 3. tangent forward
    def tangent(self, pos_dot):
        """JVP of the forward along pos_dot (N,3). Needs forward() buffers."""
        ws, F = self.ws, self.F
        j, i = self.src, self.dst
        wd = pos_dot[j] - pos_dot[i]
        t_d = (ws["r"] * wd).sum(-1)
        t_r = (wd - ws["r"] * t_d[:, None]) / ws["d"][:, None]
        ws["t_d"], ws["t_r"] = t_d, t_r
        tx = torch.zeros_like(ws["x_in0"])
        tvec = torch.zeros_like(ws["vec_in0"])
        ws["t_x_in0"], ws["t_vec_in0"] = tx, tvec
        for l in range(self.L):
            vec, xh, phi, psi = ws[f"vec_in{l}"], ws[f"xh_{l}"], ws[f"phi_{l}"], ws[f"psi_{l}"]
            tz1 = tx @ self._w(l, "W1").T
            th = dsilu(ws[f"z1_{l}"]) * tz1
            txh = th @ self._w(l, "W2").T
            tphi = psi * t_d[:, None]
            m = xh[j] * phi
            tm = txh[j] * phi + xh[j] * tphi
            mb, mc = m[:, F:2 * F], m[:, 2 * F:]
            tma, tmb, tmc = tm[:, :F], tm[:, F:2 * F], tm[:, 2 * F:]
            tdx = self._scatter(tma, i, self.N)
            tdvec = self._scatter(tvec[j] * mb[:, None, :] + vec[j] * tmb[:, None, :]
                                  + tmc[:, None, :] * ws["r"][:, :, None] + mc[:, None, :] * t_r[:, :, None], i, self.N)
            txm, tvm = tx + tdx, tvec + tdvec
            u = ws[f"u_{l}"]
            tu = tvm @ self._w(l, "U").T
            v1, v2, tv1, tv2 = u[..., :F], u[..., F:], tu[..., :F], tu[..., F:]
            n = ws[f"cat_{l}"][:, F:]
            ts = (tv1 * v2 + v1 * tv2).sum(1)
            tn = (v2 * tv2).sum(1) / n
            tcat = torch.cat([txm, tn], -1)
            tzq = tcat @ self._w(l, "V1").T
            tq = dsilu(ws[f"zq_{l}"]) * tzq
            ty = tq @ self._w(l, "V2").T
            y = ws[f"y_{l}"]
            yb, yc = y[:, F:2 * F], y[:, 2 * F:]
            tya, tyb, tyc = ty[:, :F], ty[:, F:2 * F], ty[:, 2 * F:]
            txu = txm + tya + tyb * ws[f"s_{l}"] + yb * ts
            tvu = tvm + tyc[:, None, :] * v1 + yc[:, None, :] * tv1
            ws.update({f"t_z1_{l}": tz1, f"t_h_{l}": th, f"t_xh_{l}": txh, f"t_x_msg{l}": txm, f"t_vec_msg{l}": tvm,
                       f"t_u_{l}": tu, f"t_s_{l}": ts, f"t_cat_{l}": tcat, f"t_zq_{l}": tzq, f"t_q_{l}": tq,
                       f"t_y_{l}": ty, f"t_x_upd{l}": txu, f"t_vec_upd{l}": tvu,
                       f"t_x_in{l + 1}": txu, f"t_vec_in{l + 1}": tvu})
            tx, tvec = txu, tvu
        tzo = tx @ self.P["out_energy.0.weight"].T
        ws["t_zo"] = tzo
        ws["t_e_atom"] = (dsilu(ws["zo"]) * tzo) @ self.P["out_energy.2.weight"][0]
        return ws["t_e_atom"].sum()

    # ------------------------------------------------------------------ 2./4. reverse
    def reverse(self, ge_atom, gte_atom=None):
        """dual=False (gte_atom None): force adjoint -> returns (gpos, None).
        dual=True: returns (None, grads dict)."""
        ws, F, P = self.ws, self.F, self.P
        dual = gte_atom is not None
        j, i = self.src, self.dst
        G = {k: torch.zeros_like(v) for k, v in P.items()} if dual else None
        zeros = lambda t: torch.zeros_like(t)

        # ---- readout
        zo, w2 = ws["zo"], P["out_energy.2.weight"][0]
        x_last = ws[f"x_in{self.L}"]
        gzo = ge_atom[:, None] * w2 * dsilu(zo)
        gtzo = None
        if dual:
            tzo = ws["t_zo"]
            gzo = gzo + gte_atom[:, None] * w2 * d2silu(zo) * tzo
            gtzo = gte_atom[:, None] * w2 * dsilu(zo)
            G["out_energy.2.weight"][0] = (ge_atom[:, None] * silu(zo) + gte_atom[:, None] * dsilu(zo) * tzo).sum(0)
            G["out_energy.2.bias"][0] = ge_atom.sum()
            G["out_energy.0.weight"] = gzo.T @ x_last + gtzo.T @ ws[f"t_x_in{self.L}"]
            G["out_energy.0.bias"] = gzo.sum(0)
        gx = gzo @ P["out_energy.0.weight"]
        gtx = gtzo @ P["out_energy.0.weight"] if dual else None
        gvec = zeros(ws["vec_in0"])
        gtvec = zeros(ws["vec_in0"]) if dual else None
        gd_tot = torch.zeros(self.E, dtype=gx.dtype)
        gr_tot = torch.zeros(self.E, 3, dtype=gx.dtype)

        for l in reversed(range(self.L)):
            m_, u_ = f"message_layers.{l}.", f"update_layers.{l}."
            U, V1, V2 = self._w(l, "U"), self._w(l, "V1"), self._w(l, "V2")
            W1, W2, Wr = self._w(l, "W1"), self._w(l, "W2"), self._w(l, "Wr")
            u, y, s, cat, zq, q = ws[f"u_{l}"], ws[f"y_{l}"], ws[f"s_{l}"], ws[f"cat_{l}"], ws[f"zq_{l}"], ws[f"q_{l}"]
            v1, v2, n = u[..., :F], u[..., F:], cat[:, F:]
            yb, yc = y[:, F:2 * F], y[:, 2 * F:]
            if dual:
                tu, ty, ts, tcat, tzq, tq = (ws[f"t_u_{l}"], ws[f"t_y_{l}"], ws[f"t_s_{l}"], ws[f"t_cat_{l}"],
                                             ws[f"t_zq_{l}"], ws[f"t_q_{l}"])
                tv1, tv2, tn = tu[..., :F], tu[..., F:], tcat[:, F:]
                tyb, tyc = ty[:, F:2 * F], ty[:, 2 * F:]
            # ---- update block reverse (inputs gx, gvec = adjoints of x_upd, vec_upd)
            gya = gx
            gyb = gx * s
            gs = gx * yb
            gyc = (gvec * v1).sum(1)
            gv1 = gvec * yc[:, None, :]
            if dual:
                gtya = gtx
                gyb = gyb + gtx * ts
                gtyb = gtx * s
                gs = gs + gtx * tyb
                gts = gtx * yb
                gyc = gyc + (gtvec * tv1).sum(1)
                gtyc = (gtvec * v1).sum(1)
                gv1 = gv1 + gtvec * tyc[:, None, :]
                gtv1 = gtvec * yc[:, None, :]
            gy = torch.cat([gya, gyb, gyc], -1)
            gq = gy @ V2
            gzq = gq * dsilu(zq)
            if dual:
                gty = torch.cat([gtya, gtyb, gtyc], -1)
                G[u_ + "xvec_proj.2.weight"] = gy.T @ q + gty.T @ tq
                G[u_ + "xvec_proj.2.bias"] = gy.sum(0)
                gtq = gty @ V2
                gzq = gzq + gtq * d2silu(zq) * tzq
                gtzq = gtq * dsilu(zq)
                G[u_ + "xvec_proj.0.weight"] = gzq.T @ cat + gtzq.T @ tcat
                G[u_ + "xvec_proj.0.bias"] = gzq.sum(0)
                gtcat = gtzq @ V1
                gtx = gtx + gtcat[:, :F]
                gtn = gtcat[:, F:]
            gcat = gzq @ V1
            gx = gx + gcat[:, :F]
            gn = gcat[:, F:]
            # n = sqrt(sum v2^2 + eps), s = sum v1 v2
            gv2 = (gn / n)[:, None, :] * v2 + gs[:, None, :] * v1
            gv1 = gv1 + gs[:, None, :] * v2
            if dual:
                gv2 = gv2 + (gtn / n)[:, None, :] * (tv2 - (tn / n)[:, None, :] * v2) + gts[:, None, :] * tv1
                gtv2 = (gtn / n)[:, None, :] * v2 + gts[:, None, :] * v1
                gv1 = gv1 + gts[:, None, :] * tv2
                gtv1 = gtv1 + gts[:, None, :] * v2
                gtu = torch.cat([gtv1, gtv2], -1)
            gu = torch.cat([gv1, gv2], -1)
            gvec = gvec + gu @ U
            if dual:
                G[u_ + "vec_proj.weight"] = (gu.reshape(-1, 2 * F).T @ ws[f"vec_msg{l}"].reshape(-1, F)
                                             + gtu.reshape(-1, 2 * F).T @ ws[f"t_vec_msg{l}"].reshape(-1, F))
                gtvec = gtvec + gtu @ U
            # ---- message block reverse (gx, gvec = adjoints of x_msg, vec_msg)
            vec, xh, phi, psi = ws[f"vec_in{l}"], ws[f"xh_{l}"], ws[f"phi_{l}"], ws[f"psi_{l}"]
            r = ws["r"]
            A, gma = gvec[i], gx[i]
            m = xh[j] * phi
            mb, mc = m[:, F:2 * F], m[:, 2 * F:]
            gmb = (A * vec[j]).sum(1)
            gmc = (A * r[:, :, None]).sum(1)
            gvec_src = A * mb[:, None, :]
            if dual:
                T, gtma = gtvec[i], gtx[i]
                t_d, t_r, tvec, txh = ws["t_d"], ws["t_r"], ws[f"t_vec_in{l}"], ws[f"t_xh_{l}"]
                tphi = psi * t_d[:, None]
                tm = txh[j] * phi + xh[j] * tphi
                tmb = tm[:, F:2 * F]
                gmb = gmb + (T * tvec[j]).sum(1)
                gtmb = (T * vec[j]).sum(1)
                gmc = gmc + (T * t_r[:, :, None]).sum(1)
                gtmc = (T * r[:, :, None]).sum(1)
                gvec_src = gvec_src + T * tmb[:, None, :]
                gtvec_src = T * mb[:, None, :]
                gtm = torch.cat([gtma, gtmb, gtmc], -1)
            gm = torch.cat([gma, gmb, gmc], -1)
            gxh_e = gm * phi
            gphi = gm * xh[j]
            if dual:
                gxh_e = gxh_e + gtm * tphi
                gtxh_e = gtm * phi
                gphi = gphi + gtm * txh[j]
                gpsi = gtm * xh[j] * t_d[:, None]
                G[m_ + "rbf_proj.weight"] = gphi.T @ ws["rho"] + gpsi.T @ ws["drho"]
                G[m_ + "rbf_proj.bias"] = (gphi * ws["beta"][:, None] + gpsi * ws["dbeta"][:, None]).sum(0)
                gtxh = self._scatter(gtxh_e, j, self.N)
                gtvec = gtvec + self._scatter(gtvec_src, j, self.N)
            else:
                gd_tot = gd_tot + (gphi * psi).sum(-1)
                gr_tot = gr_tot + (A * mc[:, None, :]).sum(-1)
            gxh = self._scatter(gxh_e, j, self.N)
            gvec = gvec + self._scatter(gvec_src, j, self.N)
            gh = gxh @ W2
            gz1 = gh * dsilu(ws[f"z1_{l}"])
            if dual:
                G[m_ + "x_proj.2.weight"] = gxh.T @ ws[f"h_{l}"] + gtxh.T @ ws[f"t_h_{l}"]
                G[m_ + "x_proj.2.bias"] = gxh.sum(0)
                gth = gtxh @ W2
                gz1 = gz1 + gth * d2silu(ws[f"z1_{l}"]) * ws[f"t_z1_{l}"]
                gtz1 = gth * dsilu(ws[f"z1_{l}"])
                G[m_ + "x_proj.0.weight"] = gz1.T @ ws[f"x_in{l}"] + gtz1.T @ ws[f"t_x_in{l}"]
                G[m_ + "x_proj.0.bias"] = gz1.sum(0)
                gtx = gtx + gtz1 @ W1
            gx = gx + gz1 @ W1
            ws[f"g_x_in{l}"], ws[f"g_vec_in{l}"] = gx, gvec
            if dual:
                ws[f"gt_x_in{l}"], ws[f"gt_vec_in{l}"] = gtx, gtvec

        if dual:
            G["atom_emb.embeddings.weight"] = self._scatter(gx, self.z - 1, self.cfg.num_elements)
            return None, G
        # ---- geometry reverse: d = |w|, r = w / d, w = pos[j] - pos[i]
        ws["gd"], ws["gr"] = gd_tot, gr_tot
        r, d = ws["r"], ws["d"]
        gw = gd_tot[:, None] * r + (gr_tot - (gr_tot * r).sum(-1, keepdim=True) * r) / d[:, None]
        gpos = self._scatter(gw, j, self.N) - self._scatter(gw, i, self.N)
        return gpos, None

    # ------------------------------------------------------------------ whole step
    def energy_forces(self):
        energy = self.forward()
        gpos, _ = self.reverse(torch.ones(self.N, dtype=energy.dtype))
        self.ws["forces"] = -gpos
        return energy, -gpos

    def backward(self, g_energy, g_forces):
        """Given dL/dE_b and dL/dF_i returns dL/dtheta (dict)."""
        edot = self.tangent(-g_forces)
        _, G = self.reverse(g_energy[self.batch], torch.ones(self.N, dtype=g_forces.dtype))
        self.ws["edot"] = edot
        return G


def loss_and_seeds(energy, forces, y, f_target, coef_e=1.0, coef_f=1.0):
    """L = coef_e mean|E-y| + coef_f mean_i ||F_i - Ft_i||_2 and its gradient wrt (E, F)."""
    B, N = energy.shape[0], forces.shape[0]
    diff = forces - f_target
    nrm = torch.linalg.vector_norm(diff, dim=-1)
    loss = coef_e * (energy - y).abs().mean() + coef_f * nrm.mean()
    gE = coef_e * torch.sign(energy - y) / B
    gF = coef_f * diff / (nrm[:, None] * N)
    gF = torch.where(nrm[:, None] > 0, gF, torch.zeros_like(gF))
    return loss, gE, gF
