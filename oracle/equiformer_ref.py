"""TEST INFRASTRUCTURE -- CPU restatement (torch, any float dtype) of the reference EquiformerV2 forward pass in eval mode
(nablaDFT/equiformer_v2/equiformer_v2_oc20.py:487-586, transformer_block.py, so2_ops.py, so3.py, layer_norm.py, input_block.py), pinned to the golden vectors the
REAL reference classes produced (tests/golden/equiformer_*.npz); checker of the CPU tests, ``__graft_entry__.smoke()`` and the EquiformerV2 bench's
``cpu_baseline``.  The product never imports this file.  Harmonics / S2 grids come from oracle/e3nn_mini.py (e3nn restated: parity unpinned for those)."""
import math

import numpy as np
import torch

from oracle import e3nn_mini as M
from oracle.escn_ref import frames, j_matrices, radius_graph, s2, wigner

AVG_NUM_NODES = 39.65745326960467          # equiformer_v2_oc20.py:47-48
AVG_DEGREE = 19.16009564536883


def grid(lmax, mmax, dtype):
    """(to_grid, from_grid) [points, kept coefficients] of SO3_Grid(lmax, mmax, normalization="component") (so3.py:367-429), l-primary."""
    T, F = s2(lmax, mmax, torch.float64)
    f = M._s2_degree_factor(lmax, "component")
    T, F = T * f, F / f
    if lmax != mmax:
        r = torch.cat([torch.full((2 * l + 1,), math.sqrt((2 * l + 1) / (2 * mmax + 1)) if l > mmax else 1.0, dtype=torch.float64) for l in range(lmax + 1)])
        T, F = T * r, F * r
    keep = [l * l + l + m for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]
    return T[:, keep].to(dtype), F[:, keep].to(dtype)


def forward(P, cfg, pos, z, sizes, rot=None, record=None):
    """P: state dict (reference names); returns (energy [B], forces [N, 3])."""
    dt = pos.dtype
    lmax, mmax, C = cfg["lmax_list"][0], cfg["mmax_list"][0], cfg["sphere_channels"]
    Hc, H, A, V = cfg["attn_hidden_channels"], cfg["num_heads"], cfg["attn_alpha_channels"], cfg["attn_value_channels"]
    nf = (lmax + 1) ** 2
    N, B = pos.shape[0], len(sizes)
    src, dst = radius_graph(pos, sizes, cfg["max_radius"], cfg["max_neighbors"])
    E = src.shape[0]
    vec = pos[src] - pos[dst]
    dist = vec.norm(dim=-1)
    rot = frames(vec) if rot is None else rot
    W = wigner(rot, lmax, j_matrices(lmax))
    lm = [(l, m) for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]
    red = [l * l + l + m for l, m in lm]                                           # coefficient_idx(lmax, mmax)
    mprim, msize = [lm.index((l, 0)) for l in range(lmax + 1)], [lmax + 1]
    for m in range(1, mmax + 1):
        mprim += [lm.index((l, m)) for l in range(m, lmax + 1)] + [lm.index((l, -m)) for l in range(m, lmax + 1)]
        msize.append(lmax - m + 1)
    inv = torch.empty(len(mprim), dtype=torch.long)
    inv[torch.tensor(mprim)] = torch.arange(len(mprim))
    Wr = W[:, red, :]
    rescale = torch.cat([torch.full((2 * l + 1,), math.sqrt((2 * l + 1) / (2 * mmax + 1)) if l > mmax else 1.0, dtype=dt) for l in range(lmax + 1)])
    Winv = Wr.transpose(1, 2) * rescale.view(1, -1, 1)                              # SO3_Rotation.rotate_inv (so3.py:338-343)
    Tr, Fr = grid(lmax, mmax, dt)
    Tf, Ff = grid(lmax, lmax, dt)
    act = torch.nn.functional.silu
    lin = lambda k, x: x @ P[k + ".weight"].T + (P[k + ".bias"] if (k + ".bias") in P else 0)                           # noqa: E731
    lnorm = lambda k, x: torch.nn.functional.layer_norm(x, (x.shape[-1],), P[k + ".weight"], P[k + ".bias"], 1e-5)      # noqa: E731
    off = torch.linspace(0.0, cfg["max_radius"], 600, dtype=dt)                    # GaussianSmearing(0, cutoff, 600, 2.0) in the dtype of the run (smearing.py:14-26)
    coeff = -0.5 / (2.0 * float(off[1] - off[0])) ** 2
    x_dist = torch.exp(coeff * (dist[:, None] - off[None, :]) ** 2)
    expand = torch.tensor([l for l in range(lmax + 1) for _ in range(2 * l + 1)])
    bw = torch.cat([torch.full((2 * l + 1,), 1.0 / (2 * l + 1) / lmax, dtype=dt) for l in range(1, lmax + 1)])

    def rad(k, xe):
        h = act(lnorm(k + ".net.1", lin(k + ".net.0", xe)))
        h = act(lnorm(k + ".net.4", lin(k + ".net.3", h)))
        return lin(k + ".net.6", h)

    def edge_scalars(k):
        return torch.cat([x_dist, P[k + ".source_embedding.weight"][z[src]], P[k + ".target_embedding.weight"][z[dst]]], dim=1)

    def so3_linear(k, x):
        out = torch.einsum("bmi,moi->bmo", x, P[k + ".weight"][expand])
        out[:, 0] = out[:, 0] + P[k + ".bias"]
        return out

    def norm_sh(k, x):                                                   # layer_norm.py:169-215
        out0 = lnorm(k + ".norm_l0", x[:, 0:1])
        fn = (x[:, 1:] ** 2 * bw.view(1, -1, 1)).sum(1, keepdim=True).mean(2, keepdim=True)
        s = (fn + 1e-5) ** -0.5
        return torch.cat([out0, x[:, 1:] * s * P[k + ".affine_weight"][expand[1:] - 1].unsqueeze(0)], dim=1)

    def so2(k, x, xe, extra):                                            # x [E, n_red, c] l-primary -> same (so2_ops.py:127-193)
        xm = x[:, mprim]
        w = rad(k + ".rad_func", xe) if (k + ".rad_func.net.0.weight") in P else None
        x0 = xm[:, :msize[0]].reshape(E, -1)
        o_r = 0
        if w is not None:
            x0 = x0 * w[:, :x0.shape[1]]
            o_r = x0.shape[1]
        x0 = lin(k + ".fc_m0", x0)
        x0_extra = None
        if extra:
            x0_extra, x0 = x0[:, :extra], x0[:, extra:]
        cout = x0.shape[1] // msize[0]
        out = [x0.reshape(E, -1, cout)]
        o = msize[0]
        for m in range(1, mmax + 1):
            blk = xm[:, o:o + 2 * msize[m]].reshape(E, 2, -1)
            if w is not None:
                blk = blk * w[:, o_r:o_r + blk.shape[2]].unsqueeze(1)
                o_r += blk.shape[2]
            y = lin(f"{k}.so2_m_conv.{m - 1}.fc", blk)
            half = y.shape[2] // 2
            xr, xi = y[..., :half], y[..., half:]
            out.append(torch.stack([xr[:, 0] - xi[:, 1], xr[:, 1] + xi[:, 0]], dim=1).reshape(E, -1, cout))
            o += 2 * msize[m]
        return torch.cat(out, dim=1)[:, inv], x0_extra

    def attention(k, x):                                                 # transformer_block.py:194-384
        xe = edge_scalars(k)
        msg = torch.bmm(Wr, torch.cat([x[src], x[dst]], dim=2))
        msg, extra = so2(k + ".so2_conv_1", msg, xe, H * A + Hc)
        xa = lnorm(k + ".alpha_norm", extra[:, :H * A].reshape(E, H, A))
        xa = 0.6 * xa + 0.4 * xa * (2 * torch.sigmoid(xa) - 1)                                       # SmoothLeakyReLU(0.2), activation.py:52-61
        logit = (xa * P[k + ".alpha_dot"].unsqueeze(0)).sum(-1)
        mx = torch.full((N, H), -float("inf"), dtype=dt).scatter_reduce(0, dst.view(-1, 1).expand(E, H), logit, "amax", include_self=True)
        ex = (logit - mx[dst]).exp()
        alpha = ex / (torch.zeros(N, H, dtype=dt).index_add_(0, dst, ex)[dst] + 1e-16)
        g = torch.einsum("gi,egc->eic", Fr, act(torch.einsum("gi,eic->egc", Tr, msg)))                 # S2Activation on the (lmax, mmax) grid
        msg = torch.cat([act(extra[:, H * A:]).unsqueeze(1), g[:, 1:]], dim=1)                       # SeparableS2Activation
        msg, _ = so2(k + ".so2_conv_2", msg, None, 0)
        msg = (msg.reshape(E, -1, H, V) * alpha.view(E, 1, H, 1)).reshape(E, -1, H * V)
        y = x.new_zeros(N, nf, H * V).index_add_(0, dst, torch.bmm(Winv, msg))
        return so3_linear(k + ".proj", y)

    def ffn(k, x):                                                       # transformer_block.py:472-507
        gating = act(lin(k + ".scalar_mlp.0", x[:, 0:1]))
        h = so3_linear(k + ".so3_linear_1", x)
        g = torch.einsum("gi,nic->ngc", Tf, h)
        g = lin(k + ".grid_mlp.4", act(lin(k + ".grid_mlp.2", act(lin(k + ".grid_mlp.0", g)))))
        h = torch.einsum("gi,ngc->nic", Ff, g)
        return so3_linear(k + ".so3_linear_2", torch.cat([gating, h[:, 1:]], dim=1))

    x = pos.new_zeros(N, nf, C)
    x[:, 0] = P["sphere_embedding.weight"][z]
    # edge-degree embedding (input_block.py:81-117)
    k = "edge_degree_embedding"
    m0 = rad(k + ".rad_func", edge_scalars(k)).reshape(E, lmax + 1, C)
    pad = torch.cat([m0, m0.new_zeros(E, len(red) - (lmax + 1), C)], dim=1)[:, inv]
    x = x + x.new_zeros(N, nf, C).index_add_(0, dst, torch.bmm(Winv, pad)) / AVG_DEGREE
    if record is not None:
        record["embed"] = x
    for i in range(cfg["num_layers"]):
        k = f"blocks.{i}"
        h = norm_sh(k + ".norm_1", x)
        if record is not None and i == 0:
            record["norm1"] = h
        h = attention(k + ".ga", h)
        if record is not None and i == 0:
            record["ga"] = h
        x = x + h
        x = x + ffn(k + ".ffn", norm_sh(k + ".norm_2", x))
        if record is not None:
            record[f"block{i}"] = x
    x = norm_sh("norm", x)
    e = ffn("energy_block", x)[:, 0, 0]
    batch = torch.repeat_interleave(torch.arange(B), torch.as_tensor(np.asarray(sizes)))
    energy = e.new_zeros(B).index_add_(0, batch, e) / AVG_NUM_NODES
    forces = attention("force_block", x)[:, 1:4, 0]
    return energy, forces


def loss(E, F, y, f_target):
    return 2.0 * (E - y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - f_target, dim=-1).mean()    # config/model/equiformer_v2_oc20.yaml:57-64
