"""TEST INFRASTRUCTURE -- CPU restatement of the QHNet forward pass, functional, any dtype.

Pinned (tests/test_qhnet_cpu.py): against tests/golden/qhnet_small.npz / qhnet_full.npz, which are written by the REAL reference classes
(/root/reference/nablaDFT/qhnet/qhnet.py, layers.py) running on oracle/e3nn_mini.py -- every layer output, the blocks, H, the loss and all
gradients.  The e3nn 0.5.1 arithmetic underneath (3j tensors, path normalisation, Linear / FullyConnectedNet scaling; oracle/e3nn_mini.py) is
third-party and *parity unpinned*.  Used as the checker of the GPU tests / smoke() and as bench.py's ``cpu_baseline`` (kind "port").

Works on a state_dict with the reference's parameter names.  Features are [rows, 25, C] (component l*l + m + l); e3nn's [mul, 2l+1] layout
only matters for how flat weight vectors are indexed, which is spelled out where it happens.

  radius_graph / build_graph        qhnet.py:254-291 (torch_cluster semantics: strict d < r, per centre ascending neighbours)
  rbf                               layers.py:86-120
  norm_gate                         layers.py:123-147
  invariants                        layers.py:236-258, 466-476 (+ InnerProduct :277-294)
  conv_layer / conv_net_layer       layers.py:234-274, 338-343
  self_net_layer, pair_net_layer    layers.py:565-578, 465-492
  expansion                         layers.py:598-662
  forward                           qhnet.py:186-252 (build_final_matrix :293-321 through oracle/hblock_ref.py)
"""
import math

import torch

from oracle import e3nn_mini as e3
from oracle import hblock_ref

LMAX = 4
ALL_PATHS = [(l1, l2, L) for l1 in range(LMAX + 1) for l2 in range(LMAX + 1) for L in range(abs(l1 - l2), min(l1 + l2, LMAX) + 1)]
_SSP_CST = None


def ssp_cst():
    global _SSP_CST
    if _SSP_CST is None:
        _SSP_CST = e3.normalize2mom_constant(lambda t: torch.nn.functional.softplus(t) - math.log(2.0))
    return _SSP_CST


def sl(l):
    return slice(l * l, (l + 1) * (l + 1))


def radius_graph(pos, ptr, r):
    """(dst, src): row 0 / row 1 of the reference's edge_index; centres (src) ascending, their neighbours (dst) ascending; strict d^2 < r^2."""
    dst, src = [], []
    for b in range(len(ptr) - 1):
        a0, a1 = int(ptr[b]), int(ptr[b + 1])
        p = pos[a0:a1]
        d2 = (p[:, None] - p[None]).pow(2).sum(-1)
        adj = (d2 < r * r) & ~torch.eye(a1 - a0, dtype=torch.bool)
        c, j = adj.nonzero(as_tuple=True)
        dst.append(j + a0), src.append(c + a0)
    return torch.cat(dst), torch.cat(src)


def rbf(P, r, cutoff, K):
    """ExponentialBernsteinRadialBasisFunctions.forward (layers.py:115-120) with the buffers of __init__ (:97-108)."""
    dt = r.dtype
    logfact = torch.zeros(K, dtype=torch.float64)
    for i in range(2, K):
        logfact[i] = logfact[i - 1] + math.log(i)
    v = torch.arange(K, dtype=torch.float64)
    n = (K - 1) - v
    logc = (logfact[-1] - logfact[v.long()] - logfact[n.long()]).float().to(dt)       # float32 buffers in the reference
    n, v = n.float().to(dt), v.float().to(dt)
    alpha = torch.nn.functional.softplus(P["distance_expansion._alpha"].to(dt))
    x = -alpha * r[:, None]
    x = logc + n * x + v * torch.log(-torch.expm1(x))
    rr = r[:, None]
    z = torch.zeros_like(rr)
    x_ = torch.where(rr < cutoff, rr, z)
    fc = torch.where(rr < cutoff, torch.exp(-(x_ ** 2) / ((cutoff - x_) * (cutoff + x_))), z)
    return fc * torch.exp(x)


def sph(vec):
    """o3.spherical_harmonics(l <= 4, vec[:, [1, 2, 0]], normalize=True, 'component') (qhnet.py:266-271)."""
    return e3.spherical_harmonics(e3.Irreps.spherical_harmonics(LMAX), vec[:, [1, 2, 0]], True, "component")


def o3_linear(P, name, x, c_in, c_out):
    """e3nn Linear: weight = [W_0, W_1, ...] flattened, W_l [c_in, c_out]; y_l = x_l W_l / sqrt(c_in); bias on l = 0."""
    W = P[name + ".weight"].to(x.dtype).view(-1, c_in, c_out)
    ys = [torch.einsum("rmu,uw->rmw", x[:, sl(l)], W[l]) / math.sqrt(c_in) for l in range(W.shape[0])]
    y = torch.cat(ys, dim=1)
    if name + ".bias" in P and P[name + ".bias"].numel():
        y = torch.cat([y[:, :1] + P[name + ".bias"].to(x.dtype), y[:, 1:]], dim=1)
    return y


def fc_net(P, name, x, hs):
    """e3nn FullyConnectedNet([h0, h1, h2], ssp)."""
    h = x @ (P[name + ".layer0.weight"].to(x.dtype) / math.sqrt(hs[0]))
    h = (torch.nn.functional.softplus(h) - math.log(2.0)) * ssp_cst()
    return h @ (P[name + ".layer1.weight"].to(x.dtype) / math.sqrt(hs[1]))


def mlp(P, name, x):
    h = torch.nn.functional.silu(x @ P[name + ".0.weight"].to(x.dtype).T + P[name + ".0.bias"].to(x.dtype))
    return h @ P[name + ".2.weight"].to(x.dtype).T + P[name + ".2.bias"].to(x.dtype)


def norm_gate(P, name, x):
    C = x.shape[-1]
    norms = [x[:, sl(l)].pow(2).sum(1).relu().sqrt() for l in range(1, LMAX + 1)]
    gates = mlp(P, name + ".fc", torch.cat([x[:, 0]] + norms, dim=-1))
    out = [gates[:, None, :C]] + [x[:, sl(l)] * gates[:, None, l * C:(l + 1) * C] for l in range(1, LMAX + 1)]
    return torch.cat(out, dim=1)


def invariants(x, dst, src, second_from_src):
    C = x.shape[-1]
    parts = [x[dst, 0], x[src, 0] if second_from_src else x[dst, 0]]
    if x.shape[1] > 1:
        parts += [(x[dst][:, sl(l)] * x[src][:, sl(l)]).sum(1) / (2 * l + 1) for l in range(1, LMAX + 1)]
    return torch.cat(parts, dim=-1)


def path_coefficients(paths):
    """sqrt(alpha) of e3nn's TensorProduct with QHNet's path weights (get_feasible_irrep, layers.py:60-82)."""
    cnt = {}
    for (_, _, lo) in paths:
        cnt[lo] = cnt.get(lo, 0) + 1
    return [math.sqrt((2 * lo + 1) / cnt[lo] * math.sqrt((2 * lo + 1) / len(paths))) for (_, _, lo) in paths]


def tensor_product(paths, x1, x2, w, x2_has_channels):
    """sum over paths of coef * w[r, path, u] * sum_{ij} 3j[i,j,k] x1[r,i,u] x2[r,j,(u)]; w [rows or 1, n_paths, C]."""
    C = x1.shape[-1]
    out = x1.new_zeros(x1.shape[0], 25, C)
    coef = path_coefficients(paths)
    for p, (l1, l2, lo) in enumerate(paths):
        w3j = e3.wigner_3j(l1, l2, lo, dtype=x1.dtype)
        if x2_has_channels:
            t = torch.einsum("ijk,riu,rju->rku", w3j, x1[:, sl(l1)], x2[:, sl(l2)])
        else:
            t = torch.einsum("ijk,riu,rj->rku", w3j, x1[:, sl(l1)], x2[:, sl(l2)])
        out[:, sl(lo)] = out[:, sl(lo)] + coef[p] * w[:, p, None, :] * t
    return out


def conv_layer(P, name, g, x, first, C, K):
    paths = [p for p in ALL_PATHS if sum(p) % 2 == 0 and (p[0] == 0 or not first)]
    dst, src = g["dst"], g["src"]
    if not first:
        pre_x = o3_linear(P, name + ".linear_node_pre", x, C, C)
        s0 = invariants(pre_x, dst, src, False)
        x = o3_linear(P, name + ".linear_node", norm_gate(P, name + ".norm_gate", x), C, C)
    else:
        s0 = invariants(x, dst, src, False)
    w = fc_net(P, name + ".fc_node", g["rbf"], [K, 32, len(paths) * C]) * fc_net(P, name + ".layer_l0", s0, [s0.shape[1], 32, len(paths) * C])
    x1 = x[src]
    if first:
        x1 = torch.cat([x1, x1.new_zeros(x1.shape[0], 24, C)], dim=1)
    msg = tensor_product(paths, x1, g["sh"], w.view(-1, len(paths), C), False)
    out = torch.zeros(x.shape[0], 25, C, dtype=x.dtype).index_add_(0, dst, msg)
    if not first:
        out = out + x
    return o3_linear(P, name + ".linear_out", out, C, C)


def self_net_layer(P, name, x, old, C):
    xl = o3_linear(P, name + ".linear_node_1", norm_gate(P, name + ".norm_gate_1", x), C, C)
    xr = o3_linear(P, name + ".linear_node_2", norm_gate(P, name + ".norm_gate_2", x), C, C)
    t = tensor_product(ALL_PATHS, xl, xr, P[name + ".tp.weight"].to(x.dtype).view(1, len(ALL_PATHS), C), True) + x
    t = o3_linear(P, name + ".linear_node_3", norm_gate(P, name + ".norm_gate", t), C, C)
    return t if old is None else old + t


def pair_net_layer(P, name, g, x, old, C, K):
    dst, src = g["full_dst"], g["full_src"]
    a0 = o3_linear(P, name + ".linear_node_pair_inner", x, C, C)
    s0 = invariants(a0, dst, src, True)
    xn = o3_linear(P, name + ".linear_node_pair_n", norm_gate(P, name + ".norm_gate_pre", x), C, C)
    w = fc_net(P, name + ".fc_node_pair", g["full_rbf"], [K, C, len(ALL_PATHS) * C]) * mlp(P, name + ".fc", s0)
    t = tensor_product(ALL_PATHS, xn[src], xn[dst], w.view(-1, len(ALL_PATHS), C), True)
    t = o3_linear(P, name + ".linear_node_pair", norm_gate(P, name + ".norm_gate", t), C, C)
    return t if old is None else t + old


def expansion(x, W, b, counts):
    """Expansion.forward (layers.py:598-662): x [R, 25, Cb], W [R, nw], b [R, nb] -> [R, S, S]."""
    R, _, Cb = x.shape
    S = counts[0] + 3 * counts[1] + 5 * counts[2]
    roff = [0, counts[0], counts[0] + 3 * counts[1]]
    out = x.new_zeros(R, S, S)
    wo = bo = 0
    for li in range(LMAX + 1):
        for l1 in range(3):
            for l2 in range(3):
                if not abs(l1 - l2) <= li <= l1 + l2:
                    continue
                n1, n2 = counts[l1], counts[l2]
                w = W[:, wo:wo + Cb * n1 * n2].view(R, Cb, n1, n2)
                wo += Cb * n1 * n2
                res = torch.einsum("bwuv,bkw->buvk", w, x[:, sl(li)])
                if li == 0:
                    res = res + b[:, bo:bo + n1 * n2].view(R, n1, n2, 1)
                    bo += n1 * n2
                w3j = e3.wigner_3j(l1, l2, li).to(x.dtype)      # layers.py:617: default-dtype (float32) tensor cast to the feature dtype
                blk = torch.einsum("ijk,buvk->buivj", w3j, res) / Cb
                d1, d2 = 2 * l1 + 1, 2 * l2 + 1
                out[:, roff[l1]:roff[l1] + n1 * d1, roff[l2]:roff[l2] + n2 * d2] += blk.reshape(R, n1 * d1, n2 * d2)
    return out


def forward(P, cfg, orbitals, pos, z, ptr, keep=None):
    """-> dense block-diagonal H (qhnet.py:233-238).  ``keep``: optional dict that receives the intermediates."""
    C, Cb, K, nl = cfg["hidden_size"], cfg["bottle_hidden_size"], cfg["radius_embed_dim"], cfg["num_gnn_layers"]
    dt = pos.dtype
    g = {}
    g["dst"], g["src"] = radius_graph(pos, ptr, float(cfg["max_radius"]))
    vec = pos[g["dst"]] - pos[g["src"]]
    g["rbf"], g["sh"] = rbf(P, vec.norm(dim=-1), float(cfg["max_radius"]), K), sph(vec)
    g["full_dst"], g["full_src"] = radius_graph(pos, ptr, 10000.0)
    fvec = pos[g["full_dst"]] - pos[g["full_src"]]
    g["full_rbf"] = rbf(P, fvec.norm(dim=-1), float(cfg["max_radius"]), K)
    node_attr = P["node_embedding.weight"].to(dt)[z]
    x = node_attr[:, None, :]
    fii = fij = None
    for i in range(nl):
        y = conv_layer(P, f"e3_gnn_layer.{i}.conv", g, x, i == 0, C, K)
        x = y if i == 0 else x + y
        if keep is not None:
            keep[f"conv{i}"] = x
        if i > 2:
            fii = self_net_layer(P, f"e3_gnn_node_layer.{i - 3}", x, fii, C)
            fij = pair_net_layer(P, f"e3_gnn_node_pair_layer.{i - 3}", g, x, fij, C, K)
            if keep is not None:
                keep[f"self{i - 3}"], keep[f"pair{i - 3}"] = fii, fij
    fii, fij = o3_linear(P, "output_ii", fii, C, Cb), o3_linear(P, "output_ij", fij, C, Cb)
    masks, s_max, p_max, d_max = hblock_ref.orbital_masks(orbitals)
    counts = (s_max, p_max, d_max)
    diag = expansion(fii, mlp(P, "fc_ii.hamiltonian", node_attr), mlp(P, "fc_ii_bias.hamiltonian", node_attr), counts)
    pe = torch.cat([node_attr[g["full_dst"]], node_attr[g["full_src"]]], dim=-1)
    nondiag = expansion(fij, mlp(P, "fc_ij.hamiltonian", pe), mlp(P, "fc_ij_bias.hamiltonian", pe), counts)
    if keep is not None:
        keep["diag_blocks"], keep["nondiag_blocks"], keep["graph"] = diag, nondiag, g
    return hblock_ref.build_final_matrix(z, ptr, torch.stack([g["full_dst"], g["full_src"]]), masks, diag, nondiag, symmetrize=True)


def hamiltonian_loss(pred, target, mask):
    """qhnet/loss.py:9-16."""
    diff = pred - target
    mse = torch.mean(diff ** 2)
    mae = torch.mean(torch.abs(diff))
    return (mse * (mask.numel() / mask.sum())).sqrt() + mae * (mask.numel() / mask.sum())
