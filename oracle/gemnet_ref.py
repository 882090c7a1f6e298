"""TEST INFRASTRUCTURE -- CPU restatement (torch, any float dtype) of the reference GemNet-OC forward pass, pinned to the golden vectors the REAL
reference classes produced (tests/golden/gemnet_*.npz, oracle/make_golden_gemnet.py); used by the CPU tests, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of the GemNet-OC bench.  The product never imports this file.

Written from the mathematics of nablaDFT/gemnet_oc (file:line below), not from its code structure: graphs are plain Python / numpy loops over molecules,
interactions are explicit sums over (out edge, in edge) lists instead of zero-padded tiles.
  graphs        gemnet_oc.py:777-958 (sub-graphs by cutoff and K nearest, utils.py:408-500; symmetrised main graph :694-775)
  index lists   interaction_indices.py:13-282
  bases         layers/radial_basis.py:21-37,60-77,196-220; layers/basis.py:215-295 (Y_l0); gemnet_oc.py:597-655 (dihedral half angle)
  blocks        layers/interaction_block.py, layers/atom_update_block.py, layers/embedding_block.py, layers/efficient.py, layers/base_layers.py
  outputs       gemnet_oc.py:1198-1243
Only the options of config/model/gemnet-oc.yaml (gaussian rbf, polynomial envelope, spherical_harmonics cbf, legendre_outer sbf, direct coupled forces).
"""
import math

import numpy as np
import torch

INV_SQRT2 = 1.0 / math.sqrt(2.0)


# ---- graphs ------------------------------------------------------------------------------------------------------------------------------------
def _norm(v):
    return v.norm(dim=-1)                                           # the reference's distance (gemnet_oc.py:1325)


def build_graphs(pos, sizes, cfg):
    """Edge lists in the reference's order.  pos: float tensor [N, 3]; returns dict name -> (src, dst) int64 arrays, plus id_swap for "main"."""
    N = pos.shape[0]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    src, dst = [], []
    p32 = pos.to(torch.float32)
    r2 = float(cfg["cutoff_aint"]) ** 2
    for b in range(len(sizes)):                                     # radius_graph: target ascending, sources ascending, d^2 < r^2, no self loops
        a0, a1 = starts[b], starts[b + 1]
        d2 = ((p32[a0:a1, None, :] - p32[None, a0:a1, :]) ** 2).sum(-1).numpy()
        for i in range(a1 - a0):
            for j in range(a1 - a0):
                if i != j and d2[i, j] < r2:
                    src.append(a0 + j); dst.append(a0 + i)
    src, dst = np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64)
    dist = _norm(pos[src] - pos[dst]).numpy()

    def select(cutoff, K):
        keep = np.zeros(len(src), dtype=bool)
        for i in range(N):
            rows = np.nonzero(dst == i)[0]
            rows = rows[dist[rows] <= cutoff]
            order = np.argsort(dist[rows], kind="stable")[:K]
            keep[rows[order]] = True
        return keep

    G = {"a2a": (src, dst)}
    km = select(cfg["cutoff"], cfg["max_neighbors"])
    ka = select(cfg["cutoff_aeaint"], cfg["max_neighbors_aeaint"])
    kq = select(cfg["cutoff_qint"], cfg["max_neighbors_qint"])
    G["a2ee2a"] = (src[ka], dst[ka])
    G["qint"] = (src[kq], dst[kq])
    ms, md = [], []
    for b in range(len(sizes)):                                     # keep source < target, then the flips, molecule by molecule (gemnet_oc.py:712-750)
        sel = km & (src < dst) & (dst >= starts[b]) & (dst < starts[b + 1])
        ms += list(src[sel]) + list(dst[sel]); md += list(dst[sel]) + list(src[sel])
    ms, md = np.array(ms, dtype=np.int64), np.array(md, dtype=np.int64)
    G["main"] = (ms, md)
    lookup = {(int(s), int(t)): e for e, (s, t) in enumerate(zip(ms, md))}
    G["id_swap"] = np.array([lookup[(int(t), int(s))] for s, t in zip(ms, md)], dtype=np.int64)
    return G


def _rows_by_target(dst, src, N):
    rows = [[] for _ in range(N)]
    for e in np.lexsort((src, dst)):
        rows[int(dst[e])].append(int(e))
    return rows


def triplets(out, inn, N):
    """(in edge, out edge) pairs sharing the target atom, sources different; out ascending, in by source (interaction_indices.py:13-118)."""
    rows = _rows_by_target(inn[1], inn[0], N)
    i_in, i_out = [], []
    for o in range(len(out[0])):
        for e in rows[int(out[1][o])]:
            if inn[0][e] != out[0][o]:
                i_in.append(e); i_out.append(o)
    return np.array(i_in, dtype=np.int64), np.array(i_out, dtype=np.int64)


def quadruplets(main, qint, N):
    """c -> a <- b <- d: lists (out edge c->a, qint edge b->a, main edge d->b), b != c, d not in {a, c} (interaction_indices.py:121-282)."""
    mrows, qrows = _rows_by_target(main[1], main[0], N), _rows_by_target(qint[1], qint[0], N)
    o_, q_, p_ = [], [], []
    for o in range(len(main[0])):
        c, a = int(main[0][o]), int(main[1][o])
        for q in qrows[a]:
            b = int(qint[0][q])
            if b == c:
                continue
            for p in mrows[b]:
                d = int(main[0][p])
                if d != a and d != c:
                    o_.append(o); q_.append(q); p_.append(p)
    return np.array(o_, dtype=np.int64), np.array(q_, dtype=np.int64), np.array(p_, dtype=np.int64)


# ---- bases -------------------------------------------------------------------------------------------------------------------------------------
def radial_basis(d, cutoff, offset, exponent, scale):
    ds = d / cutoff
    p = float(exponent)
    env = 1 - (p + 1) * (p + 2) / 2 * ds ** p + p * (p + 2) * ds ** (p + 1) - p * (p + 1) / 2 * ds ** (p + 2)
    env = torch.where(ds < 1, env, torch.zeros_like(ds))
    width = 1.0 / (offset.numel() - 1)
    return env[:, None] * torch.exp(-0.5 / width ** 2 * (ds[:, None] - offset[None, :]) ** 2) * scale


def zonal(z, ns):
    """Y_l0(z) = sqrt((2l+1)/(4 pi)) P_l(z), l < ns."""
    P = [torch.ones_like(z), z]
    for l in range(2, ns):
        P.append(((2 * l - 1) * z * P[-1] - (l - 1) * P[-2]) / l)
    return torch.stack([math.sqrt((2 * l + 1) / (4 * math.pi)) * P[l] for l in range(ns)], dim=1)


def _sf(P, key):
    v = float(P[key])
    return v if v != 0.0 else 1.0                                   # ScaleFactor: 0 = not fitted = identity (scale_factor.py:66-68,146-149)


# ---- layers ------------------------------------------------------------------------------------------------------------------------------------
def ssilu(x):
    return torch.nn.functional.silu(x) / 0.6


def dense(P, key, x, act=False):
    y = x @ P[key + ".linear.weight"].T
    return ssilu(y) if act else y


def residual(P, key, x):
    y = dense(P, key + ".dense_mlp.1", dense(P, key + ".dense_mlp.0", x, True), True)
    return (x + y) * INV_SQRT2


def seg_outer(sph, x, idx, n):
    """out[e, s, c] = sum_{t: idx[t] = e} sph[t, s] x[t, c]"""
    out = x.new_zeros(n, sph.shape[1], x.shape[1])
    for s in range(sph.shape[1]):
        out[:, s, :].index_add_(0, idx, sph[:, s:s + 1] * x)
    return out


def bilinear(P, key, radW1, S):
    """radW1 [n, I*NS] read as [n, I, NS] (efficient.py:103-104), S [n, NS, C] -> Dense over the flattened [I, C] products (efficient.py:241-251)."""
    n, ns, c = S.shape
    x = torch.einsum("eis,esc->eic", radW1.reshape(n, -1, ns), S)
    return x.reshape(n, -1) @ P[key + ".bilinear.linear.weight"].T


def forward(P, cfg, pos, z, sizes, want=None):
    """P: state dict (reference names) of tensors in the working dtype; returns (energy [B], forces [N, 3]) and fills ``want`` (dict) with intermediates."""
    dt = pos.dtype
    N, B = pos.shape[0], len(sizes)
    ns, nb = cfg["num_spherical"], cfg["num_blocks"]
    G = build_graphs(pos, sizes, cfg)
    T = lambda a: torch.as_tensor(a, dtype=torch.long)              # noqa: E731
    ms, md = map(T, G["main"]); as_, ad = map(T, G["a2ee2a"]); qs, qd = map(T, G["qint"]); ps, pd = map(T, G["a2a"])
    swap = T(G["id_swap"])

    def geom(s, t):
        v = pos[s] - pos[t]
        d = _norm(v)
        return d, -v / d[:, None]

    dm, vm = geom(ms, md); da, va = geom(as_, ad); dq, vq = geom(qs, qd); dp, _ = geom(ps, pd)
    exponent = cfg["envelope"]["exponent"]

    def rb(prefix, d, cutoff):
        return radial_basis(d, cutoff, P[prefix + ".rbf.offset"], exponent, _sf(P, prefix + ".scale_rbf.scale_factor") if cfg["scale_basis"] else 1.0)

    def sc(prefix, kind):
        return _sf(P, f"{prefix}.scale_{kind}.scale_factor") if cfg["scale_basis"] else 1.0

    rad_main = rb("radial_basis", dm, cfg["cutoff"])
    rad_sph = rb("cbf_basis_tint.radial_basis", dm, cfg["cutoff"])
    rad_qc = rb("cbf_basis_qint.radial_basis", dq, cfg["cutoff_qint"])
    rad_aea = rb("radial_basis_aeaint", da, cfg["cutoff_aeaint"])
    rad_eac = rb("cbf_basis_eaint.radial_basis", da, cfg["cutoff_aeaint"])
    rad_a2a = rb("radial_basis_aint", dp, cfg["cutoff_aint"])
    # index lists
    t_in, t_out = map(T, triplets(G["main"], G["main"], N))
    a2e_in, a2e_out = map(T, triplets(G["main"], G["a2ee2a"], N))
    e2a_in, e2a_out = map(T, triplets(G["a2ee2a"], G["main"], N))
    qo, qq, qp = map(T, quadruplets(G["main"], G["qint"], N))
    clamp = lambda x: x.clamp(-1, 1)                                # noqa: E731
    Y_e2e = zonal(clamp((vm[t_out] * vm[t_in]).sum(-1)), ns) * sc("cbf_basis_tint", "cbf")
    Y_a2e = zonal(clamp((vm[a2e_out] * va[a2e_in]).sum(-1)), ns) * sc("cbf_basis_aeint", "cbf")
    Y_e2a = zonal(clamp((va[e2a_out] * vm[e2a_in]).sum(-1)), ns) * sc("cbf_basis_eaint", "cbf")
    # quadruplets: cos(abd) on the (qint, main-in) pair, cos(cab) and the dihedral half angle on the full quadruplet (gemnet_oc.py:597-655)
    V_ba, V_db, V_ca = vq[qq], vm[qp], vm[qo]
    Y_abd = zonal(clamp((V_ba * V_db).sum(-1)), ns) * sc("cbf_basis_qint", "cbf")
    ca_x, db_x = torch.cross(V_ca, V_ba, dim=-1), torch.cross(V_db, V_ba, dim=-1)
    ang = torch.atan2(torch.cross(ca_x, db_x, dim=-1).norm(dim=-1).clamp(min=1e-9), (ca_x * db_x).sum(-1))
    Y_quad = (zonal(clamp((V_ca * V_ba).sum(-1)), ns)[:, :, None] * zonal(torch.cos(ang), ns)[:, None, :]).reshape(len(qo), -1) * sc("sbf_basis_qint", "sbf")
    # shared embeddings of the radial parts (gemnet_oc.py:1048-1103)
    W3 = lambda k: P[k + ".weight"].reshape(P[k + ".weight"].shape[0], -1)      # noqa: E731
    base = {"e2e_rad": dense(P, "mlp_rbf_tint", rad_main), "e2e_cir": rad_sph @ W3("mlp_cbf_tint"),
            "qint_rad": dense(P, "mlp_rbf_qint", rad_main), "qint_cir": rad_qc @ W3("mlp_cbf_qint"), "qint_sph": rad_sph @ W3("mlp_sbf_qint"),
            "a2e_rad": dense(P, "mlp_rbf_aeint", rad_aea), "a2e_cir": rad_sph @ W3("mlp_cbf_aeint"),
            "e2a_rad": dense(P, "mlp_rbf_eaint", rad_main), "e2a_cir": rad_eac @ W3("mlp_cbf_eaint"),
            "a2a": rad_a2a @ P["mlp_rbf_aint.weight"].T, "h": dense(P, "mlp_rbf_h", rad_main), "out": dense(P, "mlp_rbf_out", rad_main)}
    Em = len(ms)

    def up_sym(k, x):
        return (dense(P, k + ".up_projection_ca", x, True) + dense(P, k + ".up_projection_ac", x, True)[swap]) * INV_SQRT2

    def trip(k, x, kind):
        x = dense(P, k + ".dense_ba", x, True)
        if kind == "a2e":
            x = x[as_]
        x = x * dense(P, k + ".mlp_rbf", base[kind + "_rad"]) * _sf(P, k + ".scale_rbf.scale_factor")
        x = dense(P, k + ".down_projection", x, True)
        if kind == "e2e":
            S = seg_outer(Y_e2e, x[t_in], t_out, Em)
        elif kind == "a2e":
            S = seg_outer(Y_a2e, x[a2e_in], a2e_out, Em)
        else:
            S = seg_outer(Y_e2a, x[e2a_in], e2a_out, len(as_))
        n, _, c = S.shape
        X = torch.einsum("eis,esc->eic", base[kind + "_cir"].reshape(n, -1, ns), S)
        if kind == "e2a":                                           # second aggregation: over the a2ee2a edges of each target atom (efficient.py:231-240)
            X = X.new_zeros(N, X.shape[1], c).index_add_(0, ad, X)
        y = X.reshape(X.shape[0], -1) @ P[k + ".mlp_cbf.bilinear.linear.weight"].T * _sf(P, k + ".scale_cbf_sum.scale_factor")
        return dense(P, k + ".up_projection_ca", y, True) if kind == "e2a" else up_sym(k, y)

    def quad(k, m):
        x = dense(P, k + ".dense_db", m, True) * dense(P, k + ".mlp_rbf", base["qint_rad"]) * _sf(P, k + ".scale_rbf.scale_factor")
        x = dense(P, k + ".down_projection", x, True)
        cir = torch.einsum("tis,ts->ti", base["qint_cir"].reshape(len(qs), -1, ns)[qq], Y_abd)
        x = x[qp] * dense(P, k + ".mlp_cbf", cir) * _sf(P, k + ".scale_cbf.scale_factor")
        S = seg_outer(Y_quad, x, qo, Em)
        y = bilinear(P, k + ".mlp_sbf", base["qint_sph"], S) * _sf(P, k + ".scale_sbf_sum.scale_factor")
        return up_sym(k, y)

    def pair(k, h):
        x = dense(P, k + ".down_projection", h, True)
        S = seg_outer(base["a2a"], x[ps], pd, N)                                                         # [N, emb_rbf, C]
        y = S.reshape(N, -1) @ P[k + ".bilinear.linear.weight"].T * _sf(P, k + ".scale_rbf_sum.scale_factor")
        return dense(P, k + ".up_projection", y, True)

    def mlp(k, x, names):
        for nme in names:
            x = dense(P, f"{k}.{nme}", x, True) if (f"{k}.{nme}.linear.weight" in P) else residual(P, f"{k}.{nme}", x)
        return x

    def layer_names(k):
        idx = sorted({int(n[len(k) + 1:].split(".")[0]) for n in P if n.startswith(k + ".")})
        return [str(i) for i in idx]

    def atom_update(k, m, basis, head="layers"):
        x = m.new_zeros(N, m.shape[1]).index_add_(0, md, m * dense(P, k + ".dense_rbf", basis))
        x = x * _sf(P, k + ".scale_sum.scale_factor")
        return mlp(f"{k}.{head}", x, layer_names(f"{k}.{head}"))

    def out_block(k, h, m):
        xE = atom_update(k, m, base["out"])
        xE = (xE + h) * INV_SQRT2
        xE = mlp(k + ".seq_energy2", xE, layer_names(k + ".seq_energy2"))
        xF = mlp(k + ".seq_forces", m, layer_names(k + ".seq_forces"))
        xF = xF * dense(P, k + ".dense_rbf_F", base["out"]) * _sf(P, k + ".scale_rbf_F.scale_factor")
        return xE, xF

    def edge_embedding(k, h, m):
        return dense(P, k + ".dense", torch.cat([h[ms], h[md], m], dim=-1), True)

    h = P["atom_emb.embeddings.weight"][z - 1]
    m = edge_embedding("edge_emb", h, rad_main)
    xs = [out_block("out_blocks.0", h, m)]
    rec = want if want is not None else {}
    rec["edge_emb"], rec["out0"] = m, xs[0]
    for i in range(nb):
        k = f"int_blocks.{i}"
        x = dense(P, k + ".dense_ca", m, True) + trip(k + ".trip_interaction", m, "e2e") + quad(k + ".quad_interaction", m) + trip(k + ".atom_edge_interaction", h, "a2e")
        x = x / 2.0                                                                                     # 1 / sqrt(2 + quad + a2e)
        h = (h + trip(k + ".edge_atom_interaction", m, "e2a") + pair(k + ".atom_interaction", h)) / math.sqrt(3.0)
        x = mlp(k + ".layers_before_skip", x, layer_names(k + ".layers_before_skip"))
        m = (m + x) * INV_SQRT2
        m = mlp(k + ".layers_after_skip", m, layer_names(k + ".layers_after_skip"))
        if any(n.startswith(k + ".atom_emb_layers.") for n in P):
            h = mlp(k + ".atom_emb_layers", h, layer_names(k + ".atom_emb_layers"))
        h = (h + atom_update(k + ".atom_update", m, base["h"])) * INV_SQRT2
        m2 = edge_embedding(k + ".concat_layer", h, m)
        m2 = mlp(k + ".residual_m", m2, layer_names(k + ".residual_m"))
        m = (m + m2) * INV_SQRT2
        rec[f"int{i}"] = (h, m)
        xs.append(out_block(f"out_blocks.{i + 1}", h, m))
        rec[f"out{i + 1}"] = xs[-1]
    xE = mlp("out_mlp_E", torch.cat([x[0] for x in xs], dim=-1), layer_names("out_mlp_E"))
    xF = mlp("out_mlp_F", torch.cat([x[1] for x in xs], dim=-1), layer_names("out_mlp_F"))
    E_t = xE.to(P["out_energy.linear.weight"].dtype) @ P["out_energy.linear.weight"].T                   # the reference evaluates the two heads in fp32 (:1204-1207)
    F_st = xF.to(P["out_forces.linear.weight"].dtype) @ P["out_forces.linear.weight"].T
    batch = torch.repeat_interleave(torch.arange(B), torch.as_tensor(np.asarray(sizes)))
    energy = E_t.new_zeros(B, 1).index_add_(0, batch, E_t).squeeze(1)
    F_st = (F_st + F_st[swap]) / 2                                                                       # forces_coupled: mean over the two directions (:1217-1230)
    forces = F_st.new_zeros(N, 3).index_add_(0, md, F_st * vm.to(F_st.dtype))
    rec["graphs"] = G
    return energy, forces


def loss(E, F, y, f_target):
    return (E - y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - f_target, dim=-1).mean()          # config/model/gemnet-oc.yaml:78-85
