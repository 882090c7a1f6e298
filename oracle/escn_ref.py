"""TEST INFRASTRUCTURE -- CPU restatement (torch, any float dtype) of the reference eSCN forward pass (nablaDFT/escn/escn.py:295-433, so3.py), pinned to the
golden vectors the REAL reference classes produced (tests/golden/escn_*.npz); checker of the CPU tests, ``__graft_entry__.smoke()`` and the eSCN bench's
``cpu_baseline``.  The product never imports this file.  Harmonics / S2 grids come from oracle/e3nn_mini.py (e3nn restated: parity unpinned for those);
the J matrices of the Wigner recursion are recomputed from those harmonics (equal to the reference's Jd.pt: tests/test_escn_cpu.py)."""
import math

import numpy as np
import torch

from oracle import e3nn_mini as M

_G = torch.tensor([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)


def j_matrices(lmax):
    g = torch.Generator().manual_seed(0)
    x = torch.nn.functional.normalize(torch.randn(8 * (2 * lmax + 1), 3, generator=g, dtype=torch.float64), dim=-1)
    out = []
    for l in range(lmax + 1):
        Y, Yg = M.spherical_harmonics([l], x, True), M.spherical_harmonics([l], x @ _G.T, True)
        out.append(torch.linalg.lstsq(Y, Yg).solution.T.contiguous())
    return out


def sphere_constants(cfg, dtype):
    """(sphere_points [P, 3], sphharm_weights [P, (lmax+1)^2]) in ``dtype``: escn/sampling.py:15-36 and escn.py:170-176 (non-trainable parameters the
    reference builds in the default dtype of the run)."""
    n = cfg["num_sphere_samples"]
    golden = (1 + 5 ** 0.5) / 2
    i = torch.arange(n, dtype=dtype).view(-1, 1)
    theta = 2 * math.pi * i / golden
    phi = torch.arccos(1 - 2 * (i + 0.5) / n)
    pts = torch.cat([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], dim=1)
    d = ((pts.view(1, -1, 3) - pts.view(-1, 1, 3)) ** 2).sum(dim=2)
    s = 1.0 / torch.exp(-d / (0.5 * 0.3)).sum(dim=1)
    pts = pts * (n * s / s.sum()).view(-1, 1)
    return pts, M.spherical_harmonics(list(range(cfg["lmax_list"][0] + 1)), pts, False)


def radius_graph(pos, sizes, cutoff, K):
    src, dst = [], []
    starts = np.concatenate([[0], np.cumsum(sizes)])
    p = pos.to(torch.float32)
    for b in range(len(sizes)):
        a0, a1 = int(starts[b]), int(starts[b + 1])
        d2 = ((p[a0:a1, None] - p[None, a0:a1]) ** 2).sum(-1)
        for i in range(a1 - a0):
            js = [j for j in range(a1 - a0) if j != i and float(d2[i, j]) < float(np.float32(cutoff * cutoff))][:K]
            src += [a0 + j for j in js]; dst += [a0 + i] * len(js)
    return torch.tensor(src), torch.tensor(dst)


def frames(vec):
    """escn.py:435-487 with a deterministic helper (the coordinate axis least aligned with the edge) instead of the random vector."""
    nx = vec / vec.norm(dim=1, keepdim=True)
    helper = torch.nn.functional.one_hot(nx.abs().argmin(dim=1), 3).to(vec.dtype)
    nz = torch.nn.functional.normalize(torch.cross(nx, helper, dim=1), dim=1)
    ny = -torch.nn.functional.normalize(torch.cross(nx, nz, dim=1), dim=1)
    return torch.stack([nz, nx, ny], dim=1)                              # rows: edge_rot_mat = transpose([nz | nx | ny])


def wigner(rot, lmax, J):
    x = rot @ rot.new_tensor([0.0, 1.0, 0.0])
    alpha, beta = M.xyz_to_angles(x)
    R = M.angles_to_matrix(alpha, beta, torch.zeros_like(alpha)).transpose(-1, -2) @ rot
    gamma = torch.atan2(R[..., 0, 2], R[..., 0, 0])

    def zrot(angle, l):
        Mx = angle.new_zeros(angle.shape[0], 2 * l + 1, 2 * l + 1)
        f = torch.arange(l, -l - 1, -1, dtype=angle.dtype)
        i = torch.arange(2 * l + 1)
        Mx[:, i, 2 * l - i] = torch.sin(f * angle[:, None])
        Mx[:, i, i] = torch.cos(f * angle[:, None])
        return Mx

    W = rot.new_zeros(rot.shape[0], (lmax + 1) ** 2, (lmax + 1) ** 2)
    for l in range(lmax + 1):
        Jl = J[l].to(rot.dtype)
        W[:, l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = zrot(alpha, l) @ Jl @ zrot(beta, l) @ Jl @ zrot(gamma, l)
    return W


def s2(lmax, mmax, dtype):
    nb = 2 * (lmax + 1)
    na = 2 * (mmax + 1) + 1 if lmax == mmax else 2 * mmax + 1
    T = M._s2_samples(lmax, nb, na).reshape(nb * na, -1)
    w = M.s2_quadrature_weights(nb) * (2 * math.pi / na)
    return T.to(dtype), (T * w.repeat_interleave(na)[:, None]).to(dtype)


def forward(P, cfg, pos, z, sizes, rot=None, probe=None):
    """P: state dict (reference names); returns (energy [B], forces [N, 3]).  probe: optional dict that receives intermediates (final embedding, sphere-point
    features, per-point force magnitudes) for error localisation."""
    dt = pos.dtype
    lmax, mmax, C = cfg["lmax_list"][0], cfg["mmax_list"][0], cfg["sphere_channels"]
    nf = (lmax + 1) ** 2
    N, B = pos.shape[0], len(sizes)
    src, dst = radius_graph(pos, sizes, cfg["cutoff"], cfg["max_neighbors"])
    vec = pos[src] - pos[dst]
    dist = vec.norm(dim=-1)
    rot = frames(vec) if rot is None else rot
    W = wigner(rot, lmax, j_matrices(lmax))
    red = [l * l + l + m for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]                    # coefficient_idx(lmax, mmax), l-primary
    lm = [(l, m) for l in range(lmax + 1) for m in range(-min(l, mmax), min(l, mmax) + 1)]
    mprim = [lm.index((l, 0)) for l in range(lmax + 1)]
    msize = [lmax + 1]
    for m in range(1, mmax + 1):
        mprim += [lm.index((l, m)) for l in range(m, lmax + 1)] + [lm.index((l, -m)) for l in range(m, lmax + 1)]
        msize.append(lmax - m + 1)
    Wr = W[:, red, :]
    Tr, Fr = s2(lmax, mmax, dt)
    Tr, Fr = Tr[:, red], Fr[:, red]
    Tf, Ff = s2(lmax, lmax, dt)
    act = torch.nn.functional.silu
    lin = lambda k, x: x @ P[k + ".weight"].T + (P[k + ".bias"] if (k + ".bias") in P else 0)                           # noqa: E731
    off = P["distance_expansion.offset"]
    coeff = -0.5 / (cfg["basis_width_scalar"] * float(off[1] - off[0])) ** 2
    x_dist = torch.exp(coeff * (dist[:, None] - off[None, :]) ** 2)

    def so2(k, x, xe):                                                  # x [E, n_red, C] l-primary -> same
        E = x.shape[0]
        xm = x[:, mprim]
        out = [(lin(k + ".fc1_m0", xm[:, :msize[0]].reshape(E, -1)) * act(lin(k + ".fc1_dist0", xe)))]
        out[0] = lin(k + ".fc2_m0", out[0]).view(E, -1, C)
        o = msize[0]
        for m in range(1, mmax + 1):
            kk = f"{k}.so2_conv.{m - 1}"
            blk = xm[:, o:o + 2 * msize[m]].reshape(E, 2, -1)
            g = act(lin(kk + ".fc1_dist", xe)).view(E, 2, -1)
            xr = lin(kk + ".fc2_r", lin(kk + ".fc1_r", blk) * g[:, 0:1])
            xi = lin(kk + ".fc2_i", lin(kk + ".fc1_i", blk) * g[:, 1:2])
            out.append(torch.stack([xr[:, 0] - xi[:, 1], xr[:, 1] + xi[:, 0]], dim=1).reshape(E, -1, C))
            o += 2 * msize[m]
        y = torch.cat(out, dim=1)
        inv = torch.empty(len(mprim), dtype=torch.long)
        inv[torch.tensor(mprim)] = torch.arange(len(mprim))
        return y[:, inv]

    x = pos.new_zeros(N, nf, C)
    x[:, 0] = P["sphere_embedding.weight"][z]
    for i in range(cfg["num_layers"]):
        k = f"layer_blocks.{i}"
        mb = k + ".message_block"
        xe = act(P[mb + ".edge_block.source_embedding.weight"][z[src]] + P[mb + ".edge_block.target_embedding.weight"][z[dst]] + lin(mb + ".edge_block.fc1_dist", x_dist))
        xe = act(lin(mb + ".edge_block.fc1_edge_attr", xe))
        y = so2(mb + ".so2_block_source", torch.bmm(Wr, x[src]), xe) + so2(mb + ".so2_block_target", torch.bmm(Wr, x[dst]), xe)
        y = torch.einsum("gi,egc->eic", Fr, act(torch.einsum("gi,eic->egc", Tr, y)))
        msg = x.new_zeros(N, nf, C).index_add_(0, dst, torch.bmm(Wr.transpose(1, 2), y))
        h = torch.cat([torch.einsum("gi,nic->ngc", Tf, x), torch.einsum("gi,nic->ngc", Tf, msg)], dim=2)
        h = lin(k + ".fc3_sphere", act(lin(k + ".fc2_sphere", act(lin(k + ".fc1_sphere", h)))))
        out = torch.einsum("gi,ngc->nic", Ff, h)
        x = out if i == 0 else x + out
    x_pt = torch.einsum("nic,pi->npc", x, P["sphharm_weights.0"]).reshape(-1, C)
    sp = P["sphere_points"]
    n_pts = sp.shape[0]
    e = lin("energy_block.fc3", act(lin("energy_block.fc2", act(lin("energy_block.fc1", x_pt))))).view(N, n_pts).sum(1) / n_pts
    batch = torch.repeat_interleave(torch.arange(B), torch.as_tensor(np.asarray(sizes)))
    energy = e.new_zeros(B).index_add_(0, batch, e) * 0.001
    f = lin("force_block.fc3", act(lin("force_block.fc2", act(lin("force_block.fc1", x_pt))))).view(N, n_pts, 1)
    forces = (f * sp.view(1, n_pts, 3)).sum(1) / n_pts
    if probe is not None:
        probe.update(x=x.detach(), x_pt=x_pt.detach(), f=f.detach().view(N, n_pts), e_pt=e.detach())
    return energy, forces


def loss(E, F, y, f_target):
    return (E - y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - f_target, dim=-1).mean()          # config/model/escn-oc.yaml:41-48
