"""GemNet-OC on the MI355X against golden vectors of the REAL reference classes (oracle/make_golden_gemnet.py ran
nablaDFT/gemnet_oc/gemnet_oc.py on CPU): graphs and interaction indices bit-exact, features / energies / forces / gradients within the north-star
tolerance (1e-5 relative; the bound used per tensor is stated where it is applied, next to the reference's own fp32-vs-fp64 error)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle.gemnet_params import make_state, probe_direction  # noqa: E402  (test infrastructure: deterministic parameter values)

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
    def __init__(self, d, dev):
        sizes = d["sizes"]
        self.pos = torch.tensor(d["pos"], device=dev)
        self.z = torch.tensor(d["z"], device=dev, dtype=torch.long)
        self.batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(dev)
        self.y = torch.tensor(d["y"], device=dev, dtype=torch.float32)
        self.forces = torch.tensor(d["f_target"], device=dev, dtype=torch.float32)


from tests.helpers import assert_parity  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(GOLD, "gemnet_small.npz"))


@pytest.fixture(scope="module")
def full():
    return np.load(os.path.join(GOLD, "gemnet_full.npz"))


def build(cfg, d, dev, fit_scales):
    from nabladft_amd.gemnet_oc import GemNetOC
    torch.manual_seed(0)
    net = GemNetOC(**cfg)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters()]
    assert [n for n, _ in names] == list(d["param_names"])
    missing = net.load_state_dict(make_state(names, int(d["seed"]), fit_scales), strict=False)
    assert not missing.unexpected_keys
    return net.to(dev)


def triplets(out_ei, in_ei, exclude_same_source=True):
    """interaction_indices.py:13-118 restated on edge lists in the reference's order: for every out edge (ascending id) the in edges with the same target in
    source order (SparseTensor rows), minus those with the same source."""
    t_in = in_ei[1]
    order = np.lexsort((in_ei[0], t_in))
    by_t = {}
    for e in order:
        by_t.setdefault(int(t_in[e]), []).append(int(e))
    i_in, i_out, agg = [], [], []
    for o in range(out_ei.shape[1]):
        k = 0
        for e in by_t.get(int(out_ei[1, o]), []):
            if in_ei[0, e] != out_ei[0, o]:
                i_in.append(e); i_out.append(o); agg.append(k); k += 1
    return np.array(i_in), np.array(i_out), np.array(agg)


def test_graphs_and_indices_bit_exact(small, full):
    from nabladft_amd.gemnet_oc import build_graphs
    dev = torch.device("cuda:0")
    for d, cfg in ((small, SMALL), (full, FULL)):
        data = Data(d, dev)
        G = build_graphs(data.pos, data.batch, data.z, cfg["cutoff"], cfg["cutoff_qint"], cfg["cutoff_aeaint"], cfg["cutoff_aint"], cfg["max_neighbors"],
                         cfg["max_neighbors_qint"], cfg["max_neighbors_aeaint"], cfg["max_neighbors_aint"])
        R = G.to_reference()
        for name in ("a2a", "a2ee2a", "qint", "main"):
            assert np.array_equal(R[name]["edge_index"], d[f"f32:graph:{name}:edge_index"]), name
            assert np.array_equal(R[name]["distance"], d[f"f32:graph:{name}:distance"]), name + " distances"       # bit-exact
            assert np.abs(R[name]["vector"] - d[f"f32:graph:{name}:vector"]).max() <= 1.2e-7, name                   # one division; <= 1 ulp
        assert np.array_equal(R["id_swap"], d["f32:graph:id_swap"])
        t = {k: v.cpu().numpy() for k, v in G.t.items() if torch.is_tensor(v)}
        for name, ptr, dst in (("a2a", "row_ptr", "dst"), ("a2ee2a", "ptr_a", "a_dst")):               # target_neighbor_idx = position in the CSR row
            pos_in_row = np.arange(t[dst].shape[0]) - t[ptr][t[dst]]
            assert np.array_equal(pos_in_row, d[f"f32:graph:{name}:target_neighbor_idx"])
        # the kernels enumerate triplets / quadruplets implicitly; their definition on the device graphs reproduces the reference's lists
        main, aea, qint = R["main"]["edge_index"], R["a2ee2a"]["edge_index"], R["qint"]["edge_index"]
        i_in, i_out, agg = triplets(main, main)
        assert np.array_equal(i_out, d["f32:trip:e2e:out"]) and np.array_equal(i_in, d["f32:trip:e2e:in"])
        i_in, i_out, agg = triplets(main, aea)
        assert np.array_equal(i_out, d["f32:trip:a2e:out"])
        i_in, i_out, agg = triplets(aea, main)
        assert np.array_equal(i_out, d["f32:trip:e2a:out"])
        if "f32:trip:a2e:in" in d.files:
            i_in, i_out, agg = triplets(main, aea)
            assert np.array_equal(i_in, d["f32:trip:a2e:in"]) and np.array_equal(agg, d["f32:trip:a2e:out_agg"])
            i_in, i_out, agg = triplets(aea, main)
            assert np.array_equal(i_in, d["f32:trip:e2a:in"]) and np.array_equal(agg, d["f32:trip:e2a:out_agg"])
        # quadruplet count: sum over out edges (c -> a), qint in-edges (b -> a) with b != c, main in-edges (d -> b) with d not in {a, c}
        m_by_t = {}
        for e in range(main.shape[1]):
            m_by_t.setdefault(int(main[1, e]), []).append(int(main[0, e]))
        q_by_t = {}
        for e in range(qint.shape[1]):
            q_by_t.setdefault(int(qint[1, e]), []).append(int(qint[0, e]))
        nq = np.zeros(main.shape[1], dtype=np.int64)
        for o in range(main.shape[1]):
            c, a = int(main[0, o]), int(main[1, o])
            for b in q_by_t.get(a, []):
                if b != c:
                    nq[o] += sum(1 for dd in m_by_t.get(b, []) if dd != a and dd != c)
        assert np.array_equal(np.repeat(np.arange(main.shape[1]), nq), d["f32:quad:out"])
        # rows of the (qint edge, main in-edge of its source) table
        assert G.Tin == sum(len(m_by_t.get(int(b), [])) for b in qint[0])


def _main_rows(G, R):
    return torch.tensor(R["main_ref_id"], device="cuda:0")


def test_forward_small_matches_reference_layer_by_layer(small):
    d = small
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev, True)
    data = Data(d, dev)
    with torch.no_grad():
        E, F, inter = net(data, return_intermediates=True)
    G = inter["graphs"]
    R = G.to_reference()
    rid = R["main_ref_id"]
    B = inter["bases"]
    TOL = 2e-5                                        # fp32, ~100 layers: the reference's own fp32 run differs from its fp64 run by up to 1e-5 on these tensors
    checks = {
        "basis:rad_main_raw": (B["rad_main_raw"], True), "basis:atom_update": (B["atom_update"], True), "basis:output": (B["output"], True),
        "basis:a2a_rad": (B["a2a_rad"], False), "basis:e2e:rad": (B["e2e"]["rad"], True), "basis:a2e:rad": (B["a2e"]["rad"], False),
        "basis:qint:rad": (B["qint"]["rad"], True), "basis:e2a:rad": (B["e2a"]["rad"], True),
    }
    for k, (v, is_main) in checks.items():
        ref = d["f32:" + k]
        got = v.cpu().numpy()
        if k == "basis:a2a_rad":                      # the reference pads per atom: [N, emb_rbf, Kmax]
            tni, tgt = d["f32:graph:a2a:target_neighbor_idx"], d["f32:graph:a2a:edge_index"][1]
            ref = ref[tgt, :, tni]
        assert rel(got, ref[rid] if is_main else ref) < TOL, k
    # rad_W1 of the circular / spherical bases, the reference's reshape(E, -1, num_spherical) included
    assert rel(B["e2e"]["cir"].cpu().numpy(), d["f32:basis:e2e:cir:rad_W1"].reshape(len(rid), -1)[rid]) < TOL
    assert rel(B["qint"]["sph"].cpu().numpy(), d["f32:basis:qint:sph:rad_W1"].reshape(len(rid), -1)[rid]) < TOL
    assert rel(B["a2e"]["cir"].cpu().numpy(), d["f32:basis:a2e:cir:rad_W1"].reshape(len(rid), -1)[rid]) < TOL
    # per-block features (edge tensors compared in the reference's edge order)
    f64 = {k[4:]: d[k] for k in d.files if k.startswith("f64:")}
    pairs = [("atom_emb", inter["atom_emb"], False), ("edge_emb", inter["edge_emb"], True)]
    for i in range(SMALL["num_blocks"]):
        pairs += [(f"int{i}:0", inter[f"int{i}"][0], False), (f"int{i}:1", inter[f"int{i}"][1], True)]
    for i in range(SMALL["num_blocks"] + 1):
        pairs += [(f"out{i}:0", inter[f"out{i}"][0], False), (f"out{i}:1", inter[f"out{i}"][1], True)]
    for k, v, is_main in pairs:
        ref64 = f64[k][rid] if is_main else f64[k]
        ref32 = d["f32:" + k][rid] if is_main else d["f32:" + k]
        err, own = rel(v.cpu().numpy(), ref64), rel(ref32, ref64)
        assert err < max(TOL, 3 * own), (k, err, own)
    assert_parity("gemnet_small E", E.cpu().numpy(), d["f64:E"], d["f32:E"])
    assert_parity("gemnet_small F", F.cpu().numpy(), d["f64:F"], d["f32:F"])
    assert np.abs(E.cpu().numpy() - d["f32:E"]).max() < 1e-5 * max(1.0, np.abs(d["f32:E"]).max())


def _loss(E, F, data):
    return (E - data.y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - data.forces, dim=-1).mean()      # config/model/gemnet-oc.yaml:78-85


def test_gradients_small_match_reference(small):
    d = small
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev, True)
    data = Data(d, dev)
    E, F = net(data)
    loss = _loss(E, F, data)
    loss.backward()
    assert abs(float(loss.detach()) - float(d["f64:loss"])) < 2e-5 * abs(float(d["f64:loss"]))
    worst = 0.0
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        ref64, ref32 = d["f64:grad:" + name], d["f32:grad:" + name]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref64)
        scale = max(np.abs(ref64).max(), 1e-30)
        err, own = np.abs(g - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
        worst = max(worst, err)
        assert err < max(5e-5, 3 * own), (name, err, own)     # bound: the larger of 5e-5 and 3x the reference's own fp32-vs-fp64 error on this tensor
    assert worst > 0.0


def test_full_config_forward_and_gradients(full):
    d = full
    dev = torch.device("cuda:0")
    net = build(FULL, d, dev, False)
    assert sum(p.numel() for p in net.parameters()) == 37815873
    data = Data(d, dev)
    E, F, inter = net(data, return_intermediates=True)
    rid = inter["graphs"].to_reference()["main_ref_id"]
    TOL = 3e-5
    for k, v, is_main in (("int0:0", inter["int0"][0], False), ("int0:1", inter["int0"][1], True), ("int3:0", inter["int3"][0], False),
                          ("int3:1", inter["int3"][1], True), ("out0:0", inter["out0"][0], False), ("out4:1", inter["out4"][1], True)):
        got = v.detach().cpu().numpy()
        if is_main:
            inv = np.argsort(rid)                     # row r of the reference = CSR slot inv[r]
            got = got[inv]
        assert rel(got[::9], d["f32:" + k]) < TOL, k
    assert_parity("gemnet_full E", E.detach().cpu().numpy(), d["f64:E"], d["f32:E"])
    assert_parity("gemnet_full F", F.detach().cpu().numpy(), d["f64:F"], d["f32:F"])
    loss = _loss(E, F, data)
    loss.backward()
    assert_parity("gemnet_full loss", float(loss.detach()), d["f64:loss"], d["f32:loss"])
    names = [n for n in d["param_names"] if not n.endswith("scale_factor")]        # the fixture's gradient arrays cover the trainable tensors, in this order
    n64, p64, p32 = d["f64:grad_norm"], d["f64:grad_probe"], d["f32:grad_probe"]
    assert len(names) == len(n64)
    params = dict(net.named_parameters())
    for i, name in enumerate(names):
        p = params[name]
        assert p.requires_grad
        g = p.grad.double().cpu()
        norm = float(g.norm())
        assert abs(norm - n64[i]) < 1e-4 * max(n64[i], 1e-12), (name, norm, n64[i])
        probe = float((g * probe_direction(name, g.shape, int(d["seed"]))).sum())
        own = abs(p32[i] - p64[i])
        assert abs(probe - p64[i]) < max(1e-4 * n64[i] * np.sqrt(g.numel()) * 0.05, 3 * own), (name, probe, p64[i], own)


def test_equivariance_and_batch_independence(small):
    d = small
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev, True)
    data = Data(d, dev)
    with torch.no_grad():
        E0, F0 = net(data)
        q, _ = np.linalg.qr(np.random.default_rng(3).normal(size=(3, 3)))
        q = torch.tensor(q * np.sign(np.linalg.det(q)), device=dev, dtype=torch.float32)
        data2 = Data(d, dev)
        data2.pos = data.pos @ q.T + torch.tensor([1.5, -2.0, 0.7], device=dev)
        E1, F1 = net(data2)
        assert (E1 - E0).abs().max() < 2e-5 * max(1.0, float(E0.abs().max()))
        assert (F1 - F0 @ q.T).abs().max() < 2e-5 * max(1.0, float(F0.abs().max()))
        # the second molecule alone gives the same energy / forces as inside the batch (no cross-molecule edges, deterministic sums)
        sizes = d["sizes"]
        a0, a1 = int(sizes[0]), int(sizes[0] + sizes[1])
        one = Data(d, dev)
        one.pos, one.z, one.batch = data.pos[a0:a1], data.z[a0:a1], torch.zeros(a1 - a0, dtype=torch.long, device=dev)
        E2, F2 = net(one)
        assert (E2[0] - E0[1]).abs() < 2e-6 * max(1.0, float(E0.abs().max())) and (F2 - F0[a0:a1]).abs().max() < 2e-6 * max(1.0, float(F0.abs().max()))
        # bitwise reproducible
        E3, F3 = net(data)
        assert torch.equal(E3, E0) and torch.equal(F3, F0)


def test_bf16_gemm_matches_bf16_rounded_operands_and_model_stays_close(small, full):
    """bf16 mode (BASELINE.json configs[2]): the MFMA kernel equals an fp32 product of bf16-rounded operands to accumulation round-off (so the only
    deviation from the fp32 path is the documented operand rounding), asymmetric operands catch layout swaps; the model's outputs move by O(1e-2)."""
    import ctypes as C
    from nabladft_amd import _lib, gemnet_oc
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for M, N, K in [(300, 200, 64), (1000, 512, 512), (257, 48, 32), (4096, 1568, 128)]:
        x, W = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
        Wb, WbT = torch.empty(N, K, device=dev, dtype=torch.bfloat16), torch.empty(K, N, device=dev, dtype=torch.bfloat16)
        _lib.check(lib.nq_bf16_pack(_lib.ptr(W), N, K, _lib.ptr(Wb), _lib.ptr(WbT), _lib.stream_ptr()))
        assert torch.equal(Wb, W.to(torch.bfloat16)) and torch.equal(WbT, W.t().contiguous().to(torch.bfloat16))
        xr, Wr = x.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
        ref = xr @ Wr.T
        y = torch.empty(M, N, device=dev)
        _lib.check(lib.nq_linear_forward_bf16(_lib.ptr(x), _lib.ptr(Wb), _lib.ptr(y), None, None, 0.0, 0.0, M, N, K, _lib.stream_ptr()))
        assert (y.double() - ref).abs().max() < 2e-6 * ref.abs().max() * (K ** 0.5), (M, N, K)
        res, act = torch.randn(M, N, generator=g).to(dev), torch.empty(M, N, device=dev)
        _lib.check(lib.nq_linear_forward_bf16(_lib.ptr(x), _lib.ptr(Wb), _lib.ptr(y), _lib.ptr(act), _lib.ptr(res), 0.5, 0.25, M, N, K, _lib.stream_ptr()))
        assert (act.double() - (0.5 * res.double() + 0.25 * torch.nn.functional.silu(ref))).abs().max() < 1e-5 * ref.abs().max()
        if N % 32 == 0:
            gy = torch.randn(M, N, generator=g).to(dev)
            gx = torch.full((M, K), 1.0, device=dev)
            _lib.check(lib.nq_linear_input_grad_bf16(_lib.ptr(gy), _lib.ptr(WbT), _lib.ptr(gx), M, N, K, 1, _lib.stream_ptr()))
            refg = gy.to(torch.bfloat16).double() @ Wr + 1.0
            assert (gx.double() - refg).abs().max() < 2e-6 * refg.abs().max() * (N ** 0.5), (M, N, K)
    for M, N, K in [(5000, 512, 512), (2049, 48, 200), (20468, 512, 64), (3000, 1, 512)]:           # weight gradient: transposed bf16 operands, split over the rows
        gy, x = torch.randn(M, N, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)
        gW = torch.empty(N, K, device=dev)
        scr = torch.empty(int(lib.nq_weight_grad_bf16_scratch_bytes(M, N, K)), device=dev, dtype=torch.uint8)
        _lib.check(lib.nq_linear_weight_grad_bf16(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), _lib.stream_ptr()))
        ref = gy.to(torch.bfloat16).double().T @ x.to(torch.bfloat16).double()
        assert (gW.double() - ref).abs().max() < 2e-6 * ref.abs().max() * (M ** 0.5), (M, N, K)
        gW2 = torch.empty(N, K, device=dev)
        _lib.check(lib.nq_linear_weight_grad_bf16(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gW2), M, N, K, _lib.ptr(scr), _lib.stream_ptr()))
        assert torch.equal(gW, gW2)                                                                   # fixed-order reduction
    d = full
    net = build(FULL, d, dev, False)
    data = Data(d, dev)
    E0, F0 = net(data)
    _loss(E0, F0, data).backward()
    g0 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    gemnet_oc.set_gemm_precision("bf16")
    try:
        E1, F1 = net(data)
        _loss(E1, F1, data).backward()
    finally:
        gemnet_oc.set_gemm_precision("f32")
    assert (E1 - E0).abs().max() < 3e-2 * max(1.0, float(E0.abs().max())) and (F1 - F0).abs().max() < 3e-2 * float(F0.abs().max())
    num = sum(float(((p.grad - g0[k]) ** 2).sum()) for k, p in net.named_parameters() if p.grad is not None)
    den = sum(float((v ** 2).sum()) for v in g0.values())
    assert (num / den) ** 0.5 < 5e-2                                   # whole-gradient relative deviation of the bf16 step
    assert float((E1 - E0).abs().max()) > 0.0                           # the mode really took another path


def test_bf16_weights_follow_in_place_optimizer_updates(small):
    """FlatParameters + AdamW rewrite the parameters through one flat buffer (no per-parameter version bump): the bf16 weight copies must be re-packed
    every forward.  Three optimiser steps in bf16 must move the forward output and track the same three steps in fp32."""
    from nabladft_amd import gemnet_oc
    from nabladft_amd.trainer import FlatParameters
    dev = torch.device("cuda:0")
    cfg = dict(SMALL, emb_size_atom=64, emb_size_edge=64, emb_size_trip_in=32, emb_size_trip_out=32, emb_size_quad_in=32, emb_size_quad_out=32,
               emb_size_aint_in=32, emb_size_aint_out=32)                  # contraction sizes that are multiples of 32 -> the bf16 kernels really run
    base = Data(small, dev)

    class Rep:                                                              # >= 256 rows per Dense product: 8 copies of the fixture's molecules
        pass
    rep = Rep()
    reps = 8
    nmol = int(base.batch.max()) + 1
    rep.pos = torch.cat([base.pos + 50.0 * i for i in range(reps)])
    rep.z = base.z.repeat(reps)
    rep.batch = torch.cat([base.batch + nmol * i for i in range(reps)])
    cnt = torch.bincount(rep.batch)
    rep.ptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])
    rep.y = torch.zeros(nmol * reps, device=dev)
    rep.forces = torch.zeros_like(rep.pos)
    outs = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(3)
        net = gemnet_oc.GemNetOC(**cfg).to(dev)
        gemnet_oc.set_gemm_precision(mode)
        try:
            flat = FlatParameters(net.parameters())
            opt = torch.optim.AdamW([flat.flat], lr=1e-2, weight_decay=0)
            seq = []
            for _ in range(3):
                flat.zero_grad()
                E, F = net(rep)
                seq.append(E.detach().clone())
                ((E - 1.0) ** 2).mean().add((F ** 2).mean()).backward()
                opt.step()
            with torch.no_grad():
                seq.append(net(rep)[0].clone())
        finally:
            gemnet_oc.set_gemm_precision("f32")
        outs[mode] = seq
    f, b = outs["f32"], outs["bf16"]
    scale = float(f[0].abs().max()) + 1e-6
    assert float((b[0] - b[1]).abs().max()) > 1e-3 * scale and float((b[2] - b[3]).abs().max()) > 1e-4 * scale      # every step changed what the bf16 forward sees
    move = float((f[3] - f[0]).abs().max())
    assert move > 1e-2 * scale                                              # the three fp32 steps moved the energies visibly ...
    assert float((b[3] - f[3]).abs().max()) < 0.25 * move + 3e-2 * scale    # ... and the bf16 run followed them (stale weights would stay at step 0)


def test_invalid_inputs_fail_loudly(small):
    """Out-of-table atomic numbers and a molecule without neighbours raise instead of reading out of bounds / silently producing zeros."""
    dev = torch.device("cuda:0")
    net = build(SMALL, small, dev, True)
    data = Data(small, dev)
    bad = Data(small, dev)
    bad.z = data.z.clone()
    bad.z[3] = 99
    with pytest.raises(IndexError):
        net(bad)
    far = Data(small, dev)
    far.pos = data.pos.clone()
    far.pos[: int(small["sizes"][0])] *= 100.0                       # first molecule blown up: no pair inside the cutoffs
    with pytest.raises((ValueError, IndexError)):
        net(far)
    from nabladft_amd.escn import eSCN
    from tests.test_escn_cpu import SMALL as ES
    es = eSCN(**ES).to(dev)
    bad.z[3] = 77
    with pytest.raises(IndexError):
        es(bad)
    E_far, F_far = es(far)                                           # eSCN (like its reference) accepts atoms without neighbours: zero messages
    assert torch.isfinite(E_far).all() and torch.isfinite(F_far).all()
