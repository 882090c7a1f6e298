"""QHNet on the HIP path vs golden vectors written by the REAL reference classes (qhnet/qhnet.py, qhnet/layers.py, qhnet/loss.py) running on
oracle/e3nn_mini.py (this repo's restatement of e3nn 0.5.1 -- the one unpinned piece; oracle/make_golden_qhnet_model.py).  SURVEY.md section 8
rows a13-a20.  Tolerance: the north-star's 1e-5 relative (to the largest element of each tensor) against the reference evaluated in fp64;
the reference's own fp32 run differs from that truth by 4e-7 (H) to 4e-6 (gradients)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ORBITALS = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 7: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2], 9: [0, 0, 0, 1, 1, 2],
            16: [0, 0, 0, 0, 1, 1, 1, 2], 17: [0, 0, 0, 0, 1, 1, 1, 2], 35: [0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]}
TOL = 1e-5


class Batch:
    def __init__(self, pos, z, sizes, dev):
        self.pos = torch.tensor(pos, dtype=torch.float32, device=dev)
        self.z = torch.tensor(z, dtype=torch.long, device=dev)
        self.ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long, device=dev)
        self.batch = torch.repeat_interleave(torch.arange(len(sizes), device=dev), torch.tensor(sizes, device=dev))
        self.num_nodes = len(z)


def load_case(name, dev):
    from nabladft_amd.qhnet import QHNet
    from oracle.qhnet_params import make_state
    g = np.load(os.path.join(GOLD, f"qhnet_{name}.npz"))
    cfg = {k: (float(v) if k == "max_radius" else int(v)) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    net = QHNet(**cfg, orbitals=ORBITALS)
    state = make_state([(k, tuple(p.shape)) for k, p in net.named_parameters()], int(g["seed"]))
    missing = net.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys and all(not k.endswith(("weight", "bias", "weights", "_alpha")) or "tp" in k or "mul" in k or k.endswith(".bias")
                                               for k in missing.missing_keys), missing
    net.to(dev)
    return g, cfg, net, Batch(g["pos"], g["z"], g["sizes"], dev)


from tests.helpers import assert_parity  # noqa: E402


def rel(a, ref):
    ref = torch.as_tensor(ref, dtype=torch.float64)
    a = torch.as_tensor(a).detach().cpu().double()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-300))


def from_e3nn(x, c):
    """e3nn layout [rows, sum_l c (2l+1)] ([mul, 2l+1] per l) -> [rows, 25, c]."""
    x = torch.as_tensor(x)
    out = []
    for l in range(5):
        out.append(x[:, c * l * l:c * (l + 1) ** 2].reshape(x.shape[0], c, 2 * l + 1).transpose(1, 2))
    return torch.cat(out, dim=1)


def test_small_forward_layer_by_layer():
    dev = torch.device("cuda:0")
    g, cfg, net, batch = load_case("small", dev)
    c = cfg["hidden_size"]
    inter = {}
    hooks = []
    for i, m in enumerate(net.e3_gnn_layer):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, k=f"conv{i}": inter.__setitem__(k, out.detach())))
    for i, m in enumerate(net.e3_gnn_node_layer):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, k=f"self{i}": inter.__setitem__(k, out.detach())))
    for i, m in enumerate(net.e3_gnn_node_pair_layer):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, k=f"pair{i}": inter.__setitem__(k, out.detach())))
    hooks.append(net.expand_ii["hamiltonian"].register_forward_hook(lambda mod, inp, out: inter.__setitem__("diag_blocks", out.detach())))
    hooks.append(net.expand_ij["hamiltonian"].register_forward_hook(lambda mod, inp, out: inter.__setitem__("nondiag_blocks", out.detach())))
    with torch.no_grad():
        H = net(batch)
    # graph: bit-exact indices; geometry bases
    assert torch.equal(batch.edge_index.cpu(), torch.tensor(g["edge_index"]))
    assert torch.equal(batch.full_edge_index.cpu(), torch.tensor(g["full_edge_index"]))
    assert batch.edge_index.shape[1] < batch.full_edge_index.shape[1]           # the cutoff binds in this fixture
    assert rel(batch.edge_attr, g["edge_attr"]) < 2e-5
    assert rel(batch.edge_sh, g["edge_sh"]) < 5e-6
    assert rel(batch.node_attr, g["node_attr"]) == 0.0
    errs = {}
    for k, v in inter.items():
        ref = g["inter64_" + k]
        errs[k] = rel(v, ref) if k.endswith("blocks") else rel(v, from_e3nn(ref, c))
    errs["H"] = rel(H, g["H64"])
    assert_parity("qhnet_small H", H.cpu().numpy(), g["H64"], g["H32"])          # array-level and element-wise, yardstick = the reference's own fp32 run
    print("qhnet small, rel. error vs the reference in fp64:", {k: f"{v:.1e}" for k, v in errs.items()})
    print("reference fp32 vs fp64: H", rel(g["H32"], g["H64"]))
    assert max(errs.values()) < TOL, errs
    assert float((H - H.T).abs().max()) == 0.0                                  # H + H^T is exactly symmetric
    # keep_blocks variant (qhnet.py:239-252)
    with torch.no_grad():
        blocks = net(batch, keep_blocks=True)
    d = torch.tensor(g["inter64_diag_blocks"])
    nd = torch.tensor(g["inter64_nondiag_blocks"])
    from nabladft_amd.hamiltonian import transpose_index
    t = transpose_index(batch.ptr.cpu())
    assert rel(blocks["hamiltonian_diagonal_blocks"], d + d.transpose(-1, -2)) < TOL
    assert rel(blocks["hamiltonian_non_diagonal_blocks"], nd + nd[t].transpose(-1, -2)) < TOL


def _loss_and_grads(net, batch, target_dense):
    from nabladft_amd.hamiltonian import HamiltonianLoss
    net.zero_grad()
    Hp = net(batch, packed=True)
    plan = net.last_plan
    tgt = net._asm.from_dense(plan, torch.tensor(target_dense, dtype=torch.float32, device=Hp.device))
    loss = HamiltonianLoss()(Hp, tgt)
    loss.backward()
    return loss, {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in net.named_parameters()}


def test_small_loss_and_all_gradients():
    dev = torch.device("cuda:0")
    g, cfg, net, batch = load_case("small", dev)
    loss, grads = _loss_and_grads(net, batch, g["target"])
    assert abs(float(loss) - float(g["loss64"])) / float(g["loss64"]) < 1e-6
    unused = set(g["unused_params"].tolist())
    worst = {}
    for k, gr in grads.items():
        if k in unused:
            assert gr is None or float(gr.abs().max()) == 0.0, k          # Expansion.weights (layers.py:594-595): never read
            continue
        worst[k] = rel(gr, g["grad64_" + k])
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("qhnet small gradients: worst rel. errors", [(k, f"{v:.1e}") for k, v in top], "| reference fp32 vs fp64 worst", float(g["grad32_relerr"].max()))
    assert top[0][1] < 2e-5, top
    # dense return value carries the same gradient (the drop-in path: loss on the block_diag matrix with the mask)
    net.zero_grad()
    H = net(batch)
    tgt = torch.tensor(g["target"], dtype=torch.float32, device=dev)
    mask = (torch.block_diag(*[torch.ones(m, m) for m in np.diff(net.last_plan.mol_orb_ptr.cpu().numpy())])).to(dev)
    diff = H - tgt
    mse = torch.mean(diff ** 2)
    mae = torch.mean(torch.abs(diff))
    dense_loss = (mse * (mask.numel() / mask.sum())).sqrt() + mae * (mask.numel() / mask.sum())      # qhnet/loss.py:9-16
    dense_loss.backward()
    assert abs(float(dense_loss) - float(loss)) / float(loss) < 1e-6
    k = "e3_gnn_layer.0.conv.fc_node.layer1.weight"
    assert rel(dict(net.named_parameters())[k].grad, grads[k]) < 1e-5


def test_full_configuration():
    """config/model/qhnet.yaml sizes: hidden 128, bottleneck 32, 5 layers, 32 radial functions, cutoff 12."""
    dev = torch.device("cuda:0")
    from oracle.qhnet_params import probe_direction
    g, cfg, net, batch = load_case("full", dev)
    assert net.get_number_of_parameters() == 21891529
    inter = {}
    net.e3_gnn_layer[4].register_forward_hook(lambda mod, inp, out: inter.__setitem__("conv4", out.detach()))
    net.e3_gnn_node_layer[1].register_forward_hook(lambda mod, inp, out: inter.__setitem__("self1", out.detach()))
    net.e3_gnn_node_pair_layer[1].register_forward_hook(lambda mod, inp, out: inter.__setitem__("pair1", out.detach()))
    net.expand_ii["hamiltonian"].register_forward_hook(lambda mod, inp, out: inter.__setitem__("diag_blocks", out.detach()))
    net.expand_ij["hamiltonian"].register_forward_hook(lambda mod, inp, out: inter.__setitem__("nondiag_blocks", out.detach()))
    loss, grads = _loss_and_grads(net, batch, g["target"])
    with torch.no_grad():
        H = net(batch)
    assert torch.equal(batch.edge_index.cpu(), torch.tensor(g["edge_index"])) and torch.equal(batch.full_edge_index.cpu(), torch.tensor(g["full_edge_index"]))
    errs = {k: (rel(v, g[k]) if k.endswith("blocks") else rel(v, from_e3nn(g[k], 128))) for k, v in inter.items()}
    errs["H"] = rel(H, g["H64"])
    assert_parity("qhnet_full H", H.detach().cpu().numpy(), g["H64"], g["H32"])
    print("qhnet full, rel. error vs the reference in fp64:", {k: f"{v:.1e}" for k, v in errs.items()}, "| reference fp32:", rel(g["H32"], g["H64"]))
    assert max(errs.values()) < TOL, errs
    assert abs(float(loss) - float(g["loss64"])) / float(g["loss64"]) < 1e-6
    # gradients: projection of every tensor's gradient on a fixed random direction, relative to |g| |r| / sqrt(n) (the size of a random projection)
    worst = 0.0
    for k, n64, p64, p32 in zip(g["grad_names"], g["grad64_norm"], g["grad64_probe"], g["grad32_probe"]):
        gr = grads[str(k)].double()
        r = probe_direction(str(k), gr.shape, int(g["seed"]))
        scale = max(float(n64), 1e-300)
        e = abs(float((gr * r).sum()) - float(p64)) / scale
        assert abs(float(gr.norm()) - float(n64)) / scale < 2e-5, (k, float(gr.norm()), float(n64))
        worst = max(worst, e)
    print("qhnet full gradient projections: worst", worst)
    assert worst < 2e-5
    for k in g["unused_params"]:
        assert grads[str(k)] is None or float(grads[str(k)].abs().max()) == 0.0


def _rotation(seed):
    q, _ = np.linalg.qr(np.random.default_rng(seed).normal(size=(3, 3)))
    return q * np.sign(np.linalg.det(q))


def _wigner_blocks(Rm):
    """D_l with Y_l(R x) = D_l Y_l(x) for l = 0, 1, 2 in the model's real basis, fitted from the oracle's spherical harmonics (float64)."""
    from oracle import e3nn_mini as e3
    pts = torch.randn(200, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    sh = lambda v: e3.spherical_harmonics(e3.Irreps.spherical_harmonics(2), v[:, [1, 2, 0]], True, "component")
    A, B = sh(pts), sh(pts @ torch.tensor(Rm).T)
    return [torch.linalg.lstsq(A[:, l * l:(l + 1) ** 2], B[:, l * l:(l + 1) ** 2]).solution.T.numpy() for l in range(3)]


def test_rotation_equivariance_permutation_and_batching():
    """H(R x) = D(R) H(x) D(R)^T with D the block-diagonal orbital rotation (s, p, d shells of every atom); translation invariance; a molecule's
    block does not depend on what else is in the batch.  Full configuration sizes, random-init weights."""
    dev = torch.device("cuda:0")
    g, cfg, net, _ = load_case("full", dev)
    rng = np.random.default_rng(3)
    sizes = [9, 6]
    z = rng.choice([1, 1, 6, 7, 8, 16], size=sum(sizes))
    pos = np.concatenate([rng.normal(size=(n, 3)) * 2.5 for n in sizes]).astype(np.float32)
    with torch.no_grad():
        H = net(Batch(pos, z, sizes, dev)).cpu().double().numpy()
        Rm = _rotation(5)
        pos_r = (pos.astype(np.float64) @ Rm.T + np.array([0.3, -1.1, 0.7])).astype(np.float32)          # rotation + translation
        Hr = net(Batch(pos_r, z, sizes, dev)).cpu().double().numpy()
        H0 = net(Batch(pos[:9], z[:9], [9], dev)).cpu().double().numpy()
    D = _wigner_blocks(Rm)
    blocks = []
    for a in z:
        for l in ORBITALS[int(a)]:
            blocks.append(D[l])
    n = sum(b.shape[0] for b in blocks)
    Dfull = np.zeros((n, n))
    o = 0
    for b in blocks:
        Dfull[o:o + b.shape[0], o:o + b.shape[0]] = b
        o += b.shape[0]
    assert H.shape == (n, n)
    scale = np.abs(H).max()
    assert np.abs(Dfull @ H @ Dfull.T - Hr).max() / scale < 2e-5
    assert np.abs(H - H.T).max() == 0.0
    m0 = H0.shape[0]
    assert np.abs(H[:m0, :m0] - H0).max() / scale < 1e-5 and np.abs(H[:m0, m0:]).max() == 0.0


def test_output_shape_of_the_reference_dataset_sample():
    """tests/dataset/test_pyg_datasets.py:37-66 of the reference: Hamiltonian sample 0 has 38 atoms and 396 x 396 matrices (def2-SVP: 16 H, 20 second-row
    atoms, 2 S/Cl); tests/model/test_torch_models.py:55-62 asserts model(batch).shape == H.shape."""
    dev = torch.device("cuda:0")
    g, cfg, net, _ = load_case("full", dev)
    rng = np.random.default_rng(8)
    z = np.array([1] * 16 + [6] * 12 + [7] * 3 + [8] * 4 + [9] * 1 + [16, 17])
    rng.shuffle(z)
    pos = (rng.normal(size=(38, 3)) * 4.0).astype(np.float32)
    with torch.no_grad():
        H = net(Batch(pos, z, [38], dev))
    assert tuple(H.shape) == (396, 396) and bool(torch.isfinite(H).all())
    assert net.last_plan.m_total == 396


def test_lightning_wrapper_training_and_ema():
    """QHNetLightning (qhnet.py:345-536): constructor signature, training_step == HamiltonianLoss on the packed path == the reference's dense loss,
    EMA hooks (update after the optimiser step, validation under the averaged weights, averaged weights in the checkpoint), predict_step."""
    import functools
    import nabladft_amd as nq
    from nabladft_amd.ema import ExponentialMovingAverage
    from nabladft_amd.hamiltonian import HamiltonianLoss
    dev = torch.device("cuda:0")
    g, cfg, net, batch = load_case("small", dev)
    H64 = g["H64"]
    per_atom = np.array([sum(2 * l + 1 for l in ORBITALS[int(a)]) for a in g["z"]])
    ends = np.cumsum(g["sizes"])
    sizes_orb = [int(per_atom[e - n:e].sum()) for n, e in zip(g["sizes"], ends)]
    target = g["target"]
    o = 0
    batch.hamiltonian = []
    for m in sizes_orb:
        batch.hamiltonian.append(target[o:o + m, o:o + m].astype(np.float32))
        o += m
    task = nq.QHNetLightning("QHNet", net, functools.partial(torch.optim.AdamW, lr=1e-3, amsgrad=True, betas=(0.9, 0.95)), None,
                             {"hamiltonian": HamiltonianLoss()}, functools.partial(ExponentialMovingAverage, decay=0.5), None, {"hamiltonian": 1.0})
    task.on_fit_start()
    assert isinstance(task.ema, ExponentialMovingAverage)
    opt = task.configure_optimizers()["optimizer"]
    loss = task.training_step(batch, 0)
    assert abs(float(loss.detach()) - float(g["loss64"])) / float(g["loss64"]) < 1e-6
    loss.backward()
    before = [p.detach().clone() for p in task.parameters()]
    opt.step()
    task.on_before_zero_grad(opt)                           # EMA update (decay 0.5 -> first update uses min(0.5, 2/11))
    opt.zero_grad()
    d = min(0.5, 2.0 / 11.0)
    for p0, p1, s in zip(before, task.parameters(), task.ema.shadow_params):
        assert torch.allclose(s, p0 - (1 - d) * (p0 - p1.detach()), atol=1e-7)
    live = [p.detach().clone() for p in task.parameters()]
    vloss = task.validation_step(batch, 0)                  # evaluated with the averaged weights, live weights restored afterwards
    assert all(torch.equal(a, b.detach()) for a, b in zip(live, task.parameters()))
    with torch.no_grad():
        task.ema.store(); task.ema.copy_to()
        ref = task.step(batch)
        task.ema.restore()
    assert abs(float(vloss) - float(ref)) < 1e-7 * abs(float(ref))
    ckpt = {}
    task.on_save_checkpoint(ckpt)
    k0 = "net.node_embedding.weight"
    assert torch.equal(ckpt["state_dict"][k0], task.ema.shadow_params[0]) and not torch.equal(ckpt["state_dict"][k0], task.net.node_embedding.weight)
    # the reference's dense loss class shape: a loss without ``packed`` gets (pred, target, mask) on the block_diag matrices
    class DenseLoss(torch.nn.Module):
        def forward(self, pred, target, mask):
            diff = pred - target
            return (torch.mean(diff ** 2) * (pred.numel() / mask.sum())).sqrt() + torch.mean(torch.abs(diff)) * (pred.numel() / mask.sum())
    dense = nq.QHNetLightning("QHNet", net, None, None, {"hamiltonian": DenseLoss()}, None, None, {"hamiltonian": 1.0})
    with torch.no_grad():
        a, b = float(dense.step(batch)), float(task.step(batch))
    assert abs(a - b) < 1e-6 * abs(b)
    hs = task.predict_step(batch)
    assert [tuple(h.shape) for h in hs] == [(int(m), int(m)) for m in sizes_orb]
    assert task._get_hamiltonian_sizes(batch)[-1] == int(sum(sizes_orb))


@pytest.mark.gpu
def test_pair_generator_fusion_matches_the_materialised_path():
    """csrc/qhgen.hip (PairNetLayer weights generated inside the forward tensor-product kernel, layers.py:465-492) against the materialised path on the
    same network, inputs and parameters: the Hamiltonian and every parameter gradient agree to the accuracy of the two-piece bf16 split of the generator
    (the reverse sweep recomputes the factors with the dense engine either way); the fused kernel is seen by the profiler only when it is selected."""
    import nabladft_amd.qhnet as Q
    from nabladft_amd import _lib
    dev = torch.device("cuda:0")
    orbitals = {1: [0, 0, 1], 6: [0, 0, 0, 1, 1, 2], 8: [0, 0, 0, 1, 1, 2]}
    torch.manual_seed(3)
    net = Q.QHNet(in_node_features=1, sh_lmax=4, hidden_size=32, bottle_hidden_size=16, num_gnn_layers=5, max_radius=6.0, num_nodes=10, radius_embed_dim=16,
                  orbitals=orbitals).to(dev)
    rng = np.random.Generator(np.random.PCG64(11))
    pos = torch.tensor(rng.normal(size=(9, 3)) * 1.5, dtype=torch.float32, device=dev)
    z = torch.tensor([6, 1, 8, 1, 6, 6, 1, 8, 1], device=dev)

    class B:
        pass
    b = B()
    b.pos, b.z, b.batch, b.ptr = pos, z, torch.tensor([0, 0, 0, 0, 1, 1, 1, 1, 1], device=dev), torch.tensor([0, 4, 9], device=dev)
    out = {}
    lib = _lib.load()
    import ctypes as C
    for mode in (False, True):
        Q.set_pair_generator_fusion(mode)
        try:
            for p in net.parameters():
                p.grad = None
            names = C.create_string_buffer(256 * 64); ms = (C.c_double * 256)(); cnt = (C.c_int64 * 256)()
            lib.nq_profile_read(names, 64, ms, cnt, 256)
            lib.nq_profile_enable(1)
            H = net(b, packed=True)
            (H * torch.linspace(-1, 1, H.numel(), device=dev).view_as(H)).sum().backward()
            n = lib.nq_profile_read(names, 64, ms, cnt, 256)
            lib.nq_profile_enable(0)
            seen = {names.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode() for i in range(min(n, 256)) if cnt[i] > 0}
            assert ("qh_tp_uuu_fwd_gen" in seen) == mode and ("qh_tp_uuu_bwd_gen" in seen) == mode, seen
            out[mode] = (H.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        finally:
            Q.set_pair_generator_fusion(False)
    H0, g0 = out[False]
    H1, g1 = out[True]
    assert float((H0 - H1).abs().max() / H0.abs().max()) < 2e-5
    assert set(g0) == set(g1)
    for k in g0:
        d = float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30))
        assert d < 1e-4, (k, d)
