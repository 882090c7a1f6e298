"""EquiformerV2 (second half) on the MI355X against golden vectors of the REAL reference classes (oracle/make_golden_equiformer.py; four e3nn symbols
under them are restated, parity unpinned for those): graph bit-exact, Wigner rows, the embedding after every stage, E, F and all gradients vs the fp64 run."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
from oracle.equiformer_params import make_state, probe_direction  # noqa: E402
from tests.test_equiformer_cpu import FULL, SMALL  # noqa: E402
from tests.test_escn_gpu import Data, rel  # noqa: E402
from tests.helpers import assert_parity  # noqa: E402


def build(cfg, d, dev):
    from nabladft_amd.equiformer_v2 import EquiformerV2_OC20
    net = EquiformerV2_OC20(**cfg)
    names = [(k, tuple(v.shape)) for k, v in net.named_parameters() if v.requires_grad]
    assert [n for n, _ in names] == list(d["param_names"])
    assert not net.load_state_dict(make_state(names, int(d["seed"])), strict=False).unexpected_keys
    return net.to(dev).eval()                     # eval(): attention dropout and drop-path are random in training mode (the fixtures were made in eval mode)


def _loss(E, F, data):
    return 2.0 * (E - data.y).abs().mean() + 100.0 * torch.linalg.vector_norm(F - data.forces, dim=-1).mean()      # config/model/equiformer_v2_oc20.yaml:57-64


def test_graph_wigner_and_stages_small():
    d = np.load(os.path.join(GOLD, "equiformer_small.npz"))
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev)
    data = Data(d, dev)
    with torch.no_grad():
        E, F, rec, G = net(data, edge_rot_mat=torch.tensor(d["edge_rot_mat"]), return_intermediates=True)
    assert np.array_equal(np.stack([G.src.cpu().numpy(), G.dst.cpu().numpy()]), d["edge_index"])           # radius_graph with the cap of 5 binding
    o = net._order
    W = G.wigner.cpu().numpy().reshape(G.E, o.n_red, o.n_full)
    assert np.abs(W - d["wigner"][:, o.red_m_primary, :]).max() < 5e-6
    for k in ("embed", "norm1", "ga", "block0", "block1"):
        ref64 = d["f64:" + k].reshape(G.N, -1)
        own = rel(d["f32:" + k].reshape(G.N, -1), ref64)
        assert rel(rec[k].cpu().numpy(), ref64) < max(2e-5, 3 * own), k
    assert_parity("equiformer_small E", E.cpu().numpy(), d["f64:E"], d["f32:E"])
    assert_parity("equiformer_small F", F.cpu().numpy(), d["f64:F"], d["f32:F"])


def test_gradients_small():
    d = np.load(os.path.join(GOLD, "equiformer_small.npz"))
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev)
    data = Data(d, dev)
    E, F = net(data, edge_rot_mat=torch.tensor(d["edge_rot_mat"]))
    loss = _loss(E, F, data)
    loss.backward()
    assert abs(float(loss.detach()) - float(d["f64:loss"])) < 2e-5 * abs(float(d["f64:loss"]))
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        ref64, ref32 = d["f64:grad:" + name], d["f32:grad:" + name]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref64)
        scale = max(np.abs(ref64).max(), 1e-30)
        err, own = np.abs(g - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
        assert err <= max(5e-5, 3 * own), (name, err, own)


def test_full_configuration():
    d = np.load(os.path.join(GOLD, "equiformer_full.npz"))
    dev = torch.device("cuda:0")
    net = build(FULL, d, dev)
    data = Data(d, dev)
    E, F, rec, G = net(data, edge_rot_mat=torch.tensor(d["edge_rot_mat"]), return_intermediates=True)
    assert np.array_equal(np.stack([G.src.cpu().numpy(), G.dst.cpu().numpy()]), d["edge_index"])           # the cap of 30 binds on the 46-atom molecule
    C = FULL["sphere_channels"]
    for k in ("embed", "block0", "block11"):
        got = rec[k].detach().cpu().numpy().reshape(G.N, -1, C)[::5, :, ::8]
        assert rel(got, d["f32:" + k]) < 5e-5, k
    assert_parity("equiformer_full E", E.detach().cpu().numpy(), d["f64:E"], d["f32:E"])
    assert_parity("equiformer_full F", F.detach().cpu().numpy(), d["f64:F"], d["f32:F"])
    loss = _loss(E, F, data)
    loss.backward()
    assert_parity("equiformer_full loss", float(loss.detach()), d["f64:loss"], d["f32:loss"])
    names = list(d["param_names"])
    n64, p64, p32 = d["f64:grad_norm"], d["f64:grad_probe"], d["f32:grad_probe"]
    params = dict(net.named_parameters())
    for i, name in enumerate(names):
        g = params[name].grad.double().cpu()
        assert abs(float(g.norm()) - n64[i]) <= 2e-4 * max(n64[i], 1e-12), (name, float(g.norm()), n64[i])
        probe = float((g * probe_direction(name, g.shape, int(d["seed"]))).sum())
        assert abs(probe - p64[i]) <= max(1e-4 * n64[i] * np.sqrt(g.numel()) * 0.05, 3 * abs(p32[i] - p64[i])), (name, probe, p64[i])


def test_own_frames_training_mode_and_reproducibility():
    """The deterministic frames of the graph stage against the CPU restatement run with the SAME frames (EquiformerV2's output depends on the frame angle, see
    tests/test_equiformer_cpu.py); drop-path / attention dropout in training mode; bitwise reproducibility."""
    from oracle import equiformer_ref as R
    d = np.load(os.path.join(GOLD, "equiformer_small.npz"))
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev)
    data = Data(d, dev)
    with torch.no_grad():
        E0, F0, _, G = net(data, return_intermediates=True)
        E1, F1 = net(data)
    assert torch.equal(E0, E1) and torch.equal(F0, F1)
    P = {k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        Er, Fr = R.forward(P, SMALL, data.pos.cpu().double(), data.z.cpu(), d["sizes"].tolist(), rot=G.rot.cpu().double())
    assert rel(E0.cpu().numpy(), Er.numpy()) < 2e-5 and rel(F0.cpu().numpy(), Fr.numpy()) < 2e-5
    net.train()
    torch.manual_seed(3)
    Et, Ft = net(data, edge_rot_mat=G.rot)
    _loss(Et, Ft, data).backward()
    assert torch.isfinite(Et).all() and torch.isfinite(Ft).all() and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert not torch.equal(Et.detach(), E0)                                  # the random masks are active


def test_invalid_inputs_fail_loudly():
    d = np.load(os.path.join(GOLD, "equiformer_small.npz"))
    dev = torch.device("cuda:0")
    net = build(SMALL, d, dev)
    data = Data(d, dev)
    data.z = data.z.clone()
    data.z[0] = 77
    with pytest.raises(IndexError):
        net(data)
