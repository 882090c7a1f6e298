"""Plugin surface: Hydra-style instantiation of the model config, Lightning-shaped step, energy-only
mode, inference modes.  CPU part checks construction; GPU part checks behaviour against the oracle."""
import os

import pytest
import torch
import yaml

from oracle import painn_ref as R
from tests.helpers import rel_err

MODEL_YAML = """
_target_: nabladft_amd.PaiNNLightning
model_name: "PAINN-OC"
model:
  _target_: nabladft_amd.PaiNN
  hidden_channels: 128
  num_layers: 6
  num_rbf: 100
  cutoff: 5.0
  max_neighbors: 100
  rbf: {name: 'gaussian'}
  envelope: {name: 'polynomial', exponent: 5}
  regress_forces: true
  direct_forces: false
  use_pbc: false
  otf_graph: true
  num_elements: 100
optimizer: {_target_: torch.optim.AdamW, _partial_: true, lr: 5.0e-4, weight_decay: 0}
lr_scheduler: {_target_: torch.optim.lr_scheduler.ReduceLROnPlateau, _partial_: true, factor: 0.8, patience: 100, min_lr: 1.0e-6}
losses:
  energy: {_target_: torch.nn.L1Loss}
  forces: {_target_: nabladft_amd.L2Loss}
loss_coefs: {energy: 1.0, forces: 1.0}
metric: null
"""


def _task():
    from nabladft_amd.config import instantiate
    return instantiate(yaml.safe_load(MODEL_YAML))


def test_config_instantiates_like_hydra():
    task = _task()
    import nabladft_amd as nq
    assert isinstance(task, nq.PaiNNLightning) and isinstance(task.model, nq.PaiNN)
    assert task.model.num_params == 1341313
    out = task.configure_optimizers()
    assert isinstance(out["optimizer"], torch.optim.AdamW) and out["optimizer"].defaults["lr"] == 5e-4
    assert out["lr_scheduler"]["monitor"] == "val_loss"
    # a Lightning checkpoint of the reference stores the model under "model." -- same here
    assert all(k.startswith("model.") for k in task.state_dict())
    assert len(task.state_dict()) == 72


@pytest.mark.gpu
def test_lightning_step_and_modes_on_gpu():
    import nabladft_amd as nq
    dev = torch.device("cuda:0")
    task = _task().to(dev)
    cfg = R.PaiNNConfig()
    params = R.make_params(cfg, seed=23)
    task.model.load_state_dict(params, strict=False)
    pos, z, batch, y, ft = R.gen_conformers(21, 3)
    b = nq.Batch(pos, z, batch, y, ft).to(dev)
    e_ref, f_ref, loss_ref, g_ref = R.train_step(params, cfg, pos, z, batch, y, ft)
    # training_step == reference loss; backward fills .grad of every parameter
    loss = task.training_step(b, 0)
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    loss.backward()
    worst = max(rel_err(p.grad.cpu().numpy(), g_ref[k[len("model."):]].numpy()) for k, p in task.named_parameters())
    assert worst < 5e-5, worst
    # a torch optimizer steps the flat-buffer views in place
    opt = task.configure_optimizers()["optimizer"]
    ptr0, before = task.model.flat_parameters().data_ptr(), task.model.flat_parameters().clone()
    opt.step()
    assert task.model.flat_parameters().data_ptr() == ptr0 and not torch.equal(task.model.flat_parameters(), before)
    task.model.load_state_dict(params, strict=False)
    # validation/predict: no_grad and inference_mode give the same energies and forces (forces do not need autograd)
    task.eval()
    with torch.no_grad():
        e1, f1 = task(b)
    with torch.inference_mode():
        e2, f2 = task.predict_step(b)
    assert torch.equal(e1, e2) and torch.equal(f1, f2)
    assert rel_err(e1.cpu().numpy(), e_ref.numpy()) < 1e-5 and rel_err(f1.cpu().numpy(), f_ref.numpy()) < 1e-5


@pytest.mark.gpu
def test_energy_only_model_trains():
    """regress_forces=False (painn.py:147-148): forward returns the energy only; backward = first-order gradients."""
    import nabladft_amd as nq
    dev = torch.device("cuda:0")
    cfg = R.PaiNNConfig(hidden_channels=64, num_layers=2, num_rbf=20)
    params = R.make_params(cfg, seed=9)
    m = nq.PaiNN(64, 2, 20, 5.0, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, False, False, False, True, 100)
    m.load_state_dict(params, strict=False)
    m.to(dev)
    pos, z, batch, y, _ = R.gen_conformers(31, 4, size=(6, 18))
    energy = m(nq.Batch(pos, z, batch).to(dev))
    assert energy.shape == (4,)
    (energy * torch.tensor([1.0, -2.0, 0.5, 3.0], device=dev)).sum().backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ei, _, _ = R.build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    e_ref = R.painn_energy(Pg, cfg, pos, z, batch, ei)
    assert rel_err(energy.detach().cpu().numpy(), e_ref.detach().numpy()) < 1e-5
    g_ref = torch.autograd.grad((e_ref * torch.tensor([1.0, -2.0, 0.5, 3.0])).sum(), list(Pg.values()))
    for (k, p), g in zip(m.named_parameters(), g_ref):
        assert rel_err(p.grad.cpu().numpy(), g.numpy()) < 5e-5, k


@pytest.mark.gpu
def test_workspace_is_reused_after_forwards_without_backward():
    """optimization/calculator.py:124-130 calls model(batch) under grad mode and never runs backward: the cached workspace must be handed out
    again as soon as that forward's outputs are gone (ownership by the autograd node, not a sticky flag), and an unfinished forward whose
    outputs are still alive must not be clobbered."""
    import nabladft_amd as nq
    dev = torch.device("cuda:0")
    m = nq.PaiNN(64, 2, 20, 5.0, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100).to(dev)
    pos, z, batch, y, ft = R.gen_conformers(5, 3, size=(6, 14))
    b = nq.Batch(pos, z, batch, y, ft).to(dev)
    e, f = m(b)
    ws0 = m._last_ws.data_ptr()
    e_keep = e.detach().clone()
    del e, f                                            # the only references to the autograd node
    for _ in range(3):
        e, f = m(b)
        assert m._last_ws.data_ptr() == ws0             # the one cached buffer, no fresh multi-GB allocation per call
        del e, f
    e1, f1 = m(b)                                       # outputs alive -> its backward is still possible
    assert m._last_ws.data_ptr() == ws0
    e2, f2 = m(b)
    assert m._last_ws.data_ptr() != ws0                 # a second workspace instead of overwriting the first one's activations
    (e1.sum() + f1.sum()).backward()                    # ... which is still intact
    g1 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    del e2, f2
    e3, f3 = m(b)
    assert m._last_ws.data_ptr() == ws0 and torch.equal(e3.detach(), e_keep)
    (e3.sum() + f3.sum()).backward()
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, m.parameters()))


@pytest.mark.gpu
def test_fused_step_coerces_and_checks_targets():
    import nabladft_amd as nq
    dev = torch.device("cuda:0")
    m = nq.PaiNN(64, 2, 20, 5.0, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100).to(dev)
    pos, z, batch, y, ft = R.gen_conformers(6, 3, size=(6, 14))
    step = nq.FusedTrainStep(m, max_grad_norm=0.0)
    ref = float(step(nq.Batch(pos, z, batch, y, ft).to(dev), update=False))
    b64 = nq.Batch(pos, z, batch, y.double(), ft.double().t().contiguous().t()).to(dev)     # float64, non-contiguous forces
    assert abs(float(step(b64, update=False)) - ref) < 1e-6 * abs(ref)
    with pytest.raises(ValueError):
        step(nq.Batch(pos, z, batch, y, None).to(dev), update=False)
    with pytest.raises(ValueError):
        step(nq.Batch(pos, z, batch, y[:-1], ft).to(dev), update=False)


QHNET_YAML = """
_target_: nabladft_amd.QHNetLightning
model_name: "QHNet"
net:
  _target_: nabladft_amd.QHNet
  _convert_: partial
  sh_lmax: 4
  hidden_size: 128
  bottle_hidden_size: 32
  num_gnn_layers: 5
  max_radius: 12
  num_nodes: 83
  radius_embed_dim: 32
  orbitals:
    1: [0, 0, 1]
    6: [0, 0, 0, 1, 1, 2]
    7: [0, 0, 0, 1, 1, 2]
    8: [0, 0, 0, 1, 1, 2]
    9: [0, 0, 0, 1, 1, 2]
    16: [0, 0, 0, 0, 1, 1, 1, 2]
    17: [0, 0, 0, 0, 1, 1, 1, 2]
    35: [0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]
optimizer: {_target_: torch.optim.AdamW, _partial_: true, amsgrad: true, betas: [0.9, 0.95], lr: 5.0e-4}
lr_scheduler: {_target_: torch.optim.lr_scheduler.ReduceLROnPlateau, _partial_: true, factor: 0.8, patience: 10, min_lr: 1.0e-6}
losses:
  hamiltonian: {_target_: nabladft_amd.hamiltonian.HamiltonianLoss}
loss_coefs: {hamiltonian: 1.0}
metric: null
ema: {_target_: nabladft_amd.ema.ExponentialMovingAverage, _partial_: true, decay: 0.9999}
"""

SPK_TASK_YAML = """
_target_: nabladft_amd.AtomisticTaskFixed
model_name: "PaiNN"
model:
  _target_: nabladft_amd.spk.NeuralNetworkPotential
  representation:
    _target_: nabladft_amd.spk.PaiNN
    n_interactions: 2
    n_atom_basis: 64
    radial_basis: {_target_: nabladft_amd.spk.GaussianRBF, n_rbf: 20, cutoff: 5.0}
    cutoff_fn: {_target_: nabladft_amd.spk.CosineCutoff, cutoff: 5.0}
  input_modules: [{_target_: nabladft_amd.spk.PairwiseDistances}]
  output_modules:
    - {_target_: nabladft_amd.spk.Atomwise, n_in: 64, output_key: "energy"}
    - {_target_: nabladft_amd.spk.Forces}
  postprocessors: [{_target_: nabladft_amd.spk.AddOffsets, property: "energy", add_mean: True}]
outputs:
  - {_target_: nabladft_amd.ModelOutput, name: "energy", loss_fn: {_target_: torch.nn.MSELoss}, loss_weight: 1.0}
  - {_target_: nabladft_amd.ModelOutput, name: "forces", loss_fn: {_target_: torch.nn.MSELoss}, loss_weight: 1.0}
optimizer_cls: {_partial_: True, _target_: torch.optim.AdamW}
optimizer_args: {lr: 1.0e-4}
scheduler_cls: {_partial_: True, _target_: torch.optim.lr_scheduler.ReduceLROnPlateau}
scheduler_args: {factor: 0.8, patience: 10, min_lr: 1.0e-6}
scheduler_monitor: val_loss
"""


def test_qhnet_and_spk_task_configs_instantiate():
    """config/model/qhnet.yaml and config/model/painn.yaml (schnetpack task) with the `_target_` lines pointed at this package."""
    import nabladft_amd as nq
    from nabladft_amd.config import instantiate
    task = instantiate(yaml.safe_load(QHNET_YAML))
    assert isinstance(task, nq.QHNetLightning) and isinstance(task.net, nq.QHNet)
    assert task.net.get_number_of_parameters() == 21891529 and all(k.startswith("net.") for k in task.state_dict())
    opt = task.configure_optimizers()
    assert opt["optimizer"].defaults["amsgrad"] and opt["lr_scheduler"]["monitor"] == "val_loss"
    task._instantiate_ema()
    assert isinstance(task.ema, nq.ema.ExponentialMovingAverage) and task.ema.decay == 0.9999
    spk_task = instantiate(yaml.safe_load(SPK_TASK_YAML))
    assert isinstance(spk_task, nq.AtomisticTaskFixed) and len(spk_task.outputs) == 2 and spk_task.hparams.model_name == "PaiNN"
    optimizers, schedulers = spk_task.configure_optimizers()
    assert isinstance(optimizers[0], torch.optim.AdamW) and schedulers[0]["monitor"] == "val_loss"
    ck = {"state_dict": {"model.postprocessors.0.mean": torch.tensor(3.0)}}
    spk_task.on_save_checkpoint(ck)
    assert tuple(ck["state_dict"]["model.postprocessors.0.mean"].shape) == (1,)


GEMNET_YAML = """
_target_: nabladft_amd.GemNetOCLightning
model_name: "GemNet-OC"
net:
  _target_: nabladft_amd.GemNetOC
  num_targets: 1
  num_spherical: 7
  num_radial: 128
  num_blocks: 4
  emb_size_atom: 256
  emb_size_edge: 512
  emb_size_trip_in: 64
  emb_size_trip_out: 64
  emb_size_quad_in: 32
  emb_size_quad_out: 32
  emb_size_aint_in: 64
  emb_size_aint_out: 64
  emb_size_rbf: 16
  emb_size_cbf: 16
  emb_size_sbf: 32
  num_before_skip: 2
  num_after_skip: 2
  num_concat: 1
  num_atom: 3
  num_output_afteratom: 3
  num_atom_emb_layers: 0
  num_global_out_layers: 2
  regress_forces: true
  direct_forces: true
  use_pbc: false
  scale_backprop_forces: false
  cutoff: 12.0
  cutoff_qint: 12.0
  cutoff_aeaint: 12.0
  cutoff_aint: 12.0
  max_neighbors: 30
  max_neighbors_qint: 8
  max_neighbors_aeaint: 20
  max_neighbors_aint: 1000
  enforce_max_neighbors_strictly: true
  rbf: {name: gaussian}
  rbf_spherical: null
  envelope: {name: polynomial, exponent: 5}
  cbf: {name: spherical_harmonics}
  sbf: {name: legendre_outer}
  extensive: true
  forces_coupled: true
  output_init: HeOrthogonal
  activation: silu
  scale_file: null
  quad_interaction: true
  atom_edge_interaction: true
  edge_atom_interaction: true
  atom_interaction: true
  scale_basis: true
optimizer: {_target_: torch.optim.AdamW, _partial_: true, amsgrad: true, betas: [0.9, 0.95], lr: 1.0e-3, weight_decay: 0}
lr_scheduler: {_target_: torch.optim.lr_scheduler.ReduceLROnPlateau, _partial_: true, factor: 0.8, patience: 10}
losses:
  energy: {_target_: torch.nn.L1Loss}
  forces: {_target_: nabladft_amd.L2Loss}
loss_coefs: {energy: 1.0, forces: 100.0}
metric: null
"""


def test_gemnet_oc_config_instantiates_with_the_reference_state_dict_surface():
    """config/model/gemnet-oc.yaml with the `_target_` lines pointed at this package: parameter count, state_dict keys (order included, aliases of the
    shared modules included) as recorded from the reference class in tests/golden/gemnet_full.npz, unsupported options fail loudly."""
    import numpy as np
    import nabladft_amd as nq
    from nabladft_amd.config import instantiate
    cfg = yaml.safe_load(GEMNET_YAML)
    task = instantiate(cfg)
    assert isinstance(task, nq.GemNetOCLightning) and isinstance(task.net, nq.GemNetOC)
    assert task.net.num_params == 37815873
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "gemnet_full.npz"))
    assert list(task.net.state_dict().keys()) == list(gold["state_keys"])
    assert [n for n, _ in task.net.named_parameters()] == list(gold["param_names"])
    sd = task.net.state_dict()
    assert sd["out_blocks.0.seq_energy_pre.0.linear.weight"].data_ptr() == sd["out_blocks.0.layers.0.linear.weight"].data_ptr()
    assert sd["cbf_basis_tint.radial_basis.rbf.offset"].data_ptr() == sd["sbf_basis_qint.radial_basis.rbf.offset"].data_ptr()
    assert not any(p.requires_grad for n, p in task.net.named_parameters() if n.endswith("scale_factor"))
    opt = task.configure_optimizers()
    assert opt["optimizer"].defaults["amsgrad"] and opt["optimizer"].defaults["betas"] == (0.9, 0.95) and opt["lr_scheduler"]["monitor"] == "val_loss"
    loss = task._calculate_loss({"energy": torch.zeros(2), "forces": torch.zeros(5, 3)}, {"energy": torch.ones(2), "forces": torch.ones(5, 3)})
    assert abs(float(loss) - (1.0 + 100.0 * 3 ** 0.5)) < 1e-4
    for key, bad in (("use_pbc", True), ("direct_forces", False), ("sbf", {"name": "spherical_harmonics"}), ("rbf", {"name": "spherical_bessel"})):
        c = dict(cfg["net"]); c.pop("_target_"); c[key] = bad
        with pytest.raises(NotImplementedError):
            nq.GemNetOC(**c)
    with pytest.raises(RuntimeError, match="MI355X"):                   # no CPU path
        d = type("D", (), {})()
        d.pos, d.z, d.batch = torch.zeros(3, 3), torch.ones(3, dtype=torch.long), torch.zeros(3, dtype=torch.long)
        task.net(d)


ESCN_YAML = """
_target_: nabladft_amd.eSCNLightning
model_name: "ESCN-OC"
net:
  _target_: nabladft_amd.eSCN
  num_targets: 1
  max_num_elements: 65
  num_layers: 8
  lmax_list: [6]
  mmax_list: [2]
  sphere_channels: 128
  hidden_channels: 256
  edge_channels: 128
  use_grid: true
  num_sphere_samples: 128
  distance_function: gaussian
  regress_forces: true
  otf_graph: true
  use_pbc: false
  cutoff: 8.0
  max_neighbors: 40
  basis_width_scalar: 1.0
  distance_resolution: 0.02
  show_timing_info: false
optimizer: {_target_: torch.optim.AdamW, _partial_: true, amsgrad: true, betas: [0.9, 0.95], lr: 1.0e-3, weight_decay: 0}
lr_scheduler: {_target_: torch.optim.lr_scheduler.ReduceLROnPlateau, _partial_: true, factor: 0.8, patience: 10}
losses:
  energy: {_target_: torch.nn.L1Loss}
  forces: {_target_: nabladft_amd.L2Loss}
loss_coefs: {energy: 1.0, forces: 100.0}
metric: null
"""


def test_escn_config_instantiates():
    """config/model/escn-oc.yaml with the `_target_` lines pointed at this package."""
    import nabladft_amd as nq
    from nabladft_amd.config import instantiate
    task = instantiate(yaml.safe_load(ESCN_YAML))
    assert isinstance(task, nq.eSCNLightning) and isinstance(task.net, nq.eSCN) and task.net.num_params == 34332032
    assert all(k.startswith("net.") for k in task.state_dict())
    assert task.configure_optimizers()["optimizer"].defaults["amsgrad"]
    with pytest.raises(RuntimeError, match="MI355X"):
        d = type("D", (), {})()
        d.pos, d.z, d.batch = torch.zeros(3, 3), torch.ones(3, dtype=torch.long), torch.zeros(3, dtype=torch.long)
        task.net(d)


EQUIFORMER_YAML = """
_target_: nabladft_amd.EquiformerV2_OC20_Lightning
model_name: "Equiformer-v2"
net:
  _target_: nabladft_amd.EquiformerV2_OC20
  otf_graph: true
  num_layers: 12
  sphere_channels: 128
  attn_hidden_channels: 64
  num_heads: 8
  attn_alpha_channels: 64
  attn_value_channels: 16
  ffn_hidden_channels: 128
  norm_type: 'layer_norm_sh'
  lmax_list: [6]
  mmax_list: [2]
  num_sphere_samples: 128
  edge_channels: 128
  use_atom_edge_embedding: true
  share_atom_edge_embedding: false
  distance_function: 'gaussian'
  num_distance_basis: 512
  attn_activation: 'silu'
  use_s2_act_attn: false
  use_attn_renorm: true
  ffn_activation: 'silu'
  use_gate_act: false
  use_grid_mlp: true
  use_sep_s2_act: true
  alpha_drop: 0.1
  drop_path_rate: 0.05
  proj_drop: 0.0
  weight_init: 'uniform'
  regress_forces: true
  use_pbc: false
  max_neighbors: 30
  max_radius: 12.0
  max_num_elements: 65
optimizer: {_target_: torch.optim.AdamW, _partial_: true, lr: 0.0004, weight_decay: 0.001}
lr_scheduler:
  _target_: torch.optim.lr_scheduler.LambdaLR
  _partial_: true
  lr_lambda:
    _target_: nabladft_amd.equiformer_v2.CosineLRLambda
    scheduler_params: {warmup_factor: 0.2, warmup_epochs: 0.1, epochs: 1000, lr_min_factor: 0.01}
losses:
  energy: {_target_: torch.nn.L1Loss}
  forces: {_target_: nabladft_amd.L2Loss}
loss_coefs: {energy: 2.0, forces: 100.0}
metric: null
"""


def test_equiformer_config_instantiates():
    """config/model/equiformer_v2_oc20.yaml with the `_target_` lines pointed at this package."""
    import nabladft_amd as nq
    from nabladft_amd.config import instantiate
    task = instantiate(yaml.safe_load(EQUIFORMER_YAML))
    assert isinstance(task, nq.EquiformerV2_OC20_Lightning) and isinstance(task.net, nq.EquiformerV2_OC20) and task.net.num_params == 83072002
    assert all(k.startswith("net.") for k in task.state_dict())
    opt = task.configure_optimizers()
    assert opt["optimizer"].defaults["weight_decay"] == 0.001
    sched = opt["lr_scheduler"]["scheduler"]
    lam = sched.lr_lambdas[0]
    assert abs(lam(0) - 0.2) < 1e-12 and abs(lam(500) - (0.01 + 0.5 * 0.99)) < 1e-12 and lam(1000) == 0.01
    with pytest.raises(RuntimeError, match="MI355X"):
        d = type("D", (), {})()
        d.pos, d.z, d.batch = torch.zeros(3, 3), torch.ones(3, dtype=torch.long), torch.zeros(3, dtype=torch.long)
        task.net(d)
