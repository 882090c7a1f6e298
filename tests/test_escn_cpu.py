"""CPU side of the eSCN row (f4): the constants nabladft_amd/escn.py derives from first principles against the reference's data file (Jd.pt, read by the
fixture generator) and against the grids / orderings the REAL reference classes produced (tests/golden/escn_small.npz); state_dict surface."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
SMALL = dict(num_targets=1, use_pbc=False, regress_forces=True, otf_graph=True, use_grid=True, distance_function="gaussian", basis_width_scalar=1.0,
             show_timing_info=False, max_neighbors=5, cutoff=4.0, max_num_elements=40, num_layers=3, lmax_list=[3], mmax_list=[2], sphere_channels=16,
             hidden_channels=32, edge_channels=16, num_sphere_samples=32, distance_resolution=0.25)
FULL = dict(num_targets=1, use_pbc=False, regress_forces=True, otf_graph=True, use_grid=True, distance_function="gaussian", basis_width_scalar=1.0,
            show_timing_info=False, max_neighbors=40, cutoff=8.0, max_num_elements=65, num_layers=8, lmax_list=[6], mmax_list=[2], sphere_channels=128,
            hidden_channels=256, edge_channels=128, num_sphere_samples=128, distance_resolution=0.02)          # config/model/escn-oc.yaml:5-25


def test_j_matrices_equal_the_reference_data_file():
    from nabladft_amd.escn import j_matrices
    d = np.load(os.path.join(GOLD, "escn_small.npz"))
    for l, J in enumerate(j_matrices(6)):
        assert np.abs(J - d[f"Jd:{l}"]).max() < 1e-10, l            # pins this package's e3nn-convention harmonics up to l = 6 to e3nn's own Wigner data


def test_grids_orderings_and_sphere_samples_match_the_reference_run():
    from nabladft_amd.escn import CoefficientOrder, eSCN, s2_grids
    d = np.load(os.path.join(GOLD, "escn_small.npz"))
    for mm in (3, 2):
        T, F = s2_grids(3, mm)
        o = CoefficientOrder(3, mm)
        ref_T, ref_F = d[f"to_grid_3_{mm}"], d[f"from_grid_3_{mm}"]                                  # [lat, long, reduced coefficients (l-primary)]
        assert np.abs(T[:, o.red_l_primary] - ref_T.reshape(-1, ref_T.shape[-1])).max() < 2e-7
        assert np.abs(F[:, o.red_l_primary] - ref_F.reshape(-1, ref_F.shape[-1])).max() < 2e-7
        assert np.abs(F.T @ T - np.eye(16))[np.ix_(o.red_l_primary, o.red_l_primary)].max() < 1e-12  # exact quadrature on the band-limited space
    net = eSCN(**SMALL)
    assert list(net.state_dict().keys()) == list(d["state_keys"])
    assert [",".join(map(str, v.shape)) for v in net.state_dict().values()] == list(d["state_shapes"])
    assert np.abs(net.sphere_points.numpy() - d["state:sphere_points"]).max() < 1e-6
    assert np.abs(net.sphharm_weights[0].numpy() - d["state:sphharm_weights.0"]).max() < 2e-6
    assert [n for n, p in net.named_parameters() if p.requires_grad] == list(d["param_names"])


def test_full_configuration_surface():
    from nabladft_amd.escn import eSCN
    d = np.load(os.path.join(GOLD, "escn_full.npz"))
    net = eSCN(**FULL)
    assert net.num_params == 34332032 and list(net.state_dict().keys()) == list(d["state_keys"])
    import pytest
    with pytest.raises(NotImplementedError):
        eSCN(**dict(FULL, use_pbc=True))
    with pytest.raises(NotImplementedError):
        eSCN(**dict(FULL, lmax_list=[4, 2], mmax_list=[2, 2]))


def test_oracle_restatement_matches_the_reference_fixtures():
    """oracle/escn_ref.py (checker of smoke / cpu_baseline) against the golden vectors of the real classes: fp64 to round-off with the fixture's frames,
    and the deterministic frames give the same function up to eSCN's own grid-sampling error."""
    from oracle import escn_ref as R
    d = np.load(os.path.join(GOLD, "escn_small.npz"))
    P = {k[6:]: torch.tensor(d[k]).double() for k in d.files if k.startswith("state:")}
    P["distance_expansion.offset"] = torch.linspace(0.0, SMALL["cutoff"], int(SMALL["cutoff"] / SMALL["distance_resolution"]), dtype=torch.float64)
    P["sphere_points"], P["sphharm_weights.0"] = R.sphere_constants(SMALL, torch.float64)         # built in the run's dtype by the reference
    train = list(d["param_names"])
    for k in train:
        P[k].requires_grad_(True)
    pos, z = torch.tensor(d["pos"]).double(), torch.tensor(d["z"])
    E, F = R.forward(P, SMALL, pos, z, list(d["sizes"]), rot=torch.tensor(d["edge_rot_mat"]).double())
    assert np.abs(E.detach().numpy() - d["f64:E"]).max() < 1e-6 * np.abs(d["f64:E"]).max()
    assert np.abs(F.detach().numpy() - d["f64:F"]).max() < 1e-6 * np.abs(d["f64:F"]).max()
    L = R.loss(E, F, torch.tensor(d["y"]).double(), torch.tensor(d["f_target"]).double())
    L.backward()
    for k in train:
        ref = d["f64:grad:" + k]
        g = np.zeros_like(ref) if P[k].grad is None else P[k].grad.numpy()
        assert np.abs(g - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-30), k
    with torch.no_grad():
        E2, F2 = R.forward(P, SMALL, pos, z, list(d["sizes"]))
    assert np.abs(F2.numpy() - d["f64:F"]).max() < 3e-4 * np.abs(d["f64:F"]).max()
