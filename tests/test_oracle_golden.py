"""CPU: the oracle restatement (oracle/painn_ref.py) against the golden vectors produced by
running the real reference (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import painn_ref as R
from tests.helpers import GOLDEN, check_grads, load_case, rel_err


def test_graph_cases_bit_exact():
    gx = np.load(GOLDEN + "/graph_cases.npz")
    for c in range(int(gx["n_cases"])):
        pre = f"c{c}_"
        pos, batch = torch.tensor(gx[pre + "pos"]), torch.tensor(gx[pre + "batch"])
        ei, nb, sw = R.build_graph(pos, batch, float(gx[pre + "cutoff"]), int(gx[pre + "K"]))
        assert np.array_equal(ei.numpy(), gx[pre + "edge_index"]), f"case {c}"
        assert np.array_equal(nb.numpy(), gx[pre + "neighbors"]), f"case {c}"
        assert np.array_equal(sw.numpy(), gx[pre + "id_swap"]), f"case {c}"
        d, v = R.edge_geometry(pos, ei)
        assert np.array_equal(d.numpy(), gx[pre + "edge_dist"]), f"case {c}"
        assert np.array_equal(v.numpy(), gx[pre + "edge_vector"]), f"case {c}"


def test_six_atom_known_answer():
    """Layout documented in SURVEY.md 8(a3): kept edges (i asc, j asc) then their flips."""
    gx = np.load(GOLDEN + "/graph_cases.npz")
    ei, sw = gx["c0_edge_index"], gx["c0_id_swap"]
    h = ei.shape[1] // 2
    assert (ei[0, :h] < ei[1, :h]).all()
    assert np.array_equal(ei[:, h:], ei[::-1, :h])
    assert np.array_equal(sw, np.concatenate([np.arange(h) + h, np.arange(h)]))


@pytest.mark.parametrize("name,tol", [("painn_small_ragged.npz", 2e-5), ("painn_full_real4.npz", 2e-5), ("painn_small_expenv.npz", 2e-5), ("painn_small_bessel.npz", 2e-5), ("painn_small_bernstein.npz", 2e-5), ("painn_small_direct.npz", 2e-5)])
def test_train_step_matches_reference(name, tol):
    fx, cfg, params = load_case(name)
    pos, z, batch = torch.tensor(fx["pos"]), torch.tensor(fx["z"]), torch.tensor(fx["batch"])
    ei, nb, sw = R.build_graph(pos, batch, cfg.cutoff, cfg.max_neighbors)
    assert np.array_equal(ei.numpy(), fx["edge_index"])
    assert np.array_equal(nb.numpy(), fx["neighbors"])
    assert np.array_equal(sw.numpy(), fx["id_swap"])
    trace = {}
    e_eval, f_eval = R.energy_forces(params, cfg, pos, z, batch, ei, trace=trace)
    assert np.array_equal(trace["edge_dist"].numpy(), fx["edge_dist"])
    assert np.array_equal(trace["edge_vector"].numpy(), fx["edge_vector"])
    assert rel_err(trace["edge_rbf"].sum(0).numpy(), fx["edge_rbf_sum"]) < 1e-6
    L = cfg.num_layers
    assert rel_err(trace["x_msg0"].numpy(), fx["x_msg0"]) < 1e-6
    assert rel_err(trace[f"x_upd{L-1}"].numpy(), fx[f"x_upd{L-1}"]) < 1e-5
    assert rel_err(trace[f"vec_upd{L-1}"].numpy(), fx[f"vec_upd{L-1}"]) < 1e-5
    energy, forces, loss, grads = R.train_step(params, cfg, pos, z, batch, torch.tensor(fx["y"]),
                                               torch.tensor(fx["f_target"]), ei)
    assert rel_err(energy.numpy(), fx["energy"]) < 1e-6
    assert rel_err(forces.numpy(), fx["forces"]) < 1e-5
    assert rel_err(e_eval.numpy(), fx["energy"]) < 1e-6
    assert rel_err(f_eval.numpy(), fx["forces"]) < 1e-5
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    check_grads(fx, {k: v.numpy() for k, v in grads.items()}, tol, name)
