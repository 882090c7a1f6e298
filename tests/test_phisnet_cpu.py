"""CPU: host logic of the PhiSNet network mirror (no compute): parameter surface, electron-configuration buffer, pair-of-pairs table, loud failure
without a GPU."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN


def test_electron_configuration_table_matches_reference_data():
    from nabladft_amd.phisnet import electron_configuration_table
    ref = np.load(os.path.join(GOLDEN, "phisnet_network.npz"))["electron_config"]          # the buffer of the real Embedding module
    assert np.abs(electron_configuration_table(87).numpy() - ref).max() < 1e-7


@pytest.mark.parametrize("n", [2, 3, 5])
def test_inferred_pair_of_pairs(n):
    from nabladft_amd.phisnet import inferred_pair_of_pairs
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    want = [(p, pairs.index((i, k))) for p, (i, j) in enumerate(pairs) for k in range(n) if k not in (i, j)]
    pi, pj = inferred_pair_of_pairs(n)
    assert list(zip(pi.tolist(), pj.tolist())) == want


def test_network_surface_and_cpu_failure():
    from tests.test_phisnet_gpu import _network_from_fixture
    fx = np.load(os.path.join(GOLDEN, "phisnet_network.npz"))
    with pytest.raises((RuntimeError, AssertionError), match="HIP|cuda|CUDA|GPU"):   # construction + name-for-name surface pass, .cuda() fails here
        _network_from_fixture(fx)
    from nabladft_amd.phisnet import NeuralNetwork
    with pytest.raises(ValueError):
        NeuralNetwork(max_orbitals=(((6, 0), (6, 2)),), order=1, num_features=32, num_basis_functions=8, num_modules=1, num_residual_pre_x=1,
                      num_residual_post_x=1, num_residual_pre_vi=1, num_residual_pre_vj=1, num_residual_post_v=1, num_residual_output=1, num_residual_pc=1,
                      num_residual_pn=1, num_residual_ii=1, num_residual_ij=1, num_residual_full_ii=1, num_residual_full_ij=1, num_residual_core_ii=1,
                      num_residual_core_ij=1, num_residual_over_ij=1, basis_functions="exp-bernstein", cutoff=8.0, activation="swish")


def test_load_from_reads_the_reference_checkpoint_and_save_round_trips(tmp_path):
    """NeuralNetwork(load_from=...) (neural_network.py:97-140, :445-449): hyper-parameters and weights from the file the REAL NeuralNetwork.save wrote
    (tests/golden/phisnet_checkpoint.pt, oracle/make_golden_phisnet.py --checkpoint: the network of the golden fixture); the mirror's own save() writes the same
    flat layout and loads back."""
    from nabladft_amd.phisnet import NeuralNetwork
    from tests.so3_helpers import FixtureCG
    fx = np.load(os.path.join(GOLDEN, "phisnet_network.npz"))
    m = NeuralNetwork(load_from=os.path.join(GOLDEN, "phisnet_checkpoint.pt"), clebsch_gordan=FixtureCG(), electron_config=fx["electron_config"])
    order, F, K, nm = (int(v) for v in fx["hp"])
    assert (m.order, m.num_features, m.num_basis_functions, m.num_modules) == (order, F, K, nm)
    assert m.activation == "swish" and float(m.cutoff) == float(fx["cutoff"]) and len(m.max_orbitals) == 6
    n_checked = 0
    for n, p in m.named_parameters():
        if n.startswith("energy_predictor."):
            continue                                  # the reference builds its EnergyLayer AFTER loading (neural_network.py:453): not restored there either
        assert np.array_equal(p.detach().numpy(), fx["p:" + n]), n
        n_checked += 1
    assert n_checked > 100 and m.get_number_of_parameters() == sum(p.numel() for p in m.parameters() if p.requires_grad)
    # the mirror's save() -> load_from round trip, same flat layout as the reference's save()
    path = str(tmp_path / "own.pt")
    m.save(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    ref = torch.load(os.path.join(GOLDEN, "phisnet_checkpoint.pt"), map_location="cpu", weights_only=False)
    assert set(ck) == set(ref)
    for k in ref:
        if k != "state_dict":
            assert ck[k] == ref[k] or (isinstance(ref[k], float) and abs(ck[k] - ref[k]) < 1e-12), k
    m2 = NeuralNetwork(load_from=path, clebsch_gordan=FixtureCG(), electron_config=fx["electron_config"])
    for (n, p), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n == n2
        if not n.startswith("energy_predictor."):
            assert torch.equal(p, p2), n
    # the training script's checkpoint layout: {'args': Namespace, 'model_state_dict': ...}
    from argparse import Namespace
    torch.save({"args": Namespace(**{k: v for k, v in ref.items() if k != "state_dict"}), "model_state_dict": ref["state_dict"]}, path)
    m3 = NeuralNetwork(load_from=path, clebsch_gordan=FixtureCG(), electron_config=fx["electron_config"])
    assert torch.equal(dict(m3.named_parameters())["embedding.embedding.element_embedding"] if "embedding.embedding.element_embedding" in dict(m3.named_parameters())
                       else next(m3.parameters()), dict(m.named_parameters()).get("embedding.embedding.element_embedding", next(m.parameters())))
