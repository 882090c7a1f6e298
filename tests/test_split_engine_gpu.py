"""The golden vectors of the reference with EVERY eligible dense product on the split-bf16 engine (csrc/gemm_split.h; nq_set_gemm_variant bit 6 lifts the
size threshold that normally leaves small launches on the exact-f32 engine -- the fixtures' molecules are small, so without it most of their products would
not touch the new engine at all).  Same assertions, same tolerances as the model tests: the functions below ARE those tests, re-run under the switch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_everywhere():
    from nabladft_amd import _lib
    lib = _lib.load()
    lib.nq_set_gemm_variant(1 | 64)
    yield lib
    lib.nq_set_gemm_variant(1)


def test_switch_really_moves_small_products_to_the_split_engine(split_everywhere):
    from nabladft_amd import _lib
    lib, dev = split_everywhere, torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    A, W = torch.randn(300, 128, generator=g).to(dev), torch.randn(64, 128, generator=g).to(dev)
    st = _lib.stream_ptr()

    def run():
        out = torch.empty(300, 64, device=dev)
        _lib.check(lib.nq_linear_forward(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), None, 300, 64, 128, st))
        return out
    forced = run()
    lib.nq_set_gemm_variant(1)
    default = run()
    lib.nq_set_gemm_variant(1 | 64)
    ref = A.double() @ W.double().T
    assert not torch.equal(forced, default)                          # two different code paths ...
    e_f, e_d = float((forced.double() - ref).abs().max()), float((default.double() - ref).abs().max())
    assert e_f <= 1.5 * e_d + 1e-7 and e_d < 1e-4                      # ... of the same accuracy


@pytest.mark.parametrize("name", ["painn_small_ragged.npz", "painn_full_real4.npz", "painn_small_expenv.npz"])
def test_painn_golden(name, split_everywhere, monkeypatch):
    from tests import test_engine_gpu as T
    T.test_engine_matches_reference_golden(name, "fused", monkeypatch)


def test_painn_fused_step_golden(split_everywhere):
    from tests import test_engine_gpu as T
    T.test_fused_step_matches_golden_and_is_deterministic()


def test_gemnet_oc_golden(split_everywhere):
    import os
    from tests import test_gemnet_gpu as T
    small, full = np.load(os.path.join(T.GOLD, "gemnet_small.npz")), np.load(os.path.join(T.GOLD, "gemnet_full.npz"))
    T.test_forward_small_matches_reference_layer_by_layer(small)
    T.test_gradients_small_match_reference(small)
    T.test_full_config_forward_and_gradients(full)


def test_escn_golden(split_everywhere):
    from tests import test_escn_gpu as T
    T.test_gradients_small()
    T.test_full_configuration()


def test_equiformer_v2_golden(split_everywhere):
    from tests import test_equiformer_gpu as T
    T.test_gradients_small()
    T.test_full_configuration()


def test_qhnet_golden(split_everywhere):
    from tests import test_qhnet_gpu as T
    T.test_small_loss_and_all_gradients()
    T.test_full_configuration()


def test_phisnet_golden(split_everywhere):
    from tests import test_phisnet_gpu as T
    T.test_neural_network_matches_reference()
