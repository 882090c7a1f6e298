"""GPU: overlapped host->device feed (nabladft_amd/data.py) and a few real-data training steps on the database fixture."""
import os

import pytest
import torch

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
DB = os.path.join(GOLDEN, "energy_db_30.db")


def test_loader_on_gpu_matches_host_collate_and_feeds_training():
    import nabladft_amd as nq
    from nabladft_amd import data as D
    arena = D.read_energy_database(DB)
    ld = D.ArenaLoader(arena, 8, "cuda", shuffle=True, seed=5)
    plan = D.epoch_plan(arena.sizes, 8, True, 5, 0)
    n = 0
    for bt, sel in zip(ld, plan):
        ref = arena.batch(sel)
        assert bt.pos.is_cuda
        for k in ("pos", "z", "batch", "y", "forces", "ptr"):
            assert torch.equal(getattr(bt, k).cpu(), getattr(ref, k)), k
        n += 1
    assert n == len(plan) == 4 and arena.pos.is_pinned()

    # real energies / forces from the reference's data file through the fused step: the loss goes down
    torch.manual_seed(0)
    model = nq.PaiNN(64, 2, 20, 5.0, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100).cuda()
    step = nq.FusedTrainStep(model, lr=2e-3, max_grad_norm=5.0)
    losses = []
    for epoch in range(6):
        tot = 0.0
        for bt in ld:
            tot += float(step(bt))
        losses.append(tot)
    assert all(l == l for l in losses) and losses[-1] < 0.7 * losses[0], losses
