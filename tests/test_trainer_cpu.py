"""CPU: FlatParameters (host logic): same optimisation trajectory as per-tensor torch Adam."""
import torch

from nabladft_amd.trainer import FlatParameters


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3, bias=False), torch.nn.Linear(3, 1))


def test_flat_parameters_match_per_tensor_adam():
    a, b = _net(0), _net(0)
    b[2].weight.requires_grad = False                   # frozen tensors stay out of the flat buffer
    a[2].weight.requires_grad = False
    x, y = torch.randn(20, 6), torch.randn(20, 1)
    pa = [p for p in a.parameters() if p.requires_grad]
    opt_a = torch.optim.Adam(pa, lr=1e-2, amsgrad=True)
    fb = FlatParameters(b.parameters())
    opt_b = torch.optim.Adam([fb.flat], lr=1e-2, amsgrad=True)
    assert fb.flat.numel() == sum(p.numel() for p in pa)
    for _ in range(5):
        opt_a.zero_grad(set_to_none=True)
        ((a(x) - y) ** 2).mean().backward()
        torch.nn.utils.clip_grad_norm_(pa, 0.5)
        opt_a.step()
        fb.zero_grad()
        ((b(x) - y) ** 2).mean().backward()
        fb.clip_grad_norm_(0.5)
        opt_b.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, atol=1e-6), (p - q).abs().max()
    assert set(b.state_dict()) == set(a.state_dict())
    # parameters are views of the flat buffer
    assert b[0].weight.data_ptr() == fb.flat.data_ptr()


def test_flat_parameters_survive_optimizer_zero_grad():
    """``optimizer.zero_grad()`` defaults to set_to_none=True; on the flat parameter that must zero the gradient buffer, not detach it
    (otherwise autograd keeps accumulating into the old views while the optimiser sees grad=None and silently skips every update)."""
    import pytest
    net = _net(1)
    fp = FlatParameters(net.parameters())
    opt = torch.optim.AdamW([fp.flat], lr=0.1)
    x = torch.randn(7, 6)
    before = fp.flat.detach().clone()
    for _ in range(2):
        opt.zero_grad()                                # set_to_none=True
        assert float(fp.flat.grad.abs().max()) == 0.0
        net(x).sum().backward()
        assert float(fp.flat.grad.abs().max()) > 0.0
        fp.validate()
        opt.step()
    assert float((fp.flat.detach() - before).abs().max()) > 1e-3
    net.zero_grad(set_to_none=True)                    # detaches the per-parameter views: must be reported, not silently accepted
    with pytest.raises(RuntimeError):
        fp.clip_grad_norm_(1.0)


def test_ema_follows_torch_ema_semantics():
    """nabladft_amd.ema (torch_ema restated, qhnet.py:459-536): warm-up decay, store / copy_to / restore, and the argument checks torch_ema has
    (a parameter list of the wrong length must raise instead of being zip-truncated; to() moves the stored copies too)."""
    import pytest
    from nabladft_amd.ema import ExponentialMovingAverage
    torch.manual_seed(0)
    net = torch.nn.Linear(4, 3)
    ema = ExponentialMovingAverage(net.parameters(), decay=0.9)
    w0 = net.weight.detach().clone()
    with torch.no_grad():
        net.weight.add_(1.0)
    ema.update()
    d = min(0.9, 2.0 / 11.0)                                              # (1 + n) / (10 + n) at n = 1
    assert torch.allclose(ema.shadow_params[0], w0 + (1.0 - d) * 1.0, atol=1e-6)
    with pytest.raises(ValueError):
        ema.update([net.weight])                                          # one parameter instead of two
    with pytest.raises(ValueError):
        ema.copy_to(list(net.parameters()) + [torch.nn.Parameter(torch.zeros(1))])
    with pytest.raises(RuntimeError):
        ema.restore()
    live = net.weight.detach().clone()
    with ema.average_parameters():
        assert torch.allclose(net.weight, ema.shadow_params[0])
    assert torch.equal(net.weight, live)
    ema.store()
    ema.to(torch.device("cpu"))                                           # stored copies travel with the shadow
    assert all(c.device.type == "cpu" for c in ema._stored)
    with pytest.raises(ValueError):
        ema.to(dtype=torch.float16)
    sd = ema.state_dict()
    ema2 = ExponentialMovingAverage(net.parameters(), decay=0.5)
    ema2.load_state_dict(sd)
    assert ema2.decay == 0.9 and ema2.num_updates == 1 and torch.equal(ema2.shadow_params[1], ema.shadow_params[1])
    ema2.restore()


def test_flat_parameters_put_every_matrix_on_a_16_byte_boundary():
    """The tile engines of the dense products read 16 bytes at a time: a weight that starts off a 16-byte boundary falls back to the generic kernels.  Odd-sized
    tensors in front (QHNet's 50-element radial parameters) must therefore not shift the matrices behind them; values, gradients and the optimiser view stay
    intact, the padding stays zero, and runs of small vectors stay contiguous (block_of)."""
    import torch
    from nabladft_amd.trainer import FlatParameters
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in [(50,), (8, 16), (3,), (5,), (32, 7), (1,), (128,), (128,), (17, 4)]]
    want = [p.detach().clone() for p in ps]
    flat = FlatParameters(ps)
    for p, w in zip(ps, want):
        assert torch.equal(p.detach(), w)
        if p.numel() >= 16:
            assert p.data.storage_offset() % 4 == 0 and p.grad.storage_offset() == p.data.storage_offset()
    used = torch.zeros(flat.flat.numel(), dtype=torch.bool)
    for p in ps:
        o = flat.offset[id(p)]
        assert not used[o:o + p.numel()].any()
        used[o:o + p.numel()] = True
    assert float(flat.flat.data[~used].abs().sum()) == 0.0 and flat.flat.numel() - int(used.sum()) <= 3 * len(ps)
    assert flat.block_of([ps[6], ps[7]]) == (flat.offset[id(ps[6])], 256)            # two [128] vectors side by side
    sum((p * p).sum() for p in ps).backward()
    for p, w in zip(ps, want):
        assert torch.allclose(p.grad, 2 * w)
    flat.validate()
