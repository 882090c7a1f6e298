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
