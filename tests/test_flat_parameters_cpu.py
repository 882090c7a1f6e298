"""trainer.FlatParameters, gradient side (round 5): between ``zero_grad()`` and the end of the backward pass the per-parameter gradients belong to autograd
(``p.grad = None``: the first gradient of every tensor is taken over without an elementwise launch); an engine callback then copies them into the flat
gradient with multi-tensor copies.  Everything the optimiser sees must equal what plain per-tensor autograd produces, bit for bit."""
import copy

import torch

from nabladft_amd.trainer import FlatParameters


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.a = torch.nn.Parameter(torch.randn(50, generator=g))                 # odd-sized first tensor: the next one is padded to a 16-byte boundary
        self.l1 = torch.nn.Linear(50, 32)
        self.l2 = torch.nn.Linear(32, 32)
        self.unused = torch.nn.Parameter(torch.randn(7, generator=g))             # never reaches the loss: no gradient this step
        self.shared = torch.nn.Parameter(torch.randn(32, generator=g))            # used twice: the second contribution is accumulated by autograd

    def forward(self, x):
        h = torch.tanh(self.l1(x * self.a) + self.shared)
        return (self.l2(h) * self.shared).sum()


def _reference_grads(net, xs):
    ref = copy.deepcopy(net)
    for x in xs:
        ref(x).backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in ref.named_parameters()}


def _flat_grads(net, flat):
    out = {}
    for k, p in net.named_parameters():
        o = flat.offset[id(p)]
        out[k] = flat.flat.grad[o:o + p.numel()].view(p.shape)
    return out


def test_gathered_gradients_equal_per_tensor_autograd():
    torch.manual_seed(0)
    net = _Net()
    x = torch.randn(4, 50)
    want = _reference_grads(net, [x])
    flat = FlatParameters(net.parameters())
    seen = []
    h = net.l2.weight.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.data_ptr()))
    flat.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    net(x).backward()
    h.remove()
    got = _flat_grads(net, flat)
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.data_ptr() == got[k].data_ptr(), k       # back on its slice after the pass
        if want[k] is None:
            assert float(got[k].abs().max()) == 0.0, k                                # no gradient: the zeroed slice
        else:
            assert torch.equal(got[k], want[k]), k
    assert seen and seen[0] != got["l2.weight"].data_ptr()                            # during the pass the gradient was autograd's own tensor, not the slice
    flat.validate()
    # the padding between tensors stays zero
    total = sum(float(g.abs().sum()) for g in got.values())
    assert abs(float(flat.flat.grad.abs().sum()) - total) <= 1e-4 * max(total, 1.0)


def test_accumulation_over_two_backward_passes_and_optimizer_zero_grad():
    torch.manual_seed(1)
    net = _Net()
    x1, x2 = torch.randn(4, 50), torch.randn(4, 50)
    want = _reference_grads(net, [x1, x2])
    flat = FlatParameters(net.parameters())
    opt = torch.optim.SGD([flat.flat], lr=0.1)
    opt.zero_grad()                                         # set_to_none=True on the flat parameter: routed to FlatParameters.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    net(x1).backward()
    net(x2).backward()                                      # no zero_grad in between: added into the slices in place
    got = _flat_grads(net, flat)
    for k in want:
        if want[k] is not None:
            assert torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-7), k
    before = flat.flat.data.clone()
    opt.step()
    assert torch.allclose(flat.flat.data, before - 0.1 * flat.flat.grad)
    for p in net.parameters():                              # the module's parameters ARE the flat buffer
        o = flat.offset[id(p)]
        assert p.data.data_ptr() == flat.flat.data[o:o + 1].data_ptr()


def test_gather_off_keeps_the_in_place_views():
    torch.manual_seed(2)
    net = _Net()
    x = torch.randn(4, 50)
    want = _reference_grads(net, [x])
    flat = FlatParameters(net.parameters(), gather=False)
    flat.zero_grad()
    assert all(p.grad is not None for p in net.parameters())
    net(x).backward()
    got = _flat_grads(net, flat)
    for k in want:
        if want[k] is not None:
            assert torch.equal(got[k], want[k]), k


def test_gradients_are_slices_again_when_backward_returns_and_the_lazy_fallback():
    """Normal case: one of the sentinel parameters takes part in the pass and the gather runs as an engine callback -- ``p.grad`` of the MODULE's parameters
    are slices of the flat gradient as soon as backward() returns.  If none does (here: only one middle parameter is used), the first read of the flat
    gradient gathers."""
    torch.manual_seed(4)
    net = _Net()
    flat = FlatParameters(net.parameters())
    flat.zero_grad()
    net(torch.randn(4, 50)).backward()
    g0 = flat.flat._grad_buffer.data_ptr()
    for k, p in net.named_parameters():                     # nothing has read flat.flat.grad yet
        assert p.grad is not None and g0 <= p.grad.data_ptr() < g0 + 4 * flat.flat.numel(), k
    ps = [torch.nn.Parameter(torch.randn(8)) for _ in range(9)]
    flat = FlatParameters(ps)
    flat.zero_grad()
    (ps[4] * 2.0).sum().backward()                          # sentinels are 0, 3, 6, 8
    assert ps[4].grad is not None and ps[0].grad is None and flat._detached
    o = flat.offset[id(ps[4])]
    assert torch.equal(flat.flat.grad[o:o + 8], torch.full((8,), 2.0)) and not flat._detached
    assert ps[0].grad is not None and float(flat.flat.grad.abs().sum()) == 16.0


def test_sparse_gradients_are_added_into_their_slices():
    """the multi-tensor copy only takes dense fp32 gradients; a sparse one (nn.Embedding(sparse=True)) goes through the
    add-into-the-zeroed-slice fallback and equals what per-tensor autograd holds."""
    torch.manual_seed(5)
    emb = torch.nn.Embedding(10, 6, sparse=True)
    lin = torch.nn.Linear(6, 3)
    idx = torch.tensor([1, 4, 4, 7])
    ref_e, ref_l = copy.deepcopy(emb), copy.deepcopy(lin)
    ref_l(ref_e(idx)).sum().backward()
    flat = FlatParameters(list(emb.parameters()) + list(lin.parameters()))
    flat.zero_grad()
    lin(emb(idx)).sum().backward()
    o = flat.offset[id(emb.weight)]
    assert torch.equal(flat.flat.grad[o:o + 60].view(10, 6), ref_e.weight.grad.to_dense())
    o = flat.offset[id(lin.weight)]
    assert torch.equal(flat.flat.grad[o:o + 18].view(3, 6), ref_l.weight.grad)
    assert emb.weight.grad.layout is torch.strided and emb.weight.grad.data_ptr() == flat.flat.grad[flat.offset[id(emb.weight)]:].data_ptr()
    flat.validate()


def test_overlapped_all_reduce_with_gradient_accumulation_world_1():
    """OverlappedAllReduce reads the RAW buffer per bucket after gathering that bucket: two backward passes without zero_grad() in between must leave the
    sum of both gradients (the second pass accumulates into the slices in place)."""
    from nabladft_amd.trainer import OverlappedAllReduce
    torch.manual_seed(6)
    net = _Net()
    xs = [torch.randn(4, 50), torch.randn(4, 50)]
    want = _reference_grads(net, xs)
    flat = FlatParameters(net.parameters())
    ar = OverlappedAllReduce(flat, bucket_bytes=1024)
    ar.check_tiling()
    flat.zero_grad()
    net(xs[0]).backward()
    ar.finish()
    net(xs[1]).backward()
    g = ar.finish()
    got = _flat_grads(net, flat)
    assert g.data_ptr() == flat.flat._grad_buffer.data_ptr() and not flat._detached
    for k in want:
        if want[k] is not None:
            assert torch.allclose(got[k], want[k], rtol=0, atol=1e-6), k
