"""trainer.GraphedStep on the autograd-driven models at the reference's batch sizes: with ``data.prepared = net.prepare(data)`` the forward issues no
host synchronisation, so the whole training step can be captured into a HIP graph; replays must give what the eager steps give (same start, same batch).
QHNet (round 5): its step captures like the others since the tensors its forward leaves on the batch are detached -- a tracked ``data.node_attr`` kept the
previous step's AccumulateGrad nodes (default stream) alive, the captured backward then pulled the legacy stream into the capture and hipStreamEndCapture
crashed (scripts/debug_qhnet_capture.py: ``only:node_embedding`` was the one pruned backward that died)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
pytestmark = pytest.mark.gpu


def _reset(flat, opt, snap):
    with torch.no_grad():
        flat.flat.copy_(snap)
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()


@pytest.mark.parametrize("which", ["qhnet", "gemnet", "escn", "equiformer"])
def test_graph_replay_equals_eager_steps(which):
    import bench_graphed as BG
    from nabladft_amd.trainer import GraphedStep
    dev = torch.device("cuda:0")
    step, flat, opt, net = BG.make_step(which, BG.REFERENCE_BATCH[which], dev)
    step()                                                  # lazy initialisations (constant tables, scale factors) outside everything that is compared
    torch.cuda.synchronize()
    snap = flat.flat.detach().clone()
    _reset(flat, opt, snap)
    eager = [float(step().detach()) for _ in range(3)]
    p_eager = flat.flat.detach().clone()
    _reset(flat, opt, snap)
    g = GraphedStep(step, warmup=2)
    _reset(flat, opt, snap)
    graphed = [float(g().detach()) for _ in range(3)]
    torch.cuda.synchronize()
    p_graph = flat.flat.detach().clone()
    assert eager[0] != eager[2]                             # the optimiser really moved the model
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 1e-6 * max(abs(a), 1e-6), (eager, graphed)
    assert float((p_eager - p_graph).abs().max()) <= 1e-6 * max(float(p_eager.abs().max()), 1.0)


def test_prepared_batch_of_another_size_is_refused():
    import bench_graphed as BG
    import bench_escn as BE
    dev = torch.device("cuda:0")
    net = BE.build(dev)
    a, b = BE.synthetic_batch(2, 100, dev), BE.synthetic_batch(3, 101, dev)
    b.prepared = net.prepare(a)
    with pytest.raises(ValueError):
        net(b)


def test_prepared_batch_is_tied_to_its_geometry():
    """positions updated in place (MD, geometry optimisation, refreshed graph inputs) or another batch of the SAME size must not reuse a stale
    neighbour list / frames / Wigner rows: prepare() records (data_ptr, in-place version, shape) of the positions and forward compares it."""
    import bench_escn as BE
    import bench_qhnet as BQ
    dev = torch.device("cuda:0")
    for mod in (BE, BQ):
        net = mod.build(dev)
        a = mod.synthetic_batch(2, 100, dev)
        a.prepared = net.prepare(a)
        with torch.no_grad():
            net(a)                                          # unchanged geometry: accepted, twice
            net(a)
            a.pos.add_(0.01)                                # in place: same tensor, new version
            with pytest.raises(ValueError):
                net(a)
            a.prepared = net.prepare(a)                     # prepared again: accepted
            net(a)
            b = mod.synthetic_batch(2, 100, dev)            # same sizes, other tensor
            b.prepared = a.prepared
            with pytest.raises(ValueError):
                net(b)
        assert not any(torch.is_tensor(v) and v.requires_grad for v in vars(a.prepared).values())      # nothing autograd-tracked is cached on it


def test_qhnet_prepared_forward_is_free_of_host_synchronisation():
    """Capturing a region fails on any host read inside it: the forward on a prepared batch captures, replays, and equals the eager forward."""
    import bench_qhnet as BQ
    dev = torch.device("cuda:0")
    net = BQ.build(dev)
    b = BQ.synthetic_batch(2, 100, dev)
    with torch.no_grad():
        ref = net(b, packed=True).clone()                   # eager, graphs built inside the forward
        b.prepared = net.prepare(b)
        net(b, packed=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net(b, packed=True)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_split_k_scratch_survives_capture_growth_and_replay():
    """ADVICE r4 (gemm.hip split-K scratch): a HIP graph captures the raw pointer of the library-owned split-K buffer of its stream.  A later EAGER product on
    the same stream that needs a larger buffer must not free it: the old buffer is retired, the graph still replays into live memory and reproduces its result
    bit for bit; inside a capture nothing is allocated."""
    import ctypes as C
    from nabladft_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    M, N, K = 256, 128, 4096                 # 2 output tiles, contraction 4096: split into 8 ranges (csrc/gemm.hip splitk_count)
    A, W = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    Cg, Ce = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    M2 = 2048                                # larger product for the growth: 16 tiles x 8 ranges of slabs
    A2, C2 = torch.randn(M2, K, generator=g).to(dev), torch.empty(M2, N, device=dev)

    def state(stream):
        p, f, c, r = C.c_uint64(0), C.c_uint64(0), C.c_int32(0), C.c_int32(0)
        _lib.check(lib.nq_gemm_splitk_state(stream, C.byref(p), C.byref(f), C.byref(c), C.byref(r)))
        return p.value, f.value, c.value, r.value

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        st = _lib.stream_ptr()
        _lib.check(lib.nq_linear_forward(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(Ce), None, M, N, K, st))     # eager: allocates the stream's buffer
        side.synchronize()
        p0, f0, c0, r0 = state(st)
        assert p0 != 0 and f0 >= 8 * M * N and c0 == 0 and r0 == 0
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            st_cap = _lib.stream_ptr()
            assert st_cap.value == st.value
            _lib.check(lib.nq_linear_forward(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(Cg), None, M, N, K, st_cap))
        p1, f1, c1, r1 = state(st)
        assert (p1, f1) == (p0, f0) and c1 == 1, "the capture must reuse the existing buffer and mark it"
        graph.replay(); side.synchronize()
        assert torch.equal(Cg, Ce)
        _lib.check(lib.nq_linear_forward(_lib.ptr(A2), _lib.ptr(W), None, _lib.ptr(C2), None, M2, N, K, st))    # eager, needs 8x the slabs: growth
        side.synchronize()
        p2, f2, c2, r2 = state(st)
        assert p2 != p0 and f2 >= 8 * M2 * N and r2 == 1 and c2 == 0, "the captured buffer is retired, not freed"
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(8)]   # would land in the freed block if it had been freed
        Cg.zero_()
        graph.replay(); side.synchronize()
        assert torch.equal(Cg, Ce), "replay after the growth must still reproduce the captured product"
        ref = (A2.double() @ W.double().T).float()
        assert float((C2 - ref).abs().max()) < 1e-3 * float(ref.abs().max())
        del junk, graph
    lib.nq_gemm_splitk_release()
    assert state(st)[0] == 0
