"""Fused training step on the C ABI (no autograd graph, no per-parameter Python loops):
neighbour list -> forward + forces -> L1/L2 loss kernel -> tangent + dual reverse -> one RCCL
all-reduce of the flat gradient -> clip + AdamW kernel.  Semantically this is one
``PaiNNLightning.training_step`` + Lightning's backward / clip (config/painn-oc.yaml:18-19) /
``torch.optim.AdamW.step`` (config/model/painn-oc.yaml:23-27)."""
import ctypes as C
import os
import weakref
from types import SimpleNamespace

import torch

from . import _lib, dist as nqdist
from . import spk
from .painn import PaiNN, build_neighbor_list


class Batch:
    """Minimal PyG-batch-shaped container (pos, z, batch, ptr, y, forces, num_nodes)."""

    def __init__(self, pos, z, batch, y=None, forces=None, ptr=None):
        self.pos, self.z, self.batch, self.y, self.forces = pos, z, batch, y, forces
        if ptr is None:
            counts = torch.bincount(batch)
            ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        self.ptr = ptr
        self.num_nodes = pos.shape[0]

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return Batch(mv(self.pos), mv(self.z), mv(self.batch), mv(self.y), mv(self.forces), mv(self.ptr))


class _PygEngine:
    """nabladft_amd.PaiNN: the module's parameters are views of the engine's flat buffer -- nothing to convert."""

    def __init__(self, model: PaiNN):
        if getattr(model, "direct_forces", False) and getattr(model, "regress_forces", False):
            raise NotImplementedError("FusedTrainStep covers the autograd-force model; train direct_forces=True models through the autograd boundary")
        self.model, self.cfg, self.cutoff, self.max_neighbors = model, model._cfg, model.cutoff, model.max_neighbors
        self.offsets = model.radial_basis.engine_buffer()

    def flat(self):
        return self.model.flat_parameters()

    def entry_points(self, lib):
        return lib.nq_painn_workspace_bytes, lib.nq_painn_forward, lib.nq_painn_backward

    def writeback(self):
        pass


class _SpkEngine:
    """nabladft_amd.spk.NeuralNetworkPotential: the step trains an engine-layout copy of the parameters (a permutation of
    the spk tensors, so clip / AdamW act identically) and ``writeback`` scatters it into the spk-shaped nn.Parameters."""

    def __init__(self, pot: "spk.NeuralNetworkPotential"):
        rep = pot.representation
        self.model, self.cfg, self.cutoff, self.max_neighbors = pot, pot._cfg, rep.cutoff, 2 ** 30
        self.offsets = rep.radial_basis.offsets
        dev = self.offsets.device
        pot._prepare(dev)
        with torch.no_grad():
            self._flat = torch.cat([p.detach().to(torch.float32).reshape(-1) for p in pot._engine_params()])
            if pot._index is not None:
                self._flat = self._flat.index_select(0, pot._index)

    def flat(self):
        return self._flat

    def entry_points(self, lib):
        return self.model._entry_points(lib)

    @torch.no_grad()
    def writeback(self):
        pot = self.model
        ps = pot._engine_params()
        spk_flat = self._flat if pot._index is None else torch.cat([p.detach().reshape(-1) for p in ps]).index_copy_(0, pot._index, self._flat)
        o = 0
        for p in ps:
            p.copy_(spk_flat[o:o + p.numel()].view_as(p))
            o += p.numel()


class FusedTrainStep:
    """``loss``: "l1_l2" (painn.py:741-745, the in-tree PaiNN) or "mse" (config/model/painn.yaml:30-46, the schnetpack task);
    default by model type.  For an spk-shaped model call ``writeback()`` before reading its nn.Parameters / state_dict."""

    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=5.0, coef_energy=1.0,
                 coef_forces=1.0, group=None, loss=None):
        self.model, self.group = model, group
        self._eng = _SpkEngine(model) if isinstance(model, spk.NeuralNetworkPotential) else _PygEngine(model)
        self.loss_kind = loss or ("mse" if isinstance(self._eng, _SpkEngine) else "l1_l2")
        if self.loss_kind not in ("l1_l2", "mse"):
            raise ValueError(f"unknown loss {self.loss_kind!r}")
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.ce, self.cf = coef_energy, coef_forces
        flat = self._eng.flat()
        nqdist.broadcast_(flat, 0, group)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.grad = torch.empty_like(flat)
        self.scratch = torch.empty(512, device=flat.device, dtype=torch.float32)
        self.loss = torch.zeros(1, device=flat.device, dtype=torch.float32)
        self.t = 0
        self.energy = self.forces = None
        self._ov = None   # lazily built state of the overlapped gradient all-reduce
        self._ws = None   # persistent, grow-only workspace (forward+backward finish inside one call, so reuse is safe)

    def __call__(self, batch, update=True):
        lib = _lib.load()
        model, eng = self.model, self._eng
        flat = eng.flat()
        cfg = C.byref(eng.cfg)
        st = _lib.stream_ptr()
        nl = build_neighbor_list(batch.pos, batch.batch, batch.z, eng.cutoff, eng.max_neighbors, batch.ptr)
        if nl.E == 0:
            raise IndexError("batch has no edges within the cutoff")
        dev = flat.device
        ws_fn, fwd_fn, bwd_fn = eng.entry_points(lib)
        ws_bytes = ws_fn(cfg, nl.N, nl.E, nl.B)
        if self._ws is None or self._ws.numel() < ws_bytes:
            self._ws = None                                            # release before growing (27 GB at B=1024)
            self._ws = torch.empty((int(ws_bytes * 1.08) + 4096) // 256 * 256, device=dev, dtype=torch.uint8)
        ws = self._ws
        energy = torch.empty(nl.B, device=dev, dtype=torch.float32)
        forces = torch.empty(nl.N, 3, device=dev, dtype=torch.float32)
        gE, gF = torch.empty_like(energy), torch.empty_like(forces)
        if batch.y is None or batch.forces is None:
            raise ValueError("FusedTrainStep needs both targets: batch.y [B] and batch.forces [N, 3]")
        ty = batch.y.to(device=dev, dtype=torch.float32).contiguous().view(-1)          # raw pointers go to the loss kernel: coerce dtype / layout here
        tf = batch.forces.to(device=dev, dtype=torch.float32).contiguous()
        if ty.numel() != nl.B or tuple(tf.shape) != (nl.N, 3):
            raise ValueError(f"targets have shapes {tuple(batch.y.shape)} / {tuple(batch.forces.shape)}, expected [{nl.B}] / [{nl.N}, 3]")
        _lib.check(fwd_fn(cfg, _lib.ptr(flat), _lib.ptr(eng.offsets), C.byref(nl.c), _lib.ptr(ws), ws_bytes, _lib.ptr(energy), _lib.ptr(forces), st))
        loss_fn = lib.nq_loss_mse if self.loss_kind == "mse" else lib.nq_loss_l1_l2
        _lib.check(loss_fn(_lib.ptr(energy), _lib.ptr(ty), nl.B, _lib.ptr(forces), _lib.ptr(tf), nl.N, self.ce, self.cf,
                           _lib.ptr(self.loss), _lib.ptr(gE), _lib.ptr(gF), st))
        if self._overlap_ready(lib, bwd_fn):
            self._backward_overlapped(lib, cfg, flat, eng, nl, ws, ws_bytes, gE, gF, st)
        else:
            _lib.check(bwd_fn(cfg, _lib.ptr(flat), _lib.ptr(eng.offsets), C.byref(nl.c), _lib.ptr(ws), ws_bytes, _lib.ptr(gE), _lib.ptr(gF),
                              _lib.ptr(self.grad), st))
            self._mark_exposed(0)
            nqdist.allreduce_mean_(self.grad, self.group)
            self._mark_exposed(1)
        if update:
            self.t += 1
            _lib.check(lib.nq_adamw_step(_lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.m), _lib.ptr(self.v), flat.numel(),
                                         float(self.max_norm or 0.0), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                         _lib.ptr(self.scratch), st))
        self.energy, self.forces = energy, forces
        model._last_ws, model._last_nl = ws, nl
        return self.loss

    def writeback(self):
        """Make the module's nn.Parameters reflect the trained values (no-op for nabladft_amd.PaiNN)."""
        self._eng.writeback()

    # ---- gradient all-reduce overlapped with the reverse sweep (data parallel, world > 1) -------------------------------------------------------
    overlap = True      # set False to fall back to one all-reduce of the whole flat gradient after the backward

    def _overlap_ready(self, lib, bwd_fn):
        if not self.overlap or not nqdist.active(self.group) or bwd_fn is not lib.nq_painn_backward:
            return False
        if self._ov is None:
            L = int(self._eng.cfg.num_layers)
            ranges = (C.c_int64 * (4 * (L + 1)))()
            _lib.check(lib.nq_painn_layer_param_ranges(C.byref(self._eng.cfg), ranges))
            events = [torch.cuda.Event() for _ in range(L)]
            for e in events:
                e.record()                                     # creates the underlying hipEvent_t
            self._ov = SimpleNamespace(L=L, ranges=list(ranges), events=events, handles=(C.c_void_p * L)(*[e.cuda_event for e in events]),
                                       side=torch.cuda.Stream(), done=torch.cuda.Event())
        return True

    def _backward_overlapped(self, lib, cfg, flat, eng, nl, ws, ws_bytes, gE, gF, st):
        """The reverse sweep differentiates layer L-1 first; as soon as a layer's gradient slices are final (an event recorded by the engine) a side
        stream all-reduces them while the main stream goes on with the earlier layers.  The flat buffer is ordered [embedding | message layers |
        update layers | head], so a layer is two slices; the head goes with the last layer, the embedding after the sweep."""
        ov, g, w = self._ov, self.grad, nqdist.world_size(self.group)
        main = torch.cuda.current_stream()
        _lib.check(lib.nq_painn_backward_events(cfg, _lib.ptr(flat), _lib.ptr(eng.offsets), C.byref(nl.c), _lib.ptr(ws), ws_bytes, _lib.ptr(gE), _lib.ptr(gF),
                                                _lib.ptr(g), ov.handles, st))
        ov.done.record(main)
        r = ov.ranges

        def reduce(off, cnt):
            if cnt > 0:
                nqdist.allreduce_sum_(g[off:off + cnt], self.group)

        with torch.cuda.stream(ov.side):
            for i in range(ov.L):
                l = ov.L - 1 - i
                ov.side.wait_event(ov.events[i])
                if i == 0:
                    reduce(r[4 * ov.L], r[4 * ov.L + 1])
                reduce(r[4 * l], r[4 * l + 1])
                reduce(r[4 * l + 2], r[4 * l + 3])
            ov.side.wait_event(ov.done)
            reduce(r[4 * ov.L + 2], r[4 * ov.L + 3])
        self._mark_exposed(0)
        main.wait_stream(ov.side)
        self._mark_exposed(1)
        g.mul_(1.0 / w)

    # ---- exposed all-reduce time: what the step's stream spends waiting for (or running) the gradient collective ---------------------------------------
    _exposed = None

    def _mark_exposed(self, which):
        if not nqdist.active(self.group) or not self.grad.is_cuda:
            return
        if self._exposed is None:
            self._exposed = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._exposed[which].record(torch.cuda.current_stream())

    def allreduce_exposed_ms(self):
        """GPU time of the LAST step between the end of the reverse sweep on the step's stream and the point where the reduced gradient is available to it:
        the wait for the side-stream all-reduces (overlapped path) or the whole collective (single all-reduce).  None without a process group."""
        if self._exposed is None:
            return None
        torch.cuda.synchronize()
        return float(self._exposed[0].elapsed_time(self._exposed[1]))



class _FlatParameter(torch.nn.Parameter):
    """The flat parameter of FlatParameters: its ``.grad`` IS the persistent flat gradient buffer.  ``opt.zero_grad()`` (set_to_none=True by
    default: ``p.grad = None``) therefore zeroes the buffer instead of detaching it -- the per-parameter ``p.grad`` views that autograd
    accumulates into keep pointing at the memory the optimiser reads."""

    def __new__(cls, data, grad_buffer):
        obj = super().__new__(cls, data, requires_grad=True)
        obj._grad_buffer = grad_buffer
        obj._owner = None                                   # weakref to the FlatParameters that built it
        return obj

    @property
    def grad(self):
        owner = self._owner() if self._owner is not None else None
        if owner is not None and owner._detached:
            owner.gather()                                  # first reader after a backward pass (clipping, the optimiser, an all-reduce): see FlatParameters
        return self._grad_buffer

    @grad.setter
    def grad(self, value):
        if value is None:
            owner = self._owner() if self._owner is not None else None
            if owner is not None:
                owner.zero_grad()                           # also hands the per-parameter gradients back to autograd (FlatParameters.zero_grad)
            else:
                self._grad_buffer.zero_()
        elif value is not self._grad_buffer:
            self._grad_buffer.copy_(value)


class FlatParameters:
    """All trainable parameters of a module as views of ONE flat buffer, their gradients as views of one flat gradient buffer.  A model with
    thousands of small tensors (PhiSNet: 2.4 k) otherwise spends tens of milliseconds per step in per-tensor optimiser bookkeeping; with the
    flat pair, zero_grad is one memset, clipping one norm, and any torch optimiser (or ``nq_adamw_step``) runs on a single tensor.  The module's
    parameter objects, names and state_dict are unchanged (``p.data`` / ``p.grad`` are re-pointed).

    Gradients (``gather=True``, the default): with ``p.grad`` pointing at its slice, autograd ADDS every parameter's gradient into the slice -- one
    elementwise launch per tensor and step (QHNet 162, GemNet-OC 319, eSCN 339, EquiformerV2 530, PhiSNet 2.4 k: round 5 found them as the largest
    ATen share of the profiles).  So ``zero_grad()`` zeroes the buffer AND sets ``p.grad = None``: autograd then takes ownership of the first gradient
    tensor of each parameter without any kernel, and the first reader of the flat gradient afterwards (``flat.flat.grad``: clipping, the optimiser,
    an all-reduce) -- normally an engine callback at the end of the backward pass, queued by one of four sentinel parameters -- copies all of them into
    their slices with multi-tensor copies and points ``p.grad`` back at the slices (``gather()``).  A second backward without
    ``zero_grad()`` in between (gradient accumulation) finds the views in place and adds into them as before."""

    def __init__(self, params, gather: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self._gather_mode = bool(gather) and os.environ.get("NQ_FLAT_GATHER", "1") != "0"   # NQ_FLAT_GATHER=0: the in-place views of rounds 2-4 (A/B runs)
        self._detached = False                              # True between zero_grad() and the gather: p.grad is None or autograd's own tensor
        # every tensor of 16 or more elements starts on a 16-byte boundary of the flat buffer: the tile engines of the dense products read their operands
        # with 16-byte loads and send a weight that starts off such a boundary to the generic kernels (measured on QHNet, whose 50-element radial
        # parameters shifted the [8320 x 128] weight generators behind them: 62-82 instead of 130-180 TFLOP/s).  The padding floats stay zero (zero
        # gradient, zero AdamW update); runs of small tensors (so3.SelfMixing's per-path vectors) stay contiguous for ``block_of``.
        offs, o = [], 0
        for p in self.params:
            if p.numel() >= 16:
                o = (o + 3) & ~3
            offs.append(o)
            o += p.numel()
        n = o
        dev = self.params[0].device
        self.flat = _FlatParameter(torch.zeros(n, device=dev, dtype=torch.float32), torch.zeros(n, device=dev, dtype=torch.float32))
        self.flat._owner = weakref.ref(self)
        self.offset = {}
        self._views = []                                                # the gradient slices, in self.params order
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                k = p.numel()
                self.flat.data[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat.data[o:o + k].view(p.shape)
                p.grad = self.flat.grad[o:o + k].view(p.shape)          # until the first zero_grad(): autograd accumulates into the view in place
                self._views.append(p.grad)
                self.offset[id(p)] = o
        self._index = {id(p): i for i, p in enumerate(self.params)}
        # A handful of sentinel parameters queue the gather as an engine callback at the end of the backward pass they take part in, so that the module's
        # ``p.grad`` are slices again as soon as ``backward()`` returns (hooks on ALL parameters were measured: 2.4 k Python hook calls per pass cost PhiSNet
        # 8 ms per step).  If no sentinel receives a gradient the gather still happens at the first read of ``flat.grad``.
        self._queued = False
        self._handles = []
        if self._gather_mode:
            ref = weakref.ref(self)

            def sentinel(_p, ref=ref):
                me = ref()
                if me is not None and me._detached and not me._queued:
                    me._queued = True
                    torch.autograd.Variable._execution_engine.queue_callback(me._gather_callback)
            n = len(self.params)
            for i in sorted({0, n // 3, (2 * n) // 3, n - 1}):
                self._handles.append(self.params[i].register_post_accumulate_grad_hook(sentinel))

    def _gather_callback(self):
        self._queued = False
        self.gather()

    def attach(self, model):
        """Optional: modules that keep MANY small parameter tensors side by side (``so3.SelfMixing``: one [F] vector per Clebsch-Gordan path)
        read them as one block of the flat buffer and add their gradient block with one kernel instead of one per tensor (``block_of``).
        Returns the number of modules switched over."""
        n = 0
        for m in model.modules():
            hook = getattr(m, "_use_flat_parameters", None)
            if hook is not None and hook(self):
                n += 1
        return n

    def block_of(self, params):
        """(offset, numel) of a run of parameters if they are trainable and contiguous in the flat buffer in this order, else None."""
        if not params or any(id(p) not in self.offset for p in params):
            return None
        o0 = o = self.offset[id(params[0])]
        for p in params:
            if self.offset[id(p)] != o:
                return None
            o += p.numel()
        return o0, o - o0

    def zero_grad(self):
        self.flat._grad_buffer.zero_()
        if self._gather_mode:
            for p in self.params:
                p.grad = None                               # autograd steals the first gradient of the step instead of adding it into the slice
            self._detached, self._queued = True, False

    def gather(self, params=None):
        """Copies the gradients autograd holds for ``params`` (default: all) into their slices of the flat gradient and points ``p.grad`` back at the
        slices; parameters without a gradient keep their zeroed slice.  Idempotent; runs by itself when ``flat.flat.grad`` is read.  Call it by hand
        only for part of the parameters DURING the backward pass (OverlappedAllReduce, per bucket, then reading the raw buffer)."""
        if not self._detached:
            return
        pairs = zip(self.params, self._views) if params is None else ((p, self._views[self._index[id(p)]]) for p in params)
        src, dst, odd = [], [], []
        for p, view in pairs:
            g = p.grad
            if g is view:
                continue
            if g is not None:
                # the multi-tensor copy takes dense fp32 tensors of the slice's device; anything else autograd may hand over (a sparse gradient of
                # nn.Embedding(sparse=True), another dtype under autocast) is added into the zeroed slice the way rounds 2-4 accumulated in place
                if g.layout is torch.strided and g.dtype is view.dtype and g.device == view.device:
                    src.append(g)
                    dst.append(view)
                else:
                    odd.append((view, g))
            p.grad = view
        with torch.no_grad():
            if src:
                torch._foreach_copy_(dst, src)
            for view, g in odd:
                view.add_((g.to_dense() if g.layout is not torch.strided else g).to(device=view.device, dtype=view.dtype))
        if params is None:
            self._detached = False

    def validate(self):
        """Raises if a parameter's ``.grad`` / ``.data`` no longer aliases the flat buffers (e.g. after ``module.zero_grad(set_to_none=True)`` on
        the MODULE, ``p.grad = None`` by hand, or ``module.to(...)``): from then on the optimiser would silently see stale gradients."""
        self.gather()                                       # between zero_grad() and the end of the backward pass the gradients are autograd's
        g0, d0 = self.flat.grad.data_ptr(), self.flat.data.data_ptr()
        for p in self.params:
            o = self.offset[id(p)] * 4
            if p.grad is None or p.grad.data_ptr() != g0 + o or p.data.data_ptr() != d0 + o:
                raise RuntimeError("FlatParameters: a parameter was detached from the flat buffers (use FlatParameters.zero_grad() or "
                                   "optimizer.zero_grad() on the flat parameter, not module.zero_grad(set_to_none=True))")

    def clip_grad_norm_(self, max_norm: float):
        self.validate()
        norm = self.flat.grad.norm()
        self.flat.grad.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm


class OverlappedAllReduce:
    """Data-parallel gradient averaging for the autograd-driven models, overlapped with the backward pass: the flat gradient buffer is cut into buckets of
    consecutive parameters (~``bucket_bytes`` each); a ``post_accumulate_grad`` hook counts a bucket's parameters down and, when the last one has its
    gradient, starts the bucket's all-reduce on a side stream while autograd keeps running the rest of the backward on the main stream
    (FusedTrainStep._backward_overlapped does the same for the fused PaiNN engine, per layer).  ``finish()`` -- call it after ``loss.backward()`` --
    launches the buckets whose parameters received no gradient this step, waits for all of them, and divides by the world size.  The result equals
    ``dist.allreduce_mean_(flat.grad)`` (same per-element sums; reference semantics: Lightning DDPStrategy, nablaDFT/utils/pipelines.py:65-68).
    Payloads here are 88-332 MB per step (21.9-83.1 M parameters): at xGMI ring rates 1-2 ms that would otherwise sit after the backward."""

    def __init__(self, flat: "FlatParameters", bucket_bytes: int = 32 << 20, group=None):
        import torch.distributed as dist
        self.flat, self.group, self.dist = flat, group, dist
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.on = nqdist.active(group)                      # world > 1, or a forced 1-rank group (NQ_DIST_FORCE=1: the single-GPU RCCL test)
        self.buckets = []                                   # [lo, hi, n_params]: ranges of the FLAT buffer (alignment padding included)
        self._bucket_of = {}
        self._params_of = []                                # per bucket: its parameters
        # FlatParameters pads tensors of >= 16 elements to 16-byte boundaries: a bucket's range comes from the real offsets, never from a running
        # sum of numel().  A bucket = a run of consecutive parameters; it starts where the previous one ended (so the padding in front of its first
        # parameter travels with it) and the last bucket ends at the end of the buffer.
        total = flat.flat.numel()
        lo, count = 0, 0
        for i, p in enumerate(flat.params):
            assert flat.offset[id(p)] >= lo, "FlatParameters.params is not in buffer order"
            end = flat.offset[id(p)] + p.numel()
            count += 1
            self._bucket_of[id(p)] = len(self.buckets)
            if len(self._params_of) == len(self.buckets):
                self._params_of.append([])
            self._params_of[-1].append(p)
            if (end - lo) * 4 >= bucket_bytes and i + 1 < len(flat.params):
                self.buckets.append([lo, end, count])
                lo, count = end, 0
        if count or not self.buckets:
            self.buckets.append([lo, total, count])
        while len(self._params_of) < len(self.buckets):
            self._params_of.append([])
        self.check_tiling()
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._cuda = flat.flat.is_cuda
        self._comm = torch.cuda.Stream(device=flat.flat.device) if self._cuda else None
        self._handles = []
        if self.on:
            for p in flat.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def check_tiling(self):
        """The buckets tile [0, flat.numel()) exactly and every parameter's slice lies inside its own bucket."""
        assert self.buckets[0][0] == 0 and self.buckets[-1][1] == self.flat.flat.numel(), (self.buckets[0], self.buckets[-1], self.flat.flat.numel())
        for a, b in zip(self.buckets, self.buckets[1:]):
            assert a[1] == b[0] and a[0] < a[1], (a, b)
        for p in self.flat.params:
            lo, hi, _ = self.buckets[self._bucket_of[id(p)]]
            o = self.flat.offset[id(p)]
            assert lo <= o and o + p.numel() <= hi, (o, p.numel(), lo, hi)
        assert sum(b[2] for b in self.buckets) == len(self.flat.params)

    def _hook(self, p):
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        lo, hi, _ = self.buckets[b]
        self.flat.gather(self._params_of[b])                # the bucket's gradients into their slices (FlatParameters hands them to autograd between zero_grad and here)
        view = self.flat.flat._grad_buffer[lo:hi]           # the raw buffer: reading ``.grad`` would gather every parameter, also those still to come
        if self._cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())       # the accumulations into this slice are on the stream autograd runs this node on
            self._comm.wait_event(ready)
            with torch.cuda.stream(self._comm):
                if self.group is None and nqdist.native_comm(create=False) is not None:
                    nqdist.allreduce_sum_(view)   # NQ_RCCL_NATIVE=1: the SAME communicator as FusedTrainStep / allreduce_mean_ (two communicators issuing collectives from
                                                  # different streams in one step is the classic cross-rank ordering hazard); stream-ordered, no work handle
                else:
                    self._works.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """After backward: every bucket reduced, the flat gradient holds the mean over ranks; counters re-armed for the next step."""
        if self.on:
            for b in range(len(self.buckets)):              # parameters without a gradient this step never fire their hook
                self._launch(b)
            for w in self._works:
                w.wait()
            if self._cuda:
                torch.cuda.current_stream().wait_stream(self._comm)
            if self.world > 1:
                self.flat.flat.grad.mul_(1.0 / self.world)
        self._works = []
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        return self.flat.flat.grad

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class GraphedStep:
    """A whole training step -- zero the gradients, forward, loss, backward, clip, optimiser -- captured ONCE into a HIP graph and replayed: for models whose
    step is a thousand small launches issued from Python autograd (PhiSNet: ~1000, device busy 45 % of the wall time) the host disappears from the
    critical path.  Requirements: ``fn()`` must not synchronise with the host (prepared batch composition, device-side scalars), must read its inputs
    from tensors that keep their addresses (update them in place between replays) and must use capturable optimiser settings
    (``torch.optim.Adam(..., capturable=True)``).  ``fn`` returns a tensor (the loss) that is refreshed by every replay.  Drop every reference to
    results of earlier EAGER calls of ``fn`` (e.g. a kept loss tensor) before constructing this: their autograd graph keeps AccumulateGrad nodes of the
    default stream alive, and touching those from the capture stream aborts the capture."""

    def __init__(self, fn, warmup: int = 3):
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up outside the capture: lazy initialisations, allocator pool sizes
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
