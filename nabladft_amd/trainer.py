"""Fused training step on the C ABI (no autograd graph, no per-parameter Python loops):
neighbour list -> forward + forces -> L1/L2 loss kernel -> tangent + dual reverse -> one RCCL
all-reduce of the flat gradient -> clip + AdamW kernel.  Semantically this is one
``PaiNNLightning.training_step`` + Lightning's backward / clip (config/painn-oc.yaml:18-19) /
``torch.optim.AdamW.step`` (config/model/painn-oc.yaml:23-27)."""
import ctypes as C

import torch

from . import _lib, dist as nqdist
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


class FusedTrainStep:
    def __init__(self, model: PaiNN, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=5.0, coef_energy=1.0,
                 coef_forces=1.0, group=None):
        self.model, self.group = model, group
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.ce, self.cf = coef_energy, coef_forces
        flat = model.flat_parameters()
        nqdist.broadcast_(flat, 0, group)
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.grad = torch.empty_like(flat)
        self.scratch = torch.empty(512, device=flat.device, dtype=torch.float32)
        self.loss = torch.zeros(1, device=flat.device, dtype=torch.float32)
        self.t = 0
        self.energy = self.forces = None
        self._ws = None   # persistent, grow-only workspace (forward+backward finish inside one call, so reuse is safe)

    def __call__(self, batch, update=True):
        lib = _lib.load()
        model = self.model
        flat = model.flat_parameters()
        cfg = C.byref(model._cfg)
        st = _lib.stream_ptr()
        nl = build_neighbor_list(batch.pos, batch.batch, batch.z, model.cutoff, model.max_neighbors, batch.ptr)
        if nl.E == 0:
            raise IndexError("batch has no edges within the cutoff")
        dev = flat.device
        ws_bytes = lib.nq_painn_workspace_bytes(cfg, nl.N, nl.E, nl.B)
        if self._ws is None or self._ws.numel() < ws_bytes:
            self._ws = None                                            # release before growing (27 GB at B=1024)
            self._ws = torch.empty((int(ws_bytes * 1.08) + 4096) // 256 * 256, device=dev, dtype=torch.uint8)
        ws = self._ws
        energy = torch.empty(nl.B, device=dev, dtype=torch.float32)
        forces = torch.empty(nl.N, 3, device=dev, dtype=torch.float32)
        gE, gF = torch.empty_like(energy), torch.empty_like(forces)
        _lib.check(lib.nq_painn_forward(cfg, _lib.ptr(flat), _lib.ptr(model.radial_basis.rbf.offset), C.byref(nl.c), _lib.ptr(ws), ws_bytes,
                                        _lib.ptr(energy), _lib.ptr(forces), st))
        _lib.check(lib.nq_loss_l1_l2(_lib.ptr(energy), _lib.ptr(batch.y), nl.B, _lib.ptr(forces), _lib.ptr(batch.forces), nl.N, self.ce, self.cf,
                                     _lib.ptr(self.loss), _lib.ptr(gE), _lib.ptr(gF), st))
        _lib.check(lib.nq_painn_backward(cfg, _lib.ptr(flat), _lib.ptr(model.radial_basis.rbf.offset), C.byref(nl.c), _lib.ptr(ws), ws_bytes,
                                         _lib.ptr(gE), _lib.ptr(gF),
                                         _lib.ptr(self.grad), st))
        nqdist.allreduce_mean_(self.grad, self.group)
        if update:
            self.t += 1
            _lib.check(lib.nq_adamw_step(_lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.m), _lib.ptr(self.v), flat.numel(),
                                         float(self.max_norm or 0.0), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                         _lib.ptr(self.scratch), st))
        self.energy, self.forces = energy, forces
        model._last_ws, model._last_nl = ws, nl
        return self.loss
