"""Host-side mirror of the reference PaiNN plugin surface over libnablaq (HIP, gfx950).

Mirrors ``nablaDFT.painn_pyg.PaiNN`` (/root/reference/nablaDFT/painn_pyg/painn.py:22-148):
same constructor arguments (painn.py:28-45; config/model/painn-oc.yaml:4-20), same parameter
names/shapes (state_dict-compatible), same ``forward(data) -> (energy, forces)``
contract on a PyG-style batch (``data.pos, data.z, data.batch`` [, ``data.ptr``]).

All arithmetic happens in hand-written HIP kernels behind the C ABI (include/nablaq.h); torch is
used for device memory, streams and autograd plumbing only.  There is no CPU path: the module
raises if libnablaq.so is missing or the tensors are not on a GPU.
"""
import ctypes as C
import weakref
from typing import Dict, Optional, Union

import torch
from torch import nn

from . import _lib


class _WorkspaceToken:
    """Lives on the autograd ctx of the forward that owns the cached workspace (PaiNN._take_workspace)."""


class NeighborList:
    """Neighbour list of one batch in engine layout + (optionally) the reference's canonical outputs."""

    def __init__(self):
        self.N = self.B = self.E = 0
        self.t = {}          # name -> torch tensor (keeps device memory alive)
        self.c = None        # ctypes Graph
        self.edge_index = self.edge_dist = self.edge_vector = self.id_swap = self.neighbors = None


def build_neighbor_list(pos: torch.Tensor, batch: torch.Tensor, z: Optional[torch.Tensor], cutoff: float, max_neighbors: int,
                        ptr: Optional[torch.Tensor] = None, canonical: bool = False) -> NeighborList:
    """radius_graph + symmetrize_edges + edge geometry of the reference (painn.py:306-432) in two kernel
    launches (graph.hip).  ``canonical=True`` also materialises the reference's return values
    (edge_index int64 [2,E], neighbors, edge_dist, edge_vector, id_swap) in the reference's edge order."""
    lib = _lib.load()
    if not pos.is_cuda:
        raise RuntimeError("nabladft_amd runs on MI355X only: tensors must be on a cuda (HIP) device")
    dev = pos.device
    pos32 = pos.detach().to(torch.float32).contiguous()
    N = pos32.shape[0]
    if ptr is None:
        B = int(batch[-1].item()) + 1 if N > 0 else 0
        counts = torch.bincount(batch, minlength=B)
        ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    else:
        B = ptr.numel() - 1
        counts = ptr[1:] - ptr[:-1]
    mol_ptr = ptr.to(device=dev, dtype=torch.int32).contiguous()
    max_n = int(counts.max().item()) if B > 0 else 0
    nl = NeighborList()
    nl.N, nl.B = N, B
    i32 = dict(device=dev, dtype=torch.int32)
    deg, lowdeg = torch.empty(N, **i32), torch.empty(N, **i32)
    row_ptr, lowptr = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
    e_host = C.c_int32(0)
    st = _lib.stream_ptr()
    _lib.check(lib.nq_graph_count(_lib.ptr(pos32), _lib.ptr(mol_ptr), N, B, max_n, float(cutoff), int(max_neighbors), _lib.ptr(deg),
                                  _lib.ptr(lowdeg), _lib.ptr(row_ptr), _lib.ptr(lowptr), C.byref(e_host), st))
    E = int(e_host.value)
    nl.E = E
    col, dst, rev, s2c = (torch.empty(E, **i32) for _ in range(4))
    geom = torch.empty(E, 4, device=dev, dtype=torch.float32)
    atom_mol = torch.empty(N, **i32)
    neighbors = torch.empty(B, device=dev, dtype=torch.int64)
    if canonical:
        nl.edge_index = torch.empty(2, E, device=dev, dtype=torch.int64)
        nl.edge_dist = torch.empty(E, device=dev, dtype=torch.float32)
        nl.edge_vector = torch.empty(E, 3, device=dev, dtype=torch.float32)
        nl.id_swap = torch.empty(E, device=dev, dtype=torch.int64)
    if E > 0:
        _lib.check(lib.nq_graph_fill(_lib.ptr(pos32), _lib.ptr(mol_ptr), N, B, E, max_n, float(cutoff), int(max_neighbors),
                                     _lib.ptr(row_ptr), _lib.ptr(lowptr), _lib.ptr(col), _lib.ptr(dst), _lib.ptr(rev), _lib.ptr(geom),
                                     _lib.ptr(s2c), _lib.ptr(atom_mol), _lib.ptr(nl.edge_index), _lib.ptr(nl.edge_dist),
                                     _lib.ptr(nl.edge_vector), _lib.ptr(nl.id_swap), _lib.ptr(neighbors), st))
    else:
        neighbors.zero_()
    nl.neighbors = neighbors
    z32 = None if z is None else z.to(device=dev, dtype=torch.int32).contiguous()
    nl.t = dict(pos=pos32, mol_ptr=mol_ptr, row_ptr=row_ptr, lowptr=lowptr, col=col, dst=dst, rev=rev, geom=geom, slot2canon=s2c,
                atom_mol=atom_mol, z=z32, deg=deg)
    g = _lib.Graph()
    g.N, g.B, g.E, g.max_mol_atoms = N, B, E, max_n
    g.mol_ptr, g.row_ptr, g.col, g.dst = mol_ptr.data_ptr(), row_ptr.data_ptr(), col.data_ptr(), dst.data_ptr()
    g.rev, g.geom, g.atom_mol = rev.data_ptr(), geom.data_ptr(), atom_mol.data_ptr()
    g.z = z32.data_ptr() if z32 is not None else None
    g.lowptr = lowptr.data_ptr()
    nl.c = g
    return nl


class _EnergyForces(torch.autograd.Function):
    """(energy, forces) = f(params); backward runs the tangent + dual-reverse sweeps of the engine.
    Gradient w.r.t. positions is not provided (the reference never uses it)."""

    @staticmethod
    def forward(ctx, model, nl, want_forces, *params):
        lib = _lib.load()
        flat = model._flat
        dev = flat.device
        ws_bytes = lib.nq_painn_workspace_bytes(C.byref(model._cfg), nl.N, nl.E, nl.B)
        ws = model._take_workspace(ws_bytes, dev, ctx)
        energy = torch.empty(nl.B, device=dev, dtype=torch.float32)
        forces = torch.empty(nl.N, 3, device=dev, dtype=torch.float32) if want_forces else None
        _lib.check(lib.nq_painn_forward(C.byref(model._cfg), _lib.ptr(flat), _lib.ptr(model.radial_basis.engine_buffer()), C.byref(nl.c),
                                        _lib.ptr(ws), ws_bytes, _lib.ptr(energy), _lib.ptr(forces), _lib.stream_ptr()))
        ctx.model, ctx.nl, ctx.ws, ctx.ws_bytes, ctx.want_forces = model, nl, ws, ws_bytes, want_forces
        model._last_ws, model._last_nl = ws, nl
        if want_forces:
            return energy, forces
        return energy, energy.new_zeros(0)

    @staticmethod
    def backward(ctx, g_energy, g_forces):
        lib = _lib.load()
        model, nl = ctx.model, ctx.nl
        flat = model._flat
        grad_flat = torch.empty_like(flat)
        ge = None if g_energy is None else g_energy.to(torch.float32).contiguous()
        gf = None if (g_forces is None or not ctx.want_forces) else g_forces.to(torch.float32).contiguous()
        _lib.check(lib.nq_painn_backward(C.byref(model._cfg), _lib.ptr(flat), _lib.ptr(model.radial_basis.engine_buffer()), C.byref(nl.c),
                                         _lib.ptr(ctx.ws), ctx.ws_bytes,
                                         _lib.ptr(ge), _lib.ptr(gf), _lib.ptr(grad_flat), _lib.stream_ptr()))
        model._last_grad_flat = grad_flat
        model._release_workspace(ctx.ws, ctx)
        grads = tuple(grad_flat[o:o + n].view(s) for (o, n, s) in model._param_slices)
        return (None, None, None) + grads


class _EnergyBackbone(torch.autograd.Function):
    """direct_forces=True (painn.py:130-133): (energy, x_L, vec_L) = f(params); backward = first-order reverse of the engine seeded with
    dL/dE and the adjoints of the final node state coming from the PaiNNOutput head."""

    @staticmethod
    def forward(ctx, model, nl, *params):
        lib = _lib.load()
        flat = model._flat
        dev = flat.device
        ws_bytes = lib.nq_painn_workspace_bytes(C.byref(model._cfg), nl.N, nl.E, nl.B)
        ws = model._take_workspace(ws_bytes, dev, ctx)
        energy = torch.empty(nl.B, device=dev, dtype=torch.float32)
        _lib.check(lib.nq_painn_forward(C.byref(model._cfg), _lib.ptr(flat), _lib.ptr(model.radial_basis.engine_buffer()), C.byref(nl.c),
                                        _lib.ptr(ws), ws_bytes, _lib.ptr(energy), None, _lib.stream_ptr()))
        model._last_ws, model._last_nl = ws, nl
        F, L = model.hidden_channels, model.num_layers
        x = model.workspace_view("x_in", L).view(nl.N, F).clone()
        vec = model.workspace_view("vec_in", L).view(nl.N, 3, F).clone()
        ctx.model, ctx.nl, ctx.ws, ctx.ws_bytes = model, nl, ws, ws_bytes
        return energy, x, vec

    @staticmethod
    def backward(ctx, g_energy, g_x, g_vec):
        lib = _lib.load()
        model, nl = ctx.model, ctx.nl
        flat = model._flat
        grad_flat = torch.zeros_like(flat)                       # the head's own parameters (tail of the buffer) get theirs from autograd
        cont = lambda t: None if t is None else t.to(torch.float32).contiguous()
        _lib.check(lib.nq_painn_backward_seeded(C.byref(model._cfg), _lib.ptr(flat), _lib.ptr(model.radial_basis.engine_buffer()), C.byref(nl.c),
                                                _lib.ptr(ctx.ws), ctx.ws_bytes, _lib.ptr(cont(g_energy)), _lib.ptr(cont(g_x)), _lib.ptr(cont(g_vec)),
                                                _lib.ptr(grad_flat), _lib.stream_ptr()))
        model._last_grad_flat = grad_flat
        model._release_workspace(ctx.ws, ctx)
        n_engine = model._n_engine_params
        grads = tuple((grad_flat[o:o + n].view(s) if o < n_engine else None) for (o, n, s) in model._param_slices)
        return (None, None) + grads


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T (+ b) through the engine's fp32 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, W, b):
        lib = _lib.load()
        x, W = x.to(torch.float32).contiguous(), W.to(torch.float32).contiguous()
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_forward(_lib.ptr(x), _lib.ptr(W), _lib.ptr(None if b is None else b.to(torch.float32).contiguous()), _lib.ptr(y), None, M, N, K,
                                         _lib.stream_ptr()))
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        M, K = x.shape
        N = W.shape[0]
        gx, gW = torch.empty_like(x), torch.empty_like(W)
        _lib.check(lib.nq_linear_input_grad(_lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, 0, _lib.stream_ptr()))
        scr = torch.empty(int(lib.nq_weight_grad_scratch_floats(M, N, K)) + 64, device=x.device, dtype=torch.float32)
        _lib.check(lib.nq_linear_weight_grad(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gW), M, N, K, _lib.ptr(scr), _lib.stream_ptr()))
        return gx, gW, (g.sum(0) if ctx.has_bias else None)


class _ScaledSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        out = torch.empty_like(z)
        _lib.check(_lib.load().nq_scaled_silu(_lib.ptr(z), None, _lib.ptr(out), z.numel(), _lib.stream_ptr()))
        ctx.save_for_backward(z)
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(z)
        _lib.check(_lib.load().nq_scaled_silu(_lib.ptr(z), _lib.ptr(g), _lib.ptr(out), z.numel(), _lib.stream_ptr()))
        return out


class _GebCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v1):
        x, v1 = x.contiguous(), v1.contiguous()
        N, h = x.shape
        cat = torch.empty(N, 2 * h, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().nq_geb_cat(_lib.ptr(x), _lib.ptr(v1), N, h, _lib.ptr(cat), _lib.stream_ptr()))
        ctx.save_for_backward(v1)
        return cat

    @staticmethod
    def backward(ctx, g):
        (v1,) = ctx.saved_tensors
        N, _, h = v1.shape
        g = g.contiguous()
        gx, gv1 = torch.empty(N, h, device=g.device, dtype=torch.float32), torch.empty_like(v1)
        _lib.check(_lib.load().nq_geb_cat_backward(_lib.ptr(g), _lib.ptr(v1), N, h, _lib.ptr(gx), _lib.ptr(gv1), _lib.stream_ptr()))
        return gx, gv1


class _GebGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o2, v2):
        o2, v2 = o2.contiguous(), v2.contiguous()
        N, _, o = v2.shape
        xout, vout = torch.empty(N, o, device=o2.device, dtype=torch.float32), torch.empty_like(v2)
        _lib.check(_lib.load().nq_geb_gate(_lib.ptr(o2), _lib.ptr(v2), N, o, _lib.ptr(xout), _lib.ptr(vout), _lib.stream_ptr()))
        ctx.save_for_backward(o2, v2)
        return xout, vout

    @staticmethod
    def backward(ctx, gx, gv):
        o2, v2 = ctx.saved_tensors
        N, _, o = v2.shape
        go2, gv2 = torch.empty_like(o2), torch.empty_like(v2)
        _lib.check(_lib.load().nq_geb_gate_backward(_lib.ptr(o2), _lib.ptr(v2), _lib.ptr(gx.contiguous()), _lib.ptr(gv.contiguous()), N, o, _lib.ptr(go2),
                                                    _lib.ptr(gv2), _lib.stream_ptr()))
        return go2, gv2


class _GatedEquivariantBlock(nn.Module):
    """Parameter holder + HIP evaluation of GatedEquivariantBlock (painn.py:583-620)."""

    def __init__(self, hidden_channels, out_channels):
        super().__init__()
        self.out_channels = out_channels
        self.vec1_proj = nn.Linear(hidden_channels, hidden_channels, bias=False)
        self.vec2_proj = nn.Linear(hidden_channels, out_channels, bias=False)
        self.update_net = nn.Sequential(nn.Linear(hidden_channels * 2, hidden_channels), nn.Identity(), nn.Linear(hidden_channels, out_channels * 2))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.vec1_proj.weight)
        nn.init.xavier_uniform_(self.vec2_proj.weight)
        nn.init.xavier_uniform_(self.update_net[0].weight)
        self.update_net[0].bias.data.fill_(0)
        nn.init.xavier_uniform_(self.update_net[2].weight)
        self.update_net[2].bias.data.fill_(0)

    def forward(self, x, v):
        N, _, h = v.shape
        v1 = _LinearFn.apply(v.reshape(3 * N, h), self.vec1_proj.weight, None).view(N, 3, h)
        v2 = _LinearFn.apply(v.reshape(3 * N, h), self.vec2_proj.weight, None).view(N, 3, self.out_channels)
        u = _ScaledSiluFn.apply(_LinearFn.apply(_GebCatFn.apply(x, v1), self.update_net[0].weight, self.update_net[0].bias))
        o2 = _LinearFn.apply(u, self.update_net[2].weight, self.update_net[2].bias)
        return _GebGateFn.apply(o2, v2)


class _PaiNNOutput(nn.Module):
    """PaiNNOutput (painn.py:551-579): two gated equivariant blocks hidden -> hidden/2 -> 1; returns forces [N, 3]."""

    def __init__(self, hidden_channels):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.output_network = nn.ModuleList([_GatedEquivariantBlock(hidden_channels, hidden_channels // 2),
                                             _GatedEquivariantBlock(hidden_channels // 2, 1)])

    def forward(self, x, vec):
        for layer in self.output_network:
            x, vec = layer(x, vec)
        return vec.squeeze(-1)


class _GaussianSmearing(nn.Module):
    """Holds the ``offset`` buffer / ``coeff`` exactly as torch_geometric's GaussianSmearing(start, stop, n)."""

    def __init__(self, start=0.0, stop=1.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2 if num_gaussians > 1 else -0.5
        self.register_buffer("offset", offset)


class _RadialBasis(nn.Module):
    """Configuration holder for RadialBasis (layers.py:129-185); evaluated inside the HIP kernels."""

    def __init__(self, num_radial, cutoff, rbf, envelope):
        super().__init__()
        env = dict(envelope)
        name = env.pop("name").lower()
        if name == "polynomial":
            self.exponent = int(env.get("exponent", 5))
            assert self.exponent > 0
        elif name == "exponential":                     # ExponentialEnvelope (layers.py:36-48); exponent 0 in the C ABI
            if env:
                raise TypeError(f"ExponentialEnvelope takes no hyper-parameters, got {env}")
            self.exponent = 0
        else:
            raise ValueError(f"Unknown envelope function '{name}'.")
        rb = dict(rbf)
        rbf_name = rb.pop("name").lower()
        self.inv_cutoff = 1 / cutoff
        if rbf_name == "gaussian":
            self.rbf_type = 0
            self.rbf = _GaussianSmearing(start=0, stop=1, num_gaussians=num_radial, **rb)
        elif rbf_name == "spherical_bessel":
            self.rbf_type = 1
            self.rbf = _SphericalBesselBasis(num_radial=num_radial, cutoff=cutoff, **rb)
        elif rbf_name == "bernstein":
            self.rbf_type = 2
            self.rbf = _BernsteinBasis(num_radial=num_radial, **rb)
        else:
            raise ValueError(f"Unknown radial basis function '{rbf_name}'.")

    def engine_buffer(self):
        """The fp32 buffer the C ABI takes as ``rbf_offsets``: Gaussian offsets / Bernstein binomial prefactors / (Bessel: unused)."""
        return self.rbf.offset if self.rbf_type == 0 else (self.rbf.prefactor if self.rbf_type == 2 else self.rbf.frequencies.detach())


class _SphericalBesselBasis(nn.Module):
    """Parameter holder of SphericalBesselBasis (layers.py:51-80): learnable ``frequencies`` at k*pi."""

    def __init__(self, num_radial, cutoff):
        super().__init__()
        import math
        self.norm_const = math.sqrt(2 / (cutoff ** 3))
        self.coeff = 0.0
        self.frequencies = nn.Parameter(torch.pi * torch.arange(1, num_radial + 1, dtype=torch.float32))


class _BernsteinBasis(nn.Module):
    """Parameter holder of BernsteinBasis (layers.py:83-126): learnable ``pregamma``; ``prefactor`` = binomial coefficients (non-persistent)."""

    def __init__(self, num_radial, pregamma_initial: float = 0.45264):
        super().__init__()
        import math
        self.coeff = 0.0
        self.register_buffer("prefactor", torch.tensor([math.comb(num_radial - 1, k) for k in range(num_radial)], dtype=torch.float), persistent=False)
        self.pregamma = nn.Parameter(torch.tensor(pregamma_initial, dtype=torch.float))


class _AtomEmbedding(nn.Module):
    def __init__(self, emb_size, num_elements):
        super().__init__()
        self.emb_size = emb_size
        self.embeddings = nn.Embedding(num_elements, emb_size)
        nn.init.uniform_(self.embeddings.weight, a=-(3 ** 0.5), b=3 ** 0.5)  # layers.py:213


class _Message(nn.Module):
    def __init__(self, F, R):
        super().__init__()
        self.x_proj = nn.Sequential(nn.Linear(F, F), nn.SiLU(), nn.Linear(F, 3 * F))
        self.rbf_proj = nn.Linear(R, 3 * F)
        for lin in (self.x_proj[0], self.x_proj[2], self.rbf_proj):  # painn.py:467-473
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)


class _Update(nn.Module):
    def __init__(self, F):
        super().__init__()
        self.vec_proj = nn.Linear(F, 2 * F, bias=False)
        self.xvec_proj = nn.Sequential(nn.Linear(2 * F, F), nn.SiLU(), nn.Linear(F, 3 * F))
        nn.init.xavier_uniform_(self.vec_proj.weight)  # painn.py:528-533
        for lin in (self.xvec_proj[0], self.xvec_proj[2]):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)


class PaiNN(nn.Module):
    r"""Drop-in for ``nablaDFT.painn_pyg.PaiNN`` (painn.py:22-148) computing on MI355X through libnablaq."""

    def __init__(
        self,
        hidden_channels: int = 512,
        num_layers: int = 6,
        num_rbf: int = 128,
        cutoff: float = 12.0,
        max_neighbors: int = 50,
        rbf: Dict[str, str] = {"name": "gaussian"},
        envelope: Dict[str, Union[str, int]] = {"name": "polynomial", "exponent": 5},
        regress_forces: bool = True,
        direct_forces: bool = True,
        use_pbc: bool = True,
        otf_graph: bool = True,
        num_elements: int = 83,
    ) -> None:
        super().__init__()
        if use_pbc:
            raise NotImplementedError("nabladft_amd: use_pbc=True (radius_graph_pbc, utils.py) is outside the nablaDFT hot path "
                                      "(config/model/painn-oc.yaml:18 sets use_pbc: false)")
        if hidden_channels % 64 != 0 or not (64 <= hidden_channels <= 1024):
            raise ValueError("hidden_channels must be a multiple of 64 in [64, 1024] (one thread per channel, wavefront = 64)")
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.num_rbf = num_rbf
        self.cutoff = cutoff
        self.max_neighbors = max_neighbors
        self.regress_forces = regress_forces
        self.direct_forces = direct_forces
        self.otf_graph = otf_graph
        self.use_pbc = use_pbc
        self.symmetric_edge_symmetrization = False

        self.atom_emb = _AtomEmbedding(hidden_channels, num_elements)
        self.radial_basis = _RadialBasis(num_radial=num_rbf, cutoff=cutoff, rbf=rbf, envelope=envelope)
        self.message_layers = nn.ModuleList([_Message(hidden_channels, num_rbf) for _ in range(num_layers)])
        self.update_layers = nn.ModuleList([_Update(hidden_channels) for _ in range(num_layers)])
        self.out_energy = nn.Sequential(nn.Linear(hidden_channels, hidden_channels // 2), nn.SiLU(),
                                        nn.Linear(hidden_channels // 2, 1))
        if self.regress_forces is True and self.direct_forces is True:
            self.out_forces = _PaiNNOutput(hidden_channels)        # painn.py:84-85; its parameters follow the engine's in the flat buffer
        self.reset_parameters()

        self._flat = None
        self._n_engine_params = None
        self._param_slices = None
        self._last_ws = self._last_nl = self._last_grad_flat = None
        self._ws_cache, self._ws_owner = None, None
        cfg = _lib.PainnCfg()
        cfg.hidden_channels, cfg.num_layers, cfg.num_rbf, cfg.num_elements = hidden_channels, num_layers, num_rbf, num_elements
        cfg.max_neighbors, cfg.envelope_exponent = max_neighbors, self.radial_basis.exponent
        cfg.cutoff, cfg.rbf_coeff = float(cutoff), float(self.radial_basis.rbf.coeff)
        cfg.rbf_type = self.radial_basis.rbf_type
        self._cfg = cfg

    def reset_parameters(self) -> None:
        for lin in (self.out_energy[0], self.out_energy[2]):  # painn.py:150-154
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)

    def __getstate__(self):
        # transient engine handles (device pointers) are rebuilt lazily; keep copies/pickles clean
        state = self.__dict__.copy()
        for k in ("_flat", "_param_slices", "_last_ws", "_last_nl", "_last_grad_flat", "_ws_cache", "_n_engine_params"):
            state[k] = None
        state["_ws_owner"] = None
        return state

    # ---- flat parameter buffer (state_dict order) the C ABI consumes -------------------------------
    def flat_parameters(self) -> torch.Tensor:
        """Returns the contiguous fp32 buffer all parameters are views of (re-flattens after .to()/.cuda())."""
        params = list(self.parameters())
        ok = self._flat is not None and self._flat.device == params[0].device
        if ok:
            base = self._flat.data_ptr()
            for p, (o, n, _) in zip(params, self._param_slices):
                if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                    ok = False
                    break
        if not ok:
            flat = torch.cat([p.detach().to(torch.float32).reshape(-1) for p in params]).contiguous()
            slices, o = [], 0
            for p in params:
                n = p.numel()
                slices.append((o, n, tuple(p.shape)))
                p.data = flat[o:o + n].view(p.shape)
                o += n
            expect = _lib.load().nq_painn_num_params(C.byref(self._cfg))
            head = sum(p.numel() for p in self.out_forces.parameters()) if hasattr(self, "out_forces") else 0
            if o != expect + head:
                raise RuntimeError(f"parameter count {o} != engine layout {expect} (+ {head} force-head parameters)")
            self._n_engine_params = expect
            self._flat, self._param_slices = flat, slices
        return self._flat

    # ---- workspace cache: one grow-only buffer.  It is OWNED by the autograd node of the forward that took it (a token stored on that node's
    # ctx, tracked here by weak reference): the buffer is free again as soon as that backward has run OR the node has died (outputs dropped, a
    # forward under grad mode that never runs backward -- e.g. the optimisation calculator calling model(batch) -- or an exception in between).
    def _ws_in_use(self):
        owner = self._ws_owner
        return owner is not None and owner() is not None

    def _take_workspace(self, nbytes, dev, ctx=None):
        def claim():
            if ctx is not None and any(ctx.needs_input_grad):     # (grad mode itself is always off inside Function.forward)
                ctx.ws_token = _WorkspaceToken()
                self._ws_owner = weakref.ref(ctx.ws_token)
            else:
                self._ws_owner = None
        ws = self._ws_cache
        if ws is not None and not self._ws_in_use() and ws.device == dev and ws.numel() >= nbytes:
            claim()
            return ws
        if not self._ws_in_use():
            self._ws_cache = None                                      # drop the old buffer before growing
            self._ws_cache = torch.empty((int(nbytes * 1.08) + 4096) // 256 * 256, device=dev, dtype=torch.uint8)
            claim()
            return self._ws_cache
        return torch.empty(nbytes, device=dev, dtype=torch.uint8)     # a second forward while the first one's backward is still possible

    def _release_workspace(self, ws, ctx=None):
        if ws is self._ws_cache:
            self._ws_owner = None
            if ctx is not None:
                ctx.ws_token = None

    # ---- reference API -----------------------------------------------------------------------------------
    def generate_graph_values(self, data):
        """painn.py:306-349: (edge_index, neighbors, edge_dist, edge_vector, id_swap) in the reference's order."""
        nl = build_neighbor_list(data.pos, data.batch, None, self.cutoff, self.max_neighbors, getattr(data, "ptr", None), canonical=True)
        self._check_neighbors(nl)
        return nl.edge_index, nl.neighbors, nl.edge_dist, nl.edge_vector, nl.id_swap

    def _check_neighbors(self, nl):
        if nl.E == 0:
            # the reference fails in repeat_blocks (utils.py:96) when the whole batch has no edge
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0 (batch has no edges within the cutoff)")
        empty = nl.neighbors == 0
        if torch.any(empty):  # painn.py:323-325
            print(f"An image has no neighbors! #images = {empty.sum().item()}")

    def forward(self, data):
        pos, batch = data.pos, data.batch
        z = data.z.long()
        assert z.dim() == 1 and z.dtype == torch.long  # painn.py:106
        if not pos.is_cuda:
            raise RuntimeError("nabladft_amd.PaiNN runs on MI355X only (no CPU fallback): move the batch to cuda")
        self.flat_parameters()
        nl = build_neighbor_list(pos, batch, z, self.cutoff, self.max_neighbors, getattr(data, "ptr", None))
        self._check_neighbors(nl)
        if self.regress_forces and self.direct_forces:              # painn.py:130-133
            energy, x, vec = _EnergyBackbone.apply(self, nl, *self.parameters())
            return energy, self.out_forces(x, vec)
        energy, forces = _EnergyForces.apply(self, nl, bool(self.regress_forces), *self.parameters())
        if self.regress_forces:
            return energy, forces
        return energy

    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(hidden_channels={self.hidden_channels}, num_layers={self.num_layers}, "
                f"num_rbf={self.num_rbf}, max_neighbors={self.max_neighbors}, cutoff={self.cutoff})")

    # ---- test hook: read a named engine buffer of the last forward/backward ---------------------------
    def workspace_view(self, name: str, layer: int = 0, tangent: bool = False) -> torch.Tensor:
        lib = _lib.load()
        nl = self._last_nl
        off, cnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.nq_painn_ws_lookup(C.byref(self._cfg), nl.N, nl.E, nl.B, name.encode(), layer, int(tangent), C.byref(off), C.byref(cnt)))
        return self._last_ws.view(torch.float32)[off.value:off.value + cnt.value]
