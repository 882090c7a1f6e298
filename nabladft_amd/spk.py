"""schnetpack-shaped mirror for the PaiNN potential nablaDFT instantiates in ``config/model/painn.yaml``
(/root/reference/config/model/painn.yaml:4-28): ``NeuralNetworkPotential(representation=PaiNN(...), input_modules=
[PairwiseDistances()], output_modules=[Atomwise(n_in=128, output_key="energy"), Forces()], postprocessors=[AddOffsets(...)])``.

Same constructor keywords as the schnetpack 2.0.4 classes of those names, so the yaml only swaps the ``schnetpack.`` prefixes
for ``nabladft_amd.spk.``; the arithmetic runs in the same HIP engine as the in-tree PaiNN: the two models
differ by row permutations of the weights and by the radial filter (cosine cutoff applied after the bias), see
oracle/spk_painn_ref.py.

STATUS -- row a12 of SURVEY.md section 8, **parity unpinned**: schnetpack is not part of the reference tree and is not
installed here; layer definitions, parameter names (``representation.interactions.{l}.interatomic_context_net.{0,1}``,
``representation.mixing.{l}.{intraatomic_context_net.{0,1},mu_channel_mix}``, ``representation.filter_net``,
``output_modules.0.outnet.{0,1}``) and the train/eval behaviour of post-processors follow SURVEY.md Appendix C (recalled,
unverified).  Checked against this repo's own restatement only (tests/test_spk_cpu.py, tests/test_spk_gpu.py).
``SchNet`` (config/model/schnet.yaml) runs on its own engine entry points (csrc/schnet.hip; restatement
oracle/spk_schnet_ref.py).  Not built: trainable / non-Gaussian radial bases, shared_interactions / shared_filters, atomrefs,
stress and the ``AtomisticTask`` wrapper (nablaDFT/ase_model/task.py).
"""
import ctypes as C
import math
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib
from .painn import build_neighbor_list

properties_Z, properties_R, properties_idx_m = "_atomic_numbers", "_positions", "_idx_m"


class Dense(nn.Linear):
    """schnetpack.nn.Dense: nn.Linear + optional activation (the activation runs inside the engine)."""

    def __init__(self, in_features, out_features, bias=True, activation=None):
        super().__init__(in_features, out_features, bias)
        self.activation = activation
        nn.init.xavier_uniform_(self.weight)
        if bias:
            nn.init.zeros_(self.bias)


class GaussianRBF(nn.Module):
    def __init__(self, n_rbf: int, cutoff: float, start: float = 0.0, trainable: bool = False):
        super().__init__()
        if trainable or start != 0.0:
            raise NotImplementedError("nabladft_amd.spk.GaussianRBF: only start=0, trainable=False (config/model/painn.yaml:11-13)")
        self.n_rbf, self.cutoff = n_rbf, cutoff
        offsets = torch.linspace(start, cutoff, n_rbf)
        self.register_buffer("offsets", offsets)
        self.register_buffer("widths", torch.abs(offsets[1] - offsets[0]) * torch.ones_like(offsets))


class CosineCutoff(nn.Module):
    def __init__(self, cutoff: float):
        super().__init__()
        self.register_buffer("cutoff", torch.tensor([float(cutoff)]))


class _Interaction(nn.Module):
    def __init__(self, F):
        super().__init__()
        self.interatomic_context_net = nn.Sequential(Dense(F, F, activation="silu"), Dense(F, 3 * F))


class _Mixing(nn.Module):
    def __init__(self, F):
        super().__init__()
        self.intraatomic_context_net = nn.Sequential(Dense(2 * F, F, activation="silu"), Dense(F, 3 * F))
        self.mu_channel_mix = Dense(F, 2 * F, bias=False)


class PaiNN(nn.Module):
    """schnetpack.representation.PaiNN(n_atom_basis, n_interactions, radial_basis, cutoff_fn, ...) -- parameter holder."""

    def __init__(self, n_atom_basis: int, n_interactions: int, radial_basis: nn.Module, cutoff_fn: Optional[nn.Module] = None,
                 activation=None, max_z: int = 101, shared_interactions: bool = False, shared_filters: bool = False, epsilon: float = 1e-8):
        super().__init__()
        if shared_interactions or shared_filters:
            raise NotImplementedError("nabladft_amd.spk.PaiNN: shared_interactions / shared_filters are not built")
        if not isinstance(radial_basis, GaussianRBF) or not isinstance(cutoff_fn, CosineCutoff):
            raise NotImplementedError("nabladft_amd.spk.PaiNN: needs GaussianRBF + CosineCutoff (config/model/painn.yaml:9-16)")
        if abs(epsilon - 1e-8) > 0:
            raise NotImplementedError("nabladft_amd.spk.PaiNN: epsilon is fixed to 1e-8 in the kernels")
        self.n_atom_basis, self.n_interactions, self.max_z = n_atom_basis, n_interactions, max_z
        self.radial_basis, self.cutoff_fn = radial_basis, cutoff_fn
        self.cutoff = float(cutoff_fn.cutoff)
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.filter_net = Dense(radial_basis.n_rbf, n_interactions * 3 * n_atom_basis)
        self.interactions = nn.ModuleList([_Interaction(n_atom_basis) for _ in range(n_interactions)])
        self.mixing = nn.ModuleList([_Mixing(n_atom_basis) for _ in range(n_interactions)])


class _SchNetInteraction(nn.Module):
    """schnetpack.representation.schnet.SchNetInteraction(n_atom_basis, n_rbf, n_filters): parameter holder."""

    def __init__(self, F, R):
        super().__init__()
        self.in2f = Dense(F, F, bias=False)
        self.f2out = nn.Sequential(Dense(F, F, activation="shifted_softplus"), Dense(F, F))
        self.filter_network = nn.Sequential(Dense(R, F, activation="shifted_softplus"), Dense(F, F))


class SchNet(nn.Module):
    """schnetpack.representation.SchNet(n_atom_basis, n_interactions, radial_basis, cutoff_fn, ...) -- parameter holder
    (config/model/schnet.yaml:6-16).  PARITY UNPINNED like the rest of this module (oracle/spk_schnet_ref.py)."""

    def __init__(self, n_atom_basis: int, n_interactions: int, radial_basis: nn.Module, cutoff_fn: nn.Module, n_filters: Optional[int] = None,
                 shared_interactions: bool = False, max_z: int = 101, activation=None):
        super().__init__()
        if shared_interactions:
            raise NotImplementedError("nabladft_amd.spk.SchNet: shared_interactions is not built")
        if n_filters not in (None, n_atom_basis):
            raise NotImplementedError("nabladft_amd.spk.SchNet: n_filters must equal n_atom_basis")
        if not isinstance(radial_basis, GaussianRBF) or not isinstance(cutoff_fn, CosineCutoff):
            raise NotImplementedError("nabladft_amd.spk.SchNet: needs GaussianRBF + CosineCutoff (config/model/schnet.yaml:9-16)")
        self.n_atom_basis, self.n_interactions, self.max_z = n_atom_basis, n_interactions, max_z
        self.radial_basis, self.cutoff_fn = radial_basis, cutoff_fn
        self.cutoff = float(cutoff_fn.cutoff)
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.interactions = nn.ModuleList([_SchNetInteraction(n_atom_basis, radial_basis.n_rbf) for _ in range(n_interactions)])


class PairwiseDistances(nn.Module):
    """Marker: the neighbour list and distances are rebuilt on the GPU inside the engine."""


class Atomwise(nn.Module):
    def __init__(self, n_in: int, n_out: int = 1, n_hidden=None, n_layers: int = 2, activation=None, aggregation_mode: str = "sum",
                 output_key: str = "y", per_atom_output_key: Optional[str] = None):
        super().__init__()
        if n_out != 1 or n_hidden is not None or n_layers != 2 or aggregation_mode != "sum" or per_atom_output_key is not None:
            raise NotImplementedError("nabladft_amd.spk.Atomwise: only the defaults used by config/model/painn.yaml:19-22")
        self.output_key = output_key
        self.model_outputs = [output_key]
        self.outnet = nn.Sequential(Dense(n_in, n_in // 2, activation="silu"), Dense(n_in // 2, 1))


class Forces(nn.Module):
    def __init__(self, calc_forces: bool = True, calc_stress: bool = False, energy_key: str = "energy", force_key: str = "forces"):
        super().__init__()
        if calc_stress:
            raise NotImplementedError("nabladft_amd.spk.Forces: stress is not built")
        self.calc_forces, self.energy_key, self.force_key = calc_forces, energy_key, force_key
        self.model_outputs = [force_key] if calc_forces else []


class AddOffsets(nn.Module):
    """schnetpack.transform.AddOffsets(property, add_mean): adds mean * n_atoms at inference (the mean comes from the datamodule)."""

    def __init__(self, property: str, add_mean: bool = False, add_atomrefs: bool = False, is_extensive: bool = True, zmax: int = 100,
                 atomrefs=None, propery_mean=None):
        super().__init__()
        if add_atomrefs:
            raise NotImplementedError("nabladft_amd.spk.AddOffsets: atomrefs are not built")
        self._property, self.add_mean, self.is_extensive = property, add_mean, is_extensive
        self.register_buffer("mean", torch.zeros(1) if propery_mean is None else torch.as_tensor(propery_mean, dtype=torch.float32).reshape(1))

    def forward(self, inputs):
        if self.add_mean:
            n = torch.bincount(inputs[properties_idx_m]).to(inputs[self._property].dtype) if self.is_extensive else 1.0
            inputs[self._property] = inputs[self._property] + self.mean.to(inputs[self._property].dtype) * n
        return inputs


def _spk_index(F, L, R, max_z, device):
    """Gather index engine_flat = spk_flat[index] (same construction as oracle/spk_painn_ref.py:spk_to_engine_index, kept
    separate because the product must not import the oracle)."""
    sizes = [("emb", max_z, F), ("fw", L * 3 * F, R), ("fb", L * 3 * F, 1)]
    for l in range(L):
        sizes += [(f"i{l}w0", F, F), (f"i{l}b0", F, 1), (f"i{l}w1", 3 * F, F), (f"i{l}b1", 3 * F, 1)]
    for l in range(L):
        sizes += [(f"m{l}w0", F, 2 * F), (f"m{l}b0", F, 1), (f"m{l}w1", 3 * F, F), (f"m{l}b1", 3 * F, 1), (f"m{l}u", 2 * F, F)]
    sizes += [("o0w", F // 2, F), ("o0b", F // 2, 1), ("o1w", 1, F // 2), ("o1b", 1, 1)]
    off, o = {}, 0
    for name, r, c in sizes:
        off[name] = (o, r, c)
        o += r * c
    ar = torch.arange

    def rows(name, idx):
        o0, _, c = off[name]
        return (o0 + idx[:, None] * c + ar(c)[None, :]).reshape(-1)

    def whole(name):
        o0, r, c = off[name]
        return ar(o0, o0 + r * c)

    p021 = torch.cat([ar(F), ar(F) + 2 * F, ar(F) + F])
    mix = torch.cat([ar(F) + F, ar(F)])
    idx = [rows("emb", ar(1, max_z))]
    for l in range(L):
        idx += [whole(f"i{l}w0"), whole(f"i{l}b0"), rows(f"i{l}w1", p021), rows(f"i{l}b1", p021), rows("fw", l * 3 * F + p021),
                rows("fb", l * 3 * F + p021)]
    for l in range(L):
        idx += [rows(f"m{l}u", mix), whole(f"m{l}w0"), whole(f"m{l}b0"), rows(f"m{l}w1", p021), rows(f"m{l}b1", p021)]
    idx += [whole("o0w"), whole("o0b"), whole("o1w"), whole("o1b")]
    return torch.cat(idx).to(device), o


class _SpkEnergyForces(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pot, nl, want_forces, *params):
        lib = _lib.load()
        flat = torch.cat([p.detach().to(torch.float32).reshape(-1) for p in params])
        if pot._index is not None:
            flat = flat.index_select(0, pot._index)
        dev = flat.device
        ws_fn, fwd_fn, _ = pot._entry_points(lib)
        ws_bytes = ws_fn(C.byref(pot._cfg), nl.N, nl.E, nl.B)
        ws = torch.empty((int(ws_bytes) + 255) // 256 * 256, device=dev, dtype=torch.uint8)
        energy = torch.empty(nl.B, device=dev, dtype=torch.float32)
        forces = torch.empty(nl.N, 3, device=dev, dtype=torch.float32) if want_forces else None
        offsets = pot.representation.radial_basis.offsets
        _lib.check(fwd_fn(C.byref(pot._cfg), _lib.ptr(flat), _lib.ptr(offsets), C.byref(nl.c), _lib.ptr(ws), ws_bytes,
                          _lib.ptr(energy), _lib.ptr(forces), _lib.stream_ptr()))
        ctx.pot, ctx.nl, ctx.ws, ctx.ws_bytes, ctx.flat, ctx.want_forces = pot, nl, ws, ws_bytes, flat, want_forces
        ctx.shapes = [tuple(p.shape) for p in params]
        return (energy, forces) if want_forces else (energy, energy.new_zeros(0))

    @staticmethod
    def backward(ctx, g_energy, g_forces):
        lib = _lib.load()
        pot, nl = ctx.pot, ctx.nl
        grad_engine = torch.empty_like(ctx.flat)
        ge = None if g_energy is None else g_energy.to(torch.float32).contiguous()
        gf = None if (g_forces is None or not ctx.want_forces) else g_forces.to(torch.float32).contiguous()
        _lib.check(pot._entry_points(lib)[2](C.byref(pot._cfg), _lib.ptr(ctx.flat), _lib.ptr(pot.representation.radial_basis.offsets), C.byref(nl.c),
                                             _lib.ptr(ctx.ws), ctx.ws_bytes, _lib.ptr(ge), _lib.ptr(gf), _lib.ptr(grad_engine), _lib.stream_ptr()))
        grad_spk = grad_engine if pot._index is None else \
            torch.zeros(pot._n_spk, device=grad_engine.device, dtype=torch.float32).index_add_(0, pot._index, grad_engine)
        out, o = [], 0
        for shp in ctx.shapes:
            n = math.prod(shp)
            out.append(grad_spk[o:o + n].view(shp))
            o += n
        return (None, None, None) + tuple(out)


class NeuralNetworkPotential(nn.Module):
    """schnetpack.model.NeuralNetworkPotential(representation, input_modules, output_modules, postprocessors, ...)."""

    def __init__(self, representation: nn.Module, input_modules: List[nn.Module] = None, output_modules: List[nn.Module] = None,
                 postprocessors: Optional[List[nn.Module]] = None, input_dtype_str: str = "float32", do_postprocessing: bool = True):
        super().__init__()
        if not isinstance(representation, (PaiNN, SchNet)):
            raise NotImplementedError("nabladft_amd.spk.NeuralNetworkPotential: representation must be nabladft_amd.spk.PaiNN or .SchNet")
        self.representation = representation
        self.input_modules = nn.ModuleList(input_modules or [])
        self.output_modules = nn.ModuleList(output_modules or [])
        self.postprocessors = nn.ModuleList(postprocessors or [])
        self.do_postprocessing = do_postprocessing
        atomwise = [m for m in self.output_modules if isinstance(m, Atomwise)]
        forces = [m for m in self.output_modules if isinstance(m, Forces)]
        if len(atomwise) != 1 or len(forces) > 1 or len(atomwise) + len(forces) != len(self.output_modules):
            raise NotImplementedError("output_modules must be [Atomwise(...)] or [Atomwise(...), Forces()]")
        # plain indices, not attributes: a second reference to the sub-modules would duplicate their state_dict entries
        self._i_atomwise = [i for i, m in enumerate(self.output_modules) if isinstance(m, Atomwise)][0]
        self._i_forces = ([i for i, m in enumerate(self.output_modules) if isinstance(m, Forces)] or [None])[0]
        if self._forces is not None and self._forces.energy_key != self._atomwise.output_key:
            raise ValueError("Forces.energy_key must name the Atomwise output")
        self.model_outputs = [k for m in self.output_modules for k in m.model_outputs]
        rep = representation
        F, L, R = rep.n_atom_basis, rep.n_interactions, rep.radial_basis.n_rbf
        if F % 64 != 0 or F not in (64, 128, 256):
            raise ValueError("n_atom_basis must be 64, 128 or 256 (fused-filter engine path)")
        width = float((torch.linspace(0.0, rep.cutoff, R)[1]).item())
        self._kind = "painn" if isinstance(rep, PaiNN) else "schnet"
        if self._kind == "painn":
            cfg = _lib.PainnCfg()
            cfg.hidden_channels, cfg.num_layers, cfg.num_rbf, cfg.num_elements = F, L, R, rep.max_z - 1
            cfg.max_neighbors, cfg.envelope_exponent = 2 ** 30, 5      # ASE-style list: every pair inside the cutoff
            cfg.cutoff = float(rep.cutoff)
            cfg.rbf_coeff, cfg.filter_mode = -0.5 / width ** 2, 1
        else:
            cfg = _lib.SchnetCfg()
            cfg.n_atom_basis, cfg.n_interactions, cfg.n_rbf, cfg.max_z = F, L, R, rep.max_z
            cfg.cutoff, cfg.rbf_coeff = float(rep.cutoff), -0.5 / width ** 2
        self._cfg = cfg
        self._index = None
        self._n_spk = 0

    @property
    def _atomwise(self):
        return self.output_modules[self._i_atomwise]

    @property
    def _forces(self):
        return None if self._i_forces is None else self.output_modules[self._i_forces]

    def _entry_points(self, lib):
        if self._kind == "painn":
            return lib.nq_painn_workspace_bytes, lib.nq_painn_forward, lib.nq_painn_backward
        return lib.nq_schnet_workspace_bytes, lib.nq_schnet_forward, lib.nq_schnet_backward

    def _prepare(self, device):
        """Gather index spk layout -> engine layout (PaiNN: a permutation; SchNet: the engine uses the module order itself)."""
        rep = self.representation
        if self._kind == "painn":
            if self._index is None or self._index.device != device:
                self._index, self._n_spk = _spk_index(rep.n_atom_basis, rep.n_interactions, rep.radial_basis.n_rbf, rep.max_z, device)
        else:
            self._index, self._n_spk = None, sum(p.numel() for p in self._engine_params())

    def _engine_params(self):
        rep, aw = self.representation, self._atomwise
        if self._kind == "schnet":
            ps = [rep.embedding.weight]
            for it in rep.interactions:
                ps += [it.in2f.weight, it.filter_network[0].weight, it.filter_network[0].bias, it.filter_network[1].weight, it.filter_network[1].bias,
                       it.f2out[0].weight, it.f2out[0].bias, it.f2out[1].weight, it.f2out[1].bias]
            return ps + [aw.outnet[0].weight, aw.outnet[0].bias, aw.outnet[1].weight, aw.outnet[1].bias]
        ps = [rep.embedding.weight, rep.filter_net.weight, rep.filter_net.bias]
        for it in rep.interactions:
            n = it.interatomic_context_net
            ps += [n[0].weight, n[0].bias, n[1].weight, n[1].bias]
        for mx in rep.mixing:
            n = mx.intraatomic_context_net
            ps += [n[0].weight, n[0].bias, n[1].weight, n[1].bias, mx.mu_channel_mix.weight]
        ps += [aw.outnet[0].weight, aw.outnet[0].bias, aw.outnet[1].weight, aw.outnet[1].bias]
        return ps

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_index"] = None
        return state

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        R_, Z, idx_m = inputs[properties_R], inputs[properties_Z], inputs[properties_idx_m]
        if not R_.is_cuda:
            raise RuntimeError("nabladft_amd.spk runs on MI355X only (no CPU fallback): move the batch to cuda")
        rep = self.representation
        self._prepare(R_.device)
        nl = build_neighbor_list(R_, idx_m.long(), Z.long(), rep.cutoff, 2 ** 30)
        if nl.E == 0:
            raise IndexError("batch has no atom pair within the cutoff")
        want_forces = self._forces is not None and self._forces.calc_forces
        energy, forces = _SpkEnergyForces.apply(self, nl, want_forces, *self._engine_params())
        inputs[self._atomwise.output_key] = energy
        if want_forces:
            inputs[self._forces.force_key] = forces
        if self.do_postprocessing and not self.training:
            for pp in self.postprocessors:
                inputs = pp(inputs)
        return {k: inputs[k] for k in self.model_outputs}
