"""TEST INFRASTRUCTURE (container-only) -- imports the *real* reference PaiNN.

This file never travels as a dependency of the product: it is used by
``oracle/make_golden.py`` in the build container (where ``/root/reference`` is
mounted) to produce the golden vectors under ``tests/golden/``.  Nothing on the
GPU box imports it.

The reference module ``/root/reference/nablaDFT/painn_pyg/painn.py`` is pure
Python but depends on wheels that are not installed here (torch_geometric,
torch_scatter, pytorch_lightning).  We register minimal stand-ins for exactly
the symbols the PaiNN path touches (painn.py:4-19, layers.py:11, utils.py:9):

* ``torch_scatter.scatter / segment_coo / segment_csr``  (sum-reduce only)
* ``torch_geometric.nn.radius_graph``  -- documented torch_cluster 1.6.3 CUDA
  semantics: strict ``d^2 < r^2``, no self loops, per centre the first K
  neighbours in ascending source index, row0 = source j, row1 = target i,
  centres ascending.
* ``torch_geometric.nn.MessagePassing`` (aggr='add', node_dim=0) and
  ``torch_geometric.nn.models.schnet.GaussianSmearing``
* ``pytorch_lightning.LightningModule`` (plain nn.Module)

torch_cluster / torch_scatter are the un-vendored third-party dependencies of
this path (setup.py:37-39); no reference test pins their numeric behaviour, so
their semantics are pinned by the stand-ins below ("parity unpinned upstream").
"""
import importlib
import inspect
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert reduce in ("sum", "add")
    dim = dim % src.dim()
    n = int(dim_size) if dim_size is not None else int(index.max()) + 1
    shape = list(src.shape)
    shape[dim] = n
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add(dim, idx, src)


def segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    return scatter(src, index, 0, None, dim_size, reduce)


def segment_csr(src, indptr, out=None, reduce="sum"):
    n = indptr.numel() - 1
    seg = torch.repeat_interleave(torch.arange(n, device=src.device), indptr[1:] - indptr[:-1])
    return scatter(src, seg, 0, None, n, reduce)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    n = x.size(0)
    if batch is None:
        batch = x.new_zeros(n, dtype=torch.long)
    d2 = (x[:, None, :] - x[None, :, :]).pow(2).sum(-1)
    adj = (d2 < r * r) & (batch[:, None] == batch[None, :])
    if not loop:
        adj &= ~torch.eye(n, dtype=torch.bool, device=x.device)
    centre, nbr = adj.nonzero(as_tuple=True)  # centre-major, neighbours ascending
    rank = torch.cumsum(adj.long(), dim=1)[centre, nbr] - 1
    keep = rank < max_num_neighbors
    centre, nbr = centre[keep], nbr[keep]
    return torch.stack([nbr, centre])


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", node_dim=-2, flow="source_to_target"):
        super().__init__()
        self.node_dim = node_dim

    def jittable(self, *a, **k):
        return self

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = edge_index
        args, n = {}, None
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_j"):
                t = kwargs[name[:-2]]
                n = t.size(self.node_dim)
                args[name] = t.index_select(self.node_dim, j)
            elif name.endswith("_i"):
                t = kwargs[name[:-2]]
                n = t.size(self.node_dim)
                args[name] = t.index_select(self.node_dim, i)
            else:
                args[name] = kwargs[name]
        return self.update(self.aggregate(self.message(**args), i, None, n))


class GaussianSmearing(torch.nn.Module):
    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, dist):
        dist = dist.view(-1, 1) - self.offset.view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))


class LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass


_loaded = {}


def load_reference():
    """Returns the reference ``nablaDFT.painn_pyg.painn`` module (and loss module)."""
    if _loaded:
        return _loaded
    _mod("torch_scatter", scatter=scatter, segment_coo=segment_coo, segment_csr=segment_csr)
    _mod("torch_geometric")
    _mod("torch_geometric.nn", MessagePassing=MessagePassing, radius_graph=radius_graph)
    _mod("torch_geometric.nn.models")
    _mod("torch_geometric.nn.models.schnet", GaussianSmearing=GaussianSmearing)
    _mod("pytorch_lightning", LightningModule=LightningModule)
    # fake parent packages so nablaDFT/__init__.py (imports every model) is not executed
    pkg = _mod("nablaDFT")
    pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT"]
    gem = _mod("nablaDFT.gemnet_oc")
    gem.__path__ = [REFERENCE_ROOT + "/nablaDFT/gemnet_oc"]
    _loaded["painn"] = importlib.import_module("nablaDFT.painn_pyg.painn")
    _loaded["layers"] = importlib.import_module("nablaDFT.painn_pyg.layers")
    _loaded["loss"] = importlib.import_module("nablaDFT.gemnet_oc.loss")
    return _loaded


class Data:
    """Minimal PyG-batch-shaped object (pos, z, batch, ptr, y, forces)."""

    def __init__(self, pos, z, batch, y=None, forces=None):
        self.pos, self.z, self.batch, self.y, self.forces = pos, z, batch, y, forces
        counts = torch.bincount(batch)
        self.ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        self.num_nodes = pos.shape[0]


if __name__ == "__main__":
    # known-answer check from SURVEY.md Appendix E
    ref = load_reference()
    torch.manual_seed(0)
    m = ref["painn"].PaiNN(128, 6, 100, 5.0, 100, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5},
                           True, False, False, True, 100)
    g = torch.Generator().manual_seed(1)
    pos = torch.rand(160, 3, generator=g) * 6
    z = torch.tensor([1, 6, 7, 8, 9, 16, 17, 35])[torch.randint(0, 8, (160,), generator=g)]
    batch = torch.arange(4).repeat_interleave(40)
    e, f = m(Data(pos, z, batch))
    print(sum(p.numel() for p in m.parameters() if p.requires_grad), e.detach(), f.abs().mean().item())
