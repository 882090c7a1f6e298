"""TEST INFRASTRUCTURE (container-only) -- imports the REAL reference GemNet-OC (/root/reference/nablaDFT/gemnet_oc/*.py, an OCP port) on CPU.

The reference files are pure torch + numpy + sympy + scipy except for four wheels that are not installed here; minimal stand-ins
for exactly the symbols the GemNet-OC path touches are registered before the import:
  * ``torch_scatter.scatter / segment_coo / segment_csr`` and ``torch_geometric.nn.radius_graph`` -- the documented-semantics stand-ins of oracle/ref_import.py
    (the ones the PaiNN fixtures use); ``scatter`` additionally with reduce="mean" (forces_coupled, gemnet_oc.py:1216-1224);
  * ``torch_sparse.SparseTensor`` -- third-party (torch-sparse 0.6.18, setup.py:37-39), restated from its published behaviour [memory]: COO entries sorted by
    (row, col) at construction; ``adj[idx]`` = index_select of rows (output row r' = position in idx, entries of row idx[r'] in column order);
    ``storage.row() / col() / value()``, ``coo()``, ``set_value_``, ``sparse_sizes()``.  PARITY UNPINNED UPSTREAM for this piece, like torch_cluster;
  * ``pytorch_lightning.LightningModule``, ``torch_geometric.data.Data`` -- empty shells.
Used only by oracle/make_golden_gemnet.py; nothing on the GPU box imports this file.
"""
import importlib
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, is_sorted=False):
        n_rows, n_cols = sparse_sizes
        if not is_sorted:
            perm = torch.argsort(row * n_cols + col, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.storage = _Storage(row, col, value)
        self._sizes = (int(n_rows), int(n_cols))

    def sparse_sizes(self):
        return self._sizes

    def coo(self):
        return self.storage.row(), self.storage.col(), self.storage.value()

    def set_value_(self, value, layout="coo"):
        self.storage._value = value
        return self

    def __getitem__(self, idx):
        assert torch.is_tensor(idx) and idx.dtype == torch.long and idx.dim() == 1
        row, col, value = self.coo()
        n_rows = self._sizes[0]
        cnt = torch.bincount(row, minlength=n_rows)
        rowptr = torch.cat([cnt.new_zeros(1), cnt.cumsum(0)])
        sel = cnt[idx]
        out_row = torch.repeat_interleave(torch.arange(idx.numel()), sel)
        start = rowptr[idx]
        offs = torch.arange(int(sel.sum())) - torch.repeat_interleave(sel.cumsum(0) - sel, sel)
        src = torch.repeat_interleave(start, sel) + offs
        return SparseTensor(row=out_row, col=col[src], value=None if value is None else value[src], sparse_sizes=(idx.numel(), self._sizes[1]), is_sorted=True)


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass


_loaded = {}


def load_gemnet():
    if _loaded:
        return _loaded
    from oracle import ref_import

    def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
        if reduce in ("sum", "add"):
            return ref_import.scatter(src, index, dim, out, dim_size, "sum")
        assert reduce == "mean"
        tot = ref_import.scatter(src, index, dim, out, dim_size, "sum")
        cnt = ref_import.scatter(torch.ones_like(src), index, dim, None, dim_size, "sum").clamp(min=1)
        return tot / cnt

    _mod("torch_scatter", scatter=scatter, segment_coo=ref_import.segment_coo, segment_csr=ref_import.segment_csr)
    _mod("torch_sparse", SparseTensor=SparseTensor)
    tg = sys.modules.get("torch_geometric") or _mod("torch_geometric")
    _mod("torch_geometric.nn", radius_graph=ref_import.radius_graph, MessagePassing=ref_import.MessagePassing)
    _mod("torch_geometric.data", Data=object)
    _mod("pytorch_lightning", LightningModule=_LightningModule)
    pkg = _mod("nablaDFT")
    pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT"]
    gpkg = _mod("nablaDFT.gemnet_oc")                      # fake parent: gemnet_oc/__init__.py is not executed
    gpkg.__path__ = [REFERENCE_ROOT + "/nablaDFT/gemnet_oc"]
    _loaded["gemnet"] = importlib.import_module("nablaDFT.gemnet_oc.gemnet_oc")
    _loaded["utils"] = importlib.import_module("nablaDFT.gemnet_oc.utils")
    _loaded["indices"] = importlib.import_module("nablaDFT.gemnet_oc.interaction_indices")
    _loaded["loss"] = importlib.import_module("nablaDFT.gemnet_oc.loss")
    return _loaded
