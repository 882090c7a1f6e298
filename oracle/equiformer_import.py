"""TEST INFRASTRUCTURE (container-only) -- imports the REAL reference EquiformerV2 (/root/reference/nablaDFT/equiformer_v2/*.py) on CPU.
Third-party symbols the model needs and this image lacks:
  * e3nn: ``o3.xyz_to_angles``, ``o3.angles_to_matrix``, ``ToS2Grid``, ``FromS2Grid`` (equiformer_v2/so3.py:13-16,338-340,374-401) -> oracle/e3nn_mini.py
    (restated, PARITY UNPINNED); ``equiformer_v2/Jd.pt`` is read by the reference's own ``torch.load``;
  * ``torch_geometric.utils.softmax(src, index)`` (transformer_block.py:352): softmax over the entries that share an index value [documented semantics];
  * ``torch_geometric.nn.radius_graph``, torch_scatter, pytorch_lightning: the stand-ins of oracle/gemnet_import.py (EquiformerV2 imports
    nablaDFT.gemnet_oc.utils for compute_neighbors)."""
import importlib
import sys

import torch

from oracle import e3nn_mini
from oracle.gemnet_import import REFERENCE_ROOT, _mod, load_gemnet

_loaded = {}


def segment_softmax(src, index, ptr=None, num_nodes=None, dim=0):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.full(shape, -float("inf"), dtype=src.dtype).scatter_reduce(0, idx, src.detach(), "amax", include_self=True)
    e = (src - mx[index]).exp()
    den = torch.zeros(shape, dtype=src.dtype).index_add(0, index, e)
    return e / (den[index] + 1e-16)


def load_equiformer():
    if _loaded:
        return _loaded
    e3nn_mini.install()
    load_gemnet()
    tg = sys.modules["torch_geometric"]
    tg.utils = _mod("torch_geometric.utils", softmax=segment_softmax)
    pkg = _mod("nablaDFT.equiformer_v2")                     # fake parent: equiformer_v2/__init__.py is not executed
    pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT/equiformer_v2"]
    _loaded["model"] = importlib.import_module("nablaDFT.equiformer_v2.equiformer_v2_oc20")
    _loaded["so3"] = importlib.import_module("nablaDFT.equiformer_v2.so3")
    _loaded["blocks"] = importlib.import_module("nablaDFT.equiformer_v2.transformer_block")
    return _loaded
