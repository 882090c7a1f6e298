"""TEST INFRASTRUCTURE (container-only) -- runs the REAL reference QHNet (/root/reference/nablaDFT/qhnet/{qhnet,layers,loss}.py) on CPU.

The reference files import e3nn, torch_cluster, torch_scatter, torch_geometric and pytorch_lightning at module level; none is installed here
(SURVEY.md section 0.3).  This loader registers
  * ``e3nn``            -> oracle/e3nn_mini.py, this repo's restatement of e3nn 0.5.1 (third-party, PARITY UNPINNED: see its header);
  * ``torch_cluster.radius_graph`` / ``torch_scatter.scatter`` -> the documented-semantics stand-ins of oracle/ref_import.py (the same
    ones the PaiNN fixtures use: strict d^2 < r^2, first K neighbours by ascending index, sum-scatter);
  * ``torch_geometric.data.Data``, ``pytorch_lightning.LightningModule`` -> empty shells (never exercised by the network itself)
and then imports the reference modules unchanged.  Every QHNet-specific line -- get_feasible_irrep incl. the shadowed-variable normalisation
(layers.py:60-82), the dst||dst invariants of ConvLayer (layers.py:240-258), PairNet/SelfNet wiring, Expansion (layers.py:598-662),
build_graph / build_final_matrix (qhnet.py:254-321) -- is therefore the reference's own code; what stays unpinned is e3nn's arithmetic.

Used only by oracle/make_golden_qhnet_model.py; nothing on the GPU box imports this file.
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


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass


_loaded = {}


def load_qhnet_full():
    if _loaded:
        return _loaded
    from oracle import e3nn_mini
    from oracle import ref_import          # registers torch_scatter / torch_geometric stand-ins as a side effect of its helpers
    e3nn_mini.install()
    _mod("torch_cluster", radius_graph=ref_import.radius_graph)
    _mod("torch_scatter", scatter=ref_import.scatter, segment_coo=ref_import.segment_coo, segment_csr=ref_import.segment_csr)
    if "torch_geometric" not in sys.modules:
        _mod("torch_geometric")
    _mod("torch_geometric.data", Data=object)
    _mod("pytorch_lightning", LightningModule=_LightningModule)
    pkg = _mod("nablaDFT")
    pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT"]
    qpkg = _mod("nablaDFT.qhnet")                      # fake parent: qhnet/__init__.py (imports torchmetrics) is not executed
    qpkg.__path__ = [REFERENCE_ROOT + "/nablaDFT/qhnet"]
    for name in ("nablaDFT.qhnet.layers", "nablaDFT.qhnet.qhnet", "nablaDFT.qhnet.loss"):
        sys.modules.pop(name, None)
    _loaded["layers"] = importlib.import_module("nablaDFT.qhnet.layers")
    _loaded["qhnet"] = importlib.import_module("nablaDFT.qhnet.qhnet")
    _loaded["loss"] = importlib.import_module("nablaDFT.qhnet.loss")
    return _loaded


class Data:
    """Minimal PyG-batch shaped object: attributes pos, z, batch, ptr, num_nodes (+ whatever QHNet.forward writes onto it)."""

    def __init__(self, pos, z, batch, ptr):
        self.pos, self.z, self.batch, self.ptr = pos, z, batch, ptr
        self.num_nodes = int(pos.shape[0])
