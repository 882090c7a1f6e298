"""TEST INFRASTRUCTURE (container-only) -- imports the *real* reference QHNet class for the pieces of the Hamiltonian path that are
plain torch: ``QHNet._get_mask`` (qhnet.py:323-342), ``QHNet.build_final_matrix`` (qhnet.py:293-321), the transpose index of
``QHNet.build_graph`` (qhnet.py:273-283) and ``HamiltonianLoss`` (qhnet/loss.py:5-16).

qhnet.py / layers.py import e3nn, torch_cluster, torch_geometric, torch_scatter and pytorch_lightning at module level; none is
installed here.  The methods used below never touch them, so empty stand-in modules that only satisfy the import statements are
registered (any call into them raises).  Used only by oracle/make_golden_qhnet.py; nothing on the GPU box imports this file.
"""
import importlib
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


class _Missing:
    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self._name} is a stand-in: the dependency is not installed in this container")

    def __getattr__(self, item):
        return _Missing(self._name + "." + item)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass


_loaded = {}


def load_qhnet():
    if _loaded:
        return _loaded
    if "e3nn" not in sys.modules:
        e3 = _mod("e3nn", o3=_Missing("e3nn.o3"))
        _mod("e3nn.o3", Linear=_Missing("e3nn.o3.Linear"), TensorProduct=_Missing("e3nn.o3.TensorProduct"))
        _mod("e3nn.nn", FullyConnectedNet=_Missing("e3nn.nn.FullyConnectedNet"))
        e3.o3 = sys.modules["e3nn.o3"]
    if "torch_cluster" not in sys.modules:
        _mod("torch_cluster", radius_graph=_Missing("torch_cluster.radius_graph"))
    if "torch_scatter" not in sys.modules:
        _mod("torch_scatter", scatter=_Missing("torch_scatter.scatter"))
    if "torch_geometric" not in sys.modules:
        _mod("torch_geometric")
    if "torch_geometric.data" not in sys.modules:
        _mod("torch_geometric.data", Data=object)
    if "pytorch_lightning" not in sys.modules:
        _mod("pytorch_lightning", LightningModule=_LightningModule)
    if "nablaDFT" not in sys.modules:
        pkg = _mod("nablaDFT")
        pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT"]
    qpkg = _mod("nablaDFT.qhnet")                      # fake parent: qhnet/__init__.py (imports torchmetrics) is not executed
    qpkg.__path__ = [REFERENCE_ROOT + "/nablaDFT/qhnet"]
    _loaded["qhnet"] = importlib.import_module("nablaDFT.qhnet.qhnet")
    _loaded["loss"] = importlib.import_module("nablaDFT.qhnet.loss")
    return _loaded
