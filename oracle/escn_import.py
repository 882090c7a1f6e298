"""TEST INFRASTRUCTURE (container-only) -- imports the REAL reference eSCN (/root/reference/nablaDFT/escn/escn.py, so3.py, sampling.py, smearing.py) on CPU.
The model needs e3nn for five symbols (escn/so3.py:13-14,380-381,453-470; escn.py:170-176): ``o3.xyz_to_angles``, ``o3.angles_to_matrix``, ``ToS2Grid``,
``FromS2Grid``, ``o3.spherical_harmonics``; e3nn is not installed anywhere, so oracle/e3nn_mini.py (restated, PARITY UNPINNED) is registered as ``e3nn``.
The Wigner-D recursion data ``escn/Jd.pt`` is read from the reference tree by the reference's own ``torch.load``.  Wheels: the stand-ins of
oracle/gemnet_import.py (radius_graph, torch_scatter) -- eSCN imports nablaDFT.gemnet_oc.utils for compute_neighbors."""
import importlib
import sys

from oracle import e3nn_mini
from oracle.gemnet_import import REFERENCE_ROOT, _mod, load_gemnet

_loaded = {}


def load_escn():
    if _loaded:
        return _loaded
    e3nn_mini.install()
    load_gemnet()                                             # registers torch_scatter / torch_geometric / pytorch_lightning stand-ins and nablaDFT.gemnet_oc.*
    pkg = _mod("nablaDFT.escn")                              # fake parent: escn/__init__.py is not executed
    pkg.__path__ = [REFERENCE_ROOT + "/nablaDFT/escn"]
    _loaded["escn"] = importlib.import_module("nablaDFT.escn.escn")
    _loaded["so3"] = importlib.import_module("nablaDFT.escn.so3")
    return _loaded
