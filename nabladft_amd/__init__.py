"""nabladft_amd -- MI355X-native (gfx950, hand-written HIP) engine for the nablaDFT PaiNN hot path.

Public surface mirrors the reference plugin classes for this path:
  nabladft_amd.PaiNN            <-> nablaDFT.painn_pyg.PaiNN            (painn.py:22-148)
  nabladft_amd.PaiNNLightning   <-> nablaDFT.painn_pyg.PaiNNLightning   (painn.py:623-776)
  nabladft_amd.L2Loss           <-> nablaDFT.gemnet_oc.loss.L2Loss      (gemnet_oc/loss.py:15-22)
  nabladft_amd.QHNet            <-> nablaDFT.qhnet.QHNet                (qhnet/qhnet.py:24-342)
  nabladft_amd.QHNetLightning   <-> nablaDFT.qhnet.QHNetLightning       (qhnet/qhnet.py:345-536)
  nabladft_amd.GemNetOC         <-> nablaDFT.gemnet_oc.GemNetOC         (gemnet_oc/gemnet_oc.py:36-1340)
  nabladft_amd.GemNetOCLightning <-> nablaDFT.gemnet_oc.GemNetOCLightning (gemnet_oc/gemnet_oc.py:1343-1493)
  nabladft_amd.eSCN             <-> nablaDFT.escn.eSCN                  (escn/escn.py:36-490)
  nabladft_amd.eSCNLightning    <-> nablaDFT.escn.eSCNLightning         (escn/escn.py:1006-1159)
  nabladft_amd.EquiformerV2_OC20 <-> nablaDFT.equiformer_v2.EquiformerV2_OC20 (equiformer_v2/equiformer_v2_oc20.py:51-640)
  nabladft_amd.EquiformerV2_OC20_Lightning <-> nablaDFT.equiformer_v2.EquiformerV2_OC20_Lightning (equiformer_v2/equiformer_v2_oc20.py:643-817)
  nabladft_amd.AtomisticTaskFixed <-> nablaDFT.ase_model.AtomisticTaskFixed (ase_model/task.py:9-73)
  nabladft_amd.spk.*            <-> the schnetpack classes config/model/{painn,schnet}.yaml instantiate (parity unpinned)
  nabladft_amd.read_energy_database / ArenaLoader <-> PyGNablaDFT.process + DataLoader collate (dataset/pyg_datasets.py:101-109)
"""
from .painn import PaiNN, NeighborList, build_neighbor_list  # noqa: F401
from .lightning import AtomisticTaskFixed, EquiformerV2_OC20_Lightning, GemNetOCLightning, eSCNLightning, L2Loss, ModelOutput, PaiNNLightning, QHNetLightning  # noqa: F401
from .qhnet import QHNet  # noqa: F401
from .gemnet_oc import GemNetOC  # noqa: F401
from .escn import eSCN  # noqa: F401
from .equiformer_v2 import EquiformerV2_OC20  # noqa: F401
from . import ema  # noqa: F401
from .trainer import FusedTrainStep, Batch  # noqa: F401
from .data import ArenaLoader, ConformerArena, HamiltonianBatch, HamiltonianDatabase, HamiltonianDataset, hamiltonian_batch, read_energy_database  # noqa: F401

__all__ = ["PaiNN", "PaiNNLightning", "QHNet", "QHNetLightning", "GemNetOC", "GemNetOCLightning", "eSCN", "eSCNLightning", "EquiformerV2_OC20", "EquiformerV2_OC20_Lightning", "AtomisticTaskFixed", "ModelOutput", "L2Loss", "FusedTrainStep", "Batch", "build_neighbor_list", "NeighborList", "ArenaLoader", "ConformerArena",
           "read_energy_database", "HamiltonianDatabase", "HamiltonianDataset", "HamiltonianBatch", "hamiltonian_batch"]
