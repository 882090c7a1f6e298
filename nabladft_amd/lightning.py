"""``PaiNNLightning`` with the constructor and step methods of the reference wrapper
(/root/reference/nablaDFT/painn_pyg/painn.py:623-776) and ``L2Loss`` (gemnet_oc/loss.py:5-22).

pytorch_lightning is used when importable (then ``run.py`` / ``pipelines.py`` drive this class
unchanged: config/model/painn-oc.yaml only needs its two ``_target_`` lines pointed here, see
INTEGRATION.md); without it the class degrades to a plain nn.Module with the same methods so the
step logic stays testable.
"""
from types import SimpleNamespace
from typing import Dict

import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    _HAVE_PL = True
except Exception:  # noqa: BLE001
    _Base = nn.Module
    _HAVE_PL = False


def l2loss_atomwise(pred, target, reduction="mean"):
    dist = torch.linalg.vector_norm((pred - target), dim=-1)
    if reduction == "mean":
        return torch.mean(dist)
    if reduction == "sum":
        return torch.sum(dist)
    return dist


class L2Loss(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, pred, target):
        return l2loss_atomwise(pred, target, self.reduction)


class PaiNNLightning(_Base):
    def __init__(self, model_name: str, model: nn.Module, optimizer, lr_scheduler, losses: Dict, metric, loss_coefs) -> None:
        super().__init__()
        self.model = model
        if _HAVE_PL:
            self.save_hyperparameters(logger=True, ignore=["net"])
        else:
            self.hparams = SimpleNamespace(model_name=model_name, optimizer=optimizer, lr_scheduler=lr_scheduler, losses=losses,
                                           metric=metric, loss_coefs=loss_coefs)

    def forward(self, data):
        energy, forces = self.model(data)
        return energy, forces

    def step(self, batch, calculate_metrics: bool = False):
        y = batch.y
        energy_out, forces_out = self.model(batch)
        forces = batch.forces
        preds = {"energy": energy_out, "forces": forces_out}
        target = {"energy": y, "forces": forces}
        loss = self._calculate_loss(preds, target)
        if calculate_metrics:
            metrics = self._calculate_metrics(preds, target)
            return loss, metrics
        return loss

    def training_step(self, batch, batch_idx):
        bsz = self._get_batch_size(batch)
        loss = self.step(batch, calculate_metrics=False)
        self._log("train/loss", loss, on_step=True, on_epoch=True, logger=True, sync_dist=True, batch_size=bsz)
        return loss

    def validation_step(self, batch, batch_idx):
        bsz = self._get_batch_size(batch)
        loss, _ = self.step(batch, calculate_metrics=True)
        self._log("val/loss", loss, prog_bar=True, on_step=True, on_epoch=True, logger=True, sync_dist=True, batch_size=bsz)
        self._log("val_loss", loss, on_step=False, on_epoch=True, logger=False, sync_dist=True, batch_size=bsz)
        return loss

    def test_step(self, batch, batch_idx):
        bsz = self._get_batch_size(batch)
        loss, _ = self.step(batch, calculate_metrics=True)
        self._log("test/loss", loss, prog_bar=True, on_step=True, on_epoch=True, logger=True, sync_dist=True, batch_size=bsz)
        return loss

    def predict_step(self, data, **kwargs):
        return self(data)

    def configure_optimizers(self):
        optimizer = self.hparams.optimizer(params=self.parameters())
        if self.hparams.lr_scheduler is not None:
            scheduler = self.hparams.lr_scheduler(optimizer=optimizer)
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": scheduler, "interval": "epoch", "monitor": "val_loss", "frequency": 1}}
        return {"optimizer": optimizer}

    def on_fit_start(self) -> None:
        self._check_devices()

    def on_test_start(self) -> None:
        self._check_devices()

    def on_validation_epoch_end(self) -> None:
        self._reduce_metrics(step_type="val")

    def on_test_epoch_end(self) -> None:
        self._reduce_metrics(step_type="test")

    def _calculate_loss(self, y_pred, y_true):
        total_loss = 0.0
        for name, loss in self.hparams.losses.items():
            total_loss += self.hparams.loss_coefs[name] * loss(y_pred[name], y_true[name])
        return total_loss

    def _calculate_metrics(self, y_pred, y_true):
        if self.hparams.metric is None:
            return {}
        return self.hparams.metric(y_pred, y_true)

    def _reduce_metrics(self, step_type: str = "train"):
        if self.hparams.metric is None:
            return
        metric = self.hparams.metric.compute()
        for key in metric.keys():
            self._log(f"{step_type}/{key}", metric[key], logger=True, on_step=False, on_epoch=True, sync_dist=True)
        self.hparams.metric.reset()

    def _check_devices(self):
        if self.hparams.metric is not None:
            self.hparams.metric = self.hparams.metric.to(next(self.parameters()).device)

    def _get_batch_size(self, batch):
        return batch.batch.max().detach().item() + 1

    def _log(self, *args, **kwargs):
        if _HAVE_PL:
            self.log(*args, **kwargs)
