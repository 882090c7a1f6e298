"""Minimal ``_target_`` instantiator (the subset of hydra.utils.instantiate the nablaDFT model configs use:
``_target_``, ``_partial_``, nested dicts/lists; ``_convert_`` is accepted and ignored).  hydra-core is not a
dependency of this package; when it is installed, ``hydra.utils.instantiate`` works on the same dicts."""
import functools
import importlib
from typing import Any


def _locate(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def locate(path: str):
    """``{_target_: nabladft_amd.config.locate, path: torch.optim.AdamW}`` -> the class object itself (what hydra's ``_target_: hydra.utils.get_class`` does for the
    ``optimizer_cls`` / ``scheduler_cls`` arguments of the schnetpack task, config/model/painn.yaml:48-56)."""
    return _locate(path)


def instantiate(cfg: Any, **overrides):
    if isinstance(cfg, (list, tuple)):
        return type(cfg)(instantiate(c) for c in cfg)
    if not isinstance(cfg, dict):
        return cfg
    if "_target_" not in cfg:
        return {k: instantiate(v) for k, v in cfg.items()}
    kwargs = {k: instantiate(v) for k, v in cfg.items() if k not in ("_target_", "_partial_", "_convert_")}
    kwargs.update(overrides)
    fn = _locate(cfg["_target_"])
    if cfg.get("_partial_", False):
        return functools.partial(fn, **kwargs)
    return fn(**kwargs)
