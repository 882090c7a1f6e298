"""Exponential moving average of a parameter set with the interface QHNetLightning uses (qhnet.py:459-460, :480-482, :521-536; the
reference's config instantiates ``torch_ema.ExponentialMovingAverage(parameters, decay=0.9999)``, config/model/qhnet.yaml:51-54 -- torch_ema is a
third-party wheel, restated here from its published behaviour):

    ema = ExponentialMovingAverage(model.parameters(), decay=0.9999)
    ema.update()                          # after every optimiser step: shadow -= (1 - d_t) * (shadow - param),
                                          # d_t = min(decay, (1 + n) / (10 + n)) while use_num_updates
    with ema.average_parameters(): ...    # parameters temporarily replaced by the averages
    ema.store() / ema.copy_to() / ema.restore(), ema.to(device), state_dict() / load_state_dict()

The shadow lives in ONE flat fp32 buffer: an update is a single fused ``torch._foreach``-free lerp over views of it.
"""
import contextlib
from typing import Iterable, Optional

import torch


class ExponentialMovingAverage:
    def __init__(self, parameters: Iterable[torch.nn.Parameter], decay: float, use_num_updates: bool = True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self._params = [p for p in parameters if p.requires_grad]
        n = sum(p.numel() for p in self._params)
        dev = self._params[0].device if self._params else torch.device("cpu")
        self._flat = torch.empty(n, device=dev, dtype=torch.float32)
        self._views, o = [], 0
        with torch.no_grad():
            for p in self._params:
                v = self._flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.detach())
                self._views.append(v)
                o += p.numel()
        self._stored = None

    @property
    def shadow_params(self):
        return self._views

    def _resolve(self, parameters):
        """The parameter list an operation acts on: the tracked one, or a caller-supplied list of the SAME length (torch_ema raises on a mismatch;
        zip() would silently truncate)."""
        if parameters is None:
            return self._params
        params = [p for p in parameters if p.requires_grad]
        if len(params) != len(self._views):
            raise ValueError(f"Number of parameters passed as argument ({len(params)}) is different from number of shadow parameters maintained by this "
                             f"ExponentialMovingAverage ({len(self._views)})")
        return params

    def _rebind(self):
        o = 0
        self._views = []
        for p in self._params:
            self._views.append(self._flat[o:o + p.numel()].view(p.shape))
            o += p.numel()

    def update(self, parameters: Optional[Iterable[torch.nn.Parameter]] = None) -> None:
        params = self._resolve(parameters)
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            if params and all(p.is_cuda for p in params) and hasattr(torch, "_foreach_lerp_"):
                torch._foreach_lerp_(self._views, [p.detach().to(torch.float32) for p in params], 1.0 - decay)
            else:
                for s, p in zip(self._views, params):
                    s.sub_((1.0 - decay) * (s - p.detach()))

    def copy_to(self, parameters: Optional[Iterable[torch.nn.Parameter]] = None) -> None:
        params = self._resolve(parameters)
        with torch.no_grad():
            for s, p in zip(self._views, params):
                p.data.copy_(s)

    def store(self, parameters: Optional[Iterable[torch.nn.Parameter]] = None) -> None:
        params = self._resolve(parameters)
        self._stored = [p.detach().clone() for p in params]

    def restore(self, parameters: Optional[Iterable[torch.nn.Parameter]] = None) -> None:
        if self._stored is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        params = self._resolve(parameters)
        with torch.no_grad():
            for c, p in zip(self._stored, params):
                p.data.copy_(c)

    @contextlib.contextmanager
    def average_parameters(self, parameters: Optional[Iterable[torch.nn.Parameter]] = None):
        params = None if parameters is None else list(parameters)
        self.store(params)
        self.copy_to(params)
        try:
            yield
        finally:
            self.restore(params)
            self._stored = None

    def to(self, device=None, dtype=None) -> None:
        """Moves the shadow buffer AND any store()d copies (restore() after a device move must not copy across devices).  The shadow is kept in fp32
        whatever the parameters' dtype (an average of low-precision weights loses its small increments): a `dtype` other than float32 is refused."""
        if dtype is not None and dtype != torch.float32:
            raise ValueError("the EMA shadow is kept in float32")
        self._flat = self._flat.to(device=device)
        self._rebind()
        if self._stored is not None:
            self._stored = [c.to(device=device) for c in self._stored]

    def state_dict(self) -> dict:
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": [v.clone() for v in self._views], "collected_params": self._stored}

    def load_state_dict(self, state_dict: dict) -> None:
        self.decay, self.num_updates = state_dict["decay"], state_dict["num_updates"]
        with torch.no_grad():
            for v, s in zip(self._views, state_dict["shadow_params"]):
                v.copy_(s)
        stored = state_dict.get("collected_params")
        self._stored = None if stored is None else [c.to(device=self._flat.device) for c in stored]
