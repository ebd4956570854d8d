"""Loss weighting and step logging around the bottleneck.

Mirrors the consumer side of the path in the reference:
  * `Funnel_T5_VAE_Model._regulariser_loss_weight_schedule` (model.py:103-109):
    w = sigmoid(step * k - b) as a python float, 0 when disabled or before the first step;
  * `loss = decoder_ce + reg_loss * w` (model.py:305-306) — `w` reaches the fused MMD gradient as
    autograd's `grad_output`, so nothing is recomputed;
  * `_update_logs / get_latest_logs` (model.py:111-134, 308-311).  The reference calls
    `reg_loss.item()` and `decoder_ce.item()` every training step, a device->host sync per step;
    `DeferredLogs` keeps the running sums ON the device and reads them back once, when the logs
    are fetched, with the same normalisation (increase since the last fetch / calls since then).
"""
from __future__ import annotations

import math
from typing import Dict, Union

import torch

Number = Union[float, int, torch.Tensor]


def regulariser_loss_weight(global_step, reg_schedule_k: float = 0.0025, reg_schedule_b: float = 6.25,
                            use_reg_loss: bool = True) -> float:
    """model.py:103-109 (defaults config.py:83-84).  Pure host arithmetic, no tensor, no sync."""
    if global_step is None or not use_reg_loss:
        return 0
    x = global_step * reg_schedule_k - reg_schedule_b
    # the reference evaluates torch.sigmoid on a float32 tensor; float32 rounding of x is reproduced
    x = float(torch.tensor(x, dtype=torch.float32))
    return 1.0 / (1.0 + math.exp(-x))


def combine_loss(decoder_ce: torch.Tensor, reg_loss: torch.Tensor, reg_loss_w: float) -> torch.Tensor:
    """model.py:305-306."""
    return decoder_ce + reg_loss * reg_loss_w


class DeferredLogs:
    """Drop-in for the `latest_logs` bookkeeping of model.py:59-67,111-134 without per-step `.item()`."""

    def __init__(self):
        self.latest_logs: Dict[str, Number] = {}
        self._last_logs: Dict[str, float] = {}
        self._calls_since_last_log = 0

    def update(self, **logs: Number) -> None:
        self._calls_since_last_log += 1
        for k, v in logs.items():
            if torch.is_tensor(v):
                v = v.detach().to(torch.float32)          # stays on its device; no synchronisation
            prev = self.latest_logs.get(k, 0)
            self.latest_logs[k] = prev + v

    def get_latest_logs(self) -> Dict[str, float]:
        if self._calls_since_last_log < 1:
            return {}
        # one readback for everything that lives on a device
        tens = {k: v for k, v in self.latest_logs.items() if torch.is_tensor(v)}
        if tens:
            keys = list(tens)
            host = torch.stack([tens[k].reshape(()) for k in keys]).cpu().tolist()
            for k, val in zip(keys, host):
                self.latest_logs[k] = val
        result = {}
        for k, v in self.latest_logs.items():
            result[k] = (float(v) - self._last_logs.get(k, 0)) / self._calls_since_last_log
        self._last_logs = {k: float(v) for k, v in self.latest_logs.items()}
        self._calls_since_last_log = 0
        return result
