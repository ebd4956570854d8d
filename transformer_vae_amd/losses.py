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
    """Drop-in for the `latest_logs` bookkeeping of model.py:59-67,111-134 without per-step `.item()`.

    Device scalars are added IN PLACE into persistent fp64 accumulators on their device (one per key, plus a call
    counter), so `update` neither synchronises nor allocates after its first call and can sit inside a step captured
    by `graph.GraphedStep`: every replay advances the sums and the counter.  `get_latest_logs` reads all of them
    back in ONE transfer, returns the means over the window (the reference's numbers) and zeroes the window:
    a logged value is never the difference of two ever-growing sums.  `latest_logs` keeps the reference's running
    totals as Python doubles, refreshed at each readback."""

    def __init__(self):
        self.latest_logs: Dict[str, Number] = {}
        self._calls_since_last_log = 0          # host-side count (updates made by Python, not graph replays)
        self._window: Dict[str, torch.Tensor] = {}     # key -> fp64 scalar on the value's device, sum over the window
        self._host_window: Dict[str, float] = {}
        self._counter: Dict[torch.device, torch.Tensor] = {}   # device -> fp64 scalar, updates seen by that device

    def update(self, **logs: Number) -> None:
        self._calls_since_last_log += 1
        seen = set()
        for k, v in logs.items():
            if torch.is_tensor(v):
                v = v.detach()
                acc = self._window.get(k)
                if acc is None or acc.device != v.device:
                    acc = torch.zeros((), dtype=torch.float64, device=v.device)
                    self._window[k] = acc
                acc.add_(v.reshape(()))                        # in place: no allocation, no synchronisation
                if v.device not in seen:
                    seen.add(v.device)
                    cnt = self._counter.get(v.device)
                    if cnt is None:
                        cnt = torch.zeros((), dtype=torch.float64, device=v.device)
                        self._counter[v.device] = cnt
                    cnt.add_(1.0)
            else:
                self._host_window[k] = self._host_window.get(k, 0.0) + float(v)
            self.latest_logs.setdefault(k, 0.0)

    def get_latest_logs(self) -> Dict[str, float]:
        if self._calls_since_last_log < 1 and not self._counter:
            return {}
        # one readback for everything that lives on a device
        keys = list(self._window)
        devs = list(self._counter)
        calls_dev: Dict[torch.device, float] = {}
        sums: Dict[str, float] = {}
        if keys or devs:
            target = (self._window[keys[0]] if keys else self._counter[devs[0]]).device
            packed = torch.stack([self._window[k].to(target) for k in keys] + [self._counter[d].to(target) for d in devs])
            host = packed.cpu().tolist()
            sums = dict(zip(keys, host[:len(keys)]))
            calls_dev = dict(zip(devs, host[len(keys):]))
            for k in keys:
                self._window[k].zero_()
            for d in devs:
                self._counter[d].zero_()
        if self._calls_since_last_log < 1 and not any(c > 0 for c in calls_dev.values()):
            return {}
        result = {}
        for k, total in sums.items():
            # graph replays advance the device counter but not the host one: the device's count is the truth
            n = calls_dev.get(self._window[k].device, 0.0) or float(self._calls_since_last_log)
            result[k] = total / n
            self.latest_logs[k] = float(self.latest_logs.get(k, 0.0)) + total
        host_calls = max(self._calls_since_last_log, 1)
        for k, total in self._host_window.items():
            result[k] = total / host_calls
            self.latest_logs[k] = float(self.latest_logs.get(k, 0.0)) + total
        self._host_window = {}
        self._calls_since_last_log = 0
        return result
