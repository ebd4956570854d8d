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

    Contract kept from the reference: `latest_logs` holds running totals per key and may be written from outside
    (trainer.py:341-376 does `model.latest_logs['cycle_loss'] = ... + x.item()`); `get_latest_logs` returns, for EVERY key
    of `latest_logs`, the increase since the previous call divided by the number of updates since then (keys that did
    not move report 0), and {} when there was no update.

    What differs: device scalars passed to `update` are added IN PLACE into persistent fp64 accumulators on their device
    (one per key, plus a call counter), so `update` neither synchronises nor allocates after its first call and can sit
    inside a step captured by `graph.GraphedStep` (every replay advances the sums and the counter).  `get_latest_logs`
    folds those windows into `latest_logs` with ONE readback -- and none at all when no device value was added since
    the last call (unless an update was captured into a graph, whose replays the host cannot see)."""

    def __init__(self):
        self.latest_logs: Dict[str, Number] = {}
        self._last_logs: Dict[str, float] = {}
        self._calls_since_last_log = 0          # host-side count (updates made by Python, not graph replays)
        self._window: Dict[str, torch.Tensor] = {}     # key -> fp64 scalar on the value's device, sum over the window
        self._counter: Dict[torch.device, torch.Tensor] = {}   # device -> fp64 scalar, updates seen by that device
        self._device_dirty = False              # a device value was added since the last readback
        self._captured = False                  # an update ran under stream capture: replays advance the windows unseen

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
                self._device_dirty = True
                if v.device not in seen:
                    seen.add(v.device)
                    cnt = self._counter.get(v.device)
                    if cnt is None:
                        cnt = torch.zeros((), dtype=torch.float64, device=v.device)
                        self._counter[v.device] = cnt
                    cnt.add_(1.0)
                    if v.is_cuda and torch.cuda.is_current_stream_capturing():
                        self._captured = True
                self.latest_logs.setdefault(k, 0.0)
            else:
                self.latest_logs[k] = self.latest_logs.get(k, 0.0) + v     # model.py:113-114

    def get_latest_logs(self) -> Dict[str, float]:
        calls_dev = 0.0
        if self._counter and (self._device_dirty or self._captured):
            # one readback for everything that lives on a device
            keys = list(self._window)
            devs = list(self._counter)
            target = self._counter[devs[0]].device
            packed = torch.stack([self._window[k].to(target) for k in keys] + [self._counter[d].to(target) for d in devs])
            host = packed.cpu().tolist()
            for k, total in zip(keys, host[:len(keys)]):
                self.latest_logs[k] = float(self.latest_logs.get(k, 0.0)) + total
                self._window[k].zero_()
            # graph replays advance the device counters but not the host one: the largest device count is the truth
            calls_dev = max(host[len(keys):]) if devs else 0.0
            for d in devs:
                self._counter[d].zero_()
            self._device_dirty = False
        n = calls_dev if calls_dev > 0 else float(self._calls_since_last_log)
        if n < 1:
            return {}
        result = {}
        for k, v in self.latest_logs.items():                      # model.py:126-129
            v = float(v)
            result[k] = (v - self._last_logs.get(k, 0.0)) / n
            self._last_logs[k] = v
        self._calls_since_last_log = 0
        return result
