"""HIP-graph capture of a whole bottleneck step (forward + loss + backward).

The reference's shipped configurations (batch 8 ... 70, 5 latent tokens) are launch-bound on an
MI355X: ~30 kernels of a few microseconds each plus the autograd bookkeeping around them.  Every
entry point of the C ABI launches on the caller's stream, never allocates and never synchronises,
so a step can be captured once and replayed as ONE graph launch (0.44 -> 0.13 ms per step at BASELINE
cfg1, 0.35 -> 0.18 ms at cfg2).

    step = GraphedStep(lambda h, prior: loss_and_backward(vae, h, prior), (h, prior))
    out = step(h_new, prior_new)        # copies into the captured inputs, replays, returns static outputs

Rules of HIP-graph capture apply: fixed shapes, no host synchronisation inside ``fn`` (no ``.item()``,
use `losses.DeferredLogs`), and the tensors ``fn`` returns / the ``.grad`` fields it fills are static
buffers that the next replay overwrites.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 2,
                 zero_grads: Sequence[torch.Tensor] = ()):
        """``example_inputs``: device tensors with the shapes/dtypes of every later call (they become the
        captured input buffers).  ``zero_grads``: leaves (parameters, inputs) whose ``.grad`` ``fn`` fills;
        their grads are reset before the capture so that the graph WRITES them instead of accumulating."""
        if not example_inputs or not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedStep captures device work: inputs must live on the GPU (no CPU fallback)")
        self._inputs = list(example_inputs)
        self._leaves = list(zero_grads)
        device = self._inputs[0].device
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):              # warm-up off the capture: lazy initialisation, kernel attributes
            for _ in range(max(1, warmup)):
                self._clear()
                fn(*self._inputs)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self._clear()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*self._inputs)

    def _clear(self):
        for t in self._leaves:
            t.grad = None

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._inputs):
            raise ValueError(f"expected {len(self._inputs)} inputs, got {len(inputs)}")
        for dst, src in zip(self._inputs, inputs):
            if src is not dst:
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError(f"captured input is {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
                with torch.no_grad():
                    dst.copy_(src)
        self.graph.replay()
        return self.outputs
