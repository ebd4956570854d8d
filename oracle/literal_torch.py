"""Literal CPU port of the reference bottleneck step in torch.  TEST INFRASTRUCTURE ONLY.

What the reference executes on its CPU path is a chain of stock ATen ops driven by
autograd (op table).  This module re-creates that op sequence
— Linear over all S tokens, tanh, slice, Linear, LayerNorm, then for each of the
three kernels: tile both operands to (N,N,D), subtract, square, mean, /dim, exp,
mean — so that ``bench.py`` can time "what the reference does" on the GPU box's host
cores, where the reference itself never travels.  Equivalence to the imported
reference is asserted in ``tests/golden/make_golden.py`` (run in the build container).

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this file.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _tiled_kernel(a, b):
    # autoencoders.py:118-126 — both operands materialised at (Na, Nb, D)
    na, dim = a.shape
    nb = b.shape[0]
    ta = a.view(na, 1, dim).repeat(1, nb, 1)
    tb = b.view(1, nb, dim).repeat(na, 1, 1)
    return torch.exp(-torch.mean((ta - tb) ** 2, dim=2) / dim * 1.0)


def _mmd(prior, latent):
    # autoencoders.py:128-132
    return _tiled_kernel(prior, prior).mean() + _tiled_kernel(latent, latent).mean() \
        - 2 * _tiled_kernel(prior, latent).mean()


def forward_literal(h, params, n_tokens, prior, eps=1e-9):
    """Returns (latent, recon, reg_loss) as autograd-tracked tensors."""
    z = torch.tanh(F.linear(h, params["We"], params["be"]))[:, :n_tokens, :]   # :22
    r = F.linear(z, params["Wd"], params["bd"]) if "Wd" in params else z       # :32-39
    if "gamma" in params:
        r = F.layer_norm(r, (r.shape[-1],), params["gamma"], params["beta"], eps)  # :64-67
    bsz, n_eff, lat = z.shape
    reg = _mmd(prior, z.reshape(-1, lat)) / bsz * n_eff                        # :111-112
    return z, r, reg


def step_literal(h, params, n_tokens, prior, g_recon, reg_weight, eps=1e-9):
    """One fwd+bwd of  reg_weight*reg + <recon, g_recon>; returns the loss value.

    ``h`` and every tensor in ``params`` must be leaves with requires_grad=True."""
    for t in (h, *params.values()):
        t.grad = None
    _, r, reg = forward_literal(h, params, n_tokens, prior, eps)
    loss = reg * reg_weight + (r * g_recon).sum()
    loss.backward()
    return float(loss.detach())
