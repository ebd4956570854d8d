"""Global-batch MMD, row-sharded over a process group (one process per GPU, RCCL over xGMI).

The reference has no collective of its own: under DDP every rank computes the MMD of its
LOCAL batch.  That behaviour is ``mmd_scope="local"`` and needs nothing
from this file.  ``mmd_scope="global"`` evaluates the MMD of the concatenated batch:

  1. all-gather the (N/R, D) latent and prior row blocks          -> (N, D) on every rank
  2. each rank evaluates ITS row block of Kzz / Kzy / Kyy against all N columns and the
     COMPLETE gradient of its own rows (no gradient exchange: SURVEY.md 8e)
  3. all-reduce three fp64 partial sums                            -> identical loss everywhere

The compute of step 2 is injected (``row_block_fn``) so that the collective logic can be
exercised on CPU with gloo by the tests; the package default is the HIP kernel and nothing
else.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import functional as F


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate equally sized row blocks from every rank, in rank order."""
    world = dist.get_world_size(group)
    x = x.contiguous()
    if x.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device collectives: rehearsal / test runs only (the product backend is RCCL)
        return all_gather_rows(x.cpu(), group).to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if x.is_cuda:
        dist.all_gather_into_tensor(out, x, group=group)
    else:
        dist.all_gather(list(out.chunk(world, dim=0)), x, group=group)
    return out


def all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    """Sum over ranks, in place; returns x."""
    if x.is_cuda and dist.get_backend(group) == "gloo":
        c = x.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        x.copy_(c)
        return x
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def sharded_mmd_reg_loss(latent2d: torch.Tensor, prior_local: Optional[torch.Tensor], batch_size_local: int, n_latents: int,
                         group=None, compensate_ddp_mean: bool = False,
                         row_block_fn: Optional[Callable] = None, strict_fp32: bool = False,
                         prior_all: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reg_loss of the GLOBAL batch; every rank returns the same value.

    latent2d / prior_local: this rank's (N/R, D) rows (equal counts on all ranks).
    ``prior_all``: the (N, D) prior of ALL ranks in rank order, when every rank can produce it itself (seeded draws:
    EncoderDecoderVAE under global scope) -- the prior all-gather, half of the bytes on the wire, is then skipped.
    The gradient that flows into ``latent2d`` is d(global reg)/d(local rows), complete.
    ``compensate_ddp_mean``: DDP averages parameter gradients over R ranks while each rank
    holds only its rows' share of the global loss gradient; multiplying the local gradient
    by R makes the averaged result equal the single-process global-batch gradient.  Default False (gradients summed
    across ranks, or no DDP at all); EncoderDecoderVAE passes True under global scope, where DDP is the caller."""
    fn = row_block_fn or (lambda *a: F.mmd_row_block(*a, strict_fp32))
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local, dim = latent2d.shape
    z_all = all_gather_rows(latent2d.detach(), group)
    n_total = n_local * world
    if prior_all is not None:
        if tuple(prior_all.shape) != (n_total, dim):
            raise ValueError(f"prior_all must be ({n_total}, {dim}), got {tuple(prior_all.shape)}")
        y_all = prior_all
    else:
        y_all = all_gather_rows(prior_local, group)
    loss_scale, grad_scale = F.mmd_scales(batch_size_local * world, n_latents, n_total, dim)
    if compensate_ddp_mean:
        grad_scale *= world
    local, sums = fn(latent2d, z_all, y_all, rank * n_local, loss_scale, grad_scale)
    sums = all_reduce_sum(sums.detach().clone(), group)
    reg = ((sums[2] + sums[0] - 2.0 * sums[1]) * loss_scale).to(torch.float32)
    # value: the global loss; gradient: through this rank's row block only
    return reg.detach() + (local - local.detach())
