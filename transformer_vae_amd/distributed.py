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

Large problems (``STAGED_MIN_ELEMENTS``) take the STAGED form instead (``_StagedMMD``): nothing is
replicated before the tile kernels --

  A. each rank reduces ITS rows to column sums and max |value|    -> all-gather of a (2 D + 2) fp64 record
  B. each rank centres, scales and splits ITS rows into the fp16 image the tile kernels read, with norms,
     centred moments and (Q-form gradient) its share of Z^T Z, Y^T Y
                                                                  -> all-gather of the image rows and norms (prior
                                                                     first), of the moment records; all-reduce of
                                                                     the two D x D matrices
  C. tile kernels over the rank's rows: the y-y sum-only launch goes out as soon as the prior's image is
     complete, while the latent image is still on the wire; then the gradient launch and finalize
  D. all-reduce of the three fp64 partial sums, as above.

The stages are three C ABI calls (include/tvae_hip.h) behind ``functional.HipShardStages``; tests pass a
stand-in with the same four methods to rehearse the collectives with gloo on CPU.
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


import os

# n_total * D from which the staged form is used (below: one gather, one call); TVAE_STAGED_MIN_ELEMENTS overrides (tests)
STAGED_MIN_ELEMENTS = int(os.environ.get("TVAE_STAGED_MIN_ELEMENTS", 1 << 22))


def _all_gather_flat(x: torch.Tensor, group=None) -> torch.Tensor:
    """(world, x.numel()) of every rank's 1-D tensor."""
    return all_gather_rows(x.reshape(1, -1), group)


class _Pending:
    """An all-gather of equal byte ranges INTO a workspace region, possibly still on the wire."""

    def __init__(self, region: torch.Tensor, rank: int, world: int, group, overlap: bool):
        parts = region.view(world, -1)
        self.keep = parts[rank].clone()            # (the input of an in-flight collective must not alias its output)
        self.work = None
        if region.is_cuda and dist.get_backend(group) != "gloo":
            self.work = dist.all_gather_into_tensor(region, self.keep, group=group, async_op=overlap)
        else:
            parts.copy_(_all_gather_flat(self.keep, group))

    def wait(self):
        if self.work is not None and hasattr(self.work, "wait"):
            self.work.wait()                       # the current stream waits; the host does not
        self.work = self.keep = None


class _StagedMMD(torch.autograd.Function):
    """reg share and gradient of this rank's rows, the preparation sharded too (module docstring, A-C)."""

    @staticmethod
    def forward(ctx, z_rows, y_rows, world: int, rank: int, loss_scale: float, grad_scale: float, code: int, lay, group, stages,
                grad_enabled: bool):
        n_rows, dim = z_rows.shape
        n_total, row0 = n_rows * world, rank * n_rows
        want_grad = bool(grad_enabled) and bool(ctx.needs_input_grad[0])
        dt = z_rows.dtype
        z = z_rows.detach().contiguous()
        y = y_rows.detach().to(dt).contiguous()
        dev = z.device
        ws = stages.workspace(lay[0], dev)

        def region(off, nbytes, dtype=torch.uint8):
            return ws[off:off + nbytes].view(dtype)

        def combine(records, max_word):
            total = records.sum(0)
            total[max_word] = records[:, max_word].max()
            return total

        # A: column sums and max |value| of the own rows, summed over ranks
        if lay[7] >= 0:
            stages.stats(z, y, n_total, row0, code, want_grad, ws)
            rec_a = region(lay[7], lay[8] * 8, torch.float64)
            rec_a.copy_(combine(_all_gather_flat(rec_a.clone(), group), lay[8] - 2))
        # B: the own rows' image, norms and moments
        stages.prep(z, y, n_total, row0, code, want_grad, ws)
        rec_b = region(lay[9], lay[11] * 8, torch.float64)
        rec_b.copy_(combine(_all_gather_flat(region(lay[10], lay[11] * 8, torch.float64).clone(), group), lay[11] - 2))
        need_lo = lay[4] > 0
        if need_lo and lay[14]:
            # other ranks' lo rows are read only when the device drops the Q-form (4 max-norm / D^2 >= 1/16): skip their
            # gather when that is out of reach by a margin -- one scalar read back, the only host sync of the call
            need_lo = not (4.0 * float(rec_b[lay[11] - 2]) / float(dim) ** 2 < 0.9 / 16.0)
        overlap = True
        row_bytes = lay[3]
        pend_y, pend_z = [], []
        for pend, img_off, nrm_off in ((pend_y, lay[2], lay[6]), (pend_z, lay[1], lay[5])):     # the prior first
            pend.append(_Pending(region(nrm_off, n_total * 4), rank, world, group, overlap))
            pend.append(_Pending(region(img_off, n_total * row_bytes), rank, world, group, overlap))
            if need_lo:
                pend.append(_Pending(region(img_off + lay[4], n_total * row_bytes), rank, world, group, overlap))
        gram = None
        if lay[12] >= 0:
            gram = region(lay[12], lay[13] * 4, torch.float32)
            if gram.is_cuda and dist.get_backend(group) != "gloo":
                gram = (gram, dist.all_reduce(gram, op=dist.ReduceOp.SUM, group=group, async_op=overlap))
            else:
                all_reduce_sum(gram, group)
                gram = None
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dz = torch.empty((n_rows, dim), dtype=dt, device=dev) if want_grad else None
        # C: the prior-only part while the latent image travels, then the rest
        for pnd in pend_y:
            pnd.wait()
        stages.tiles(z, n_total, row0, code, sums, loss, loss_scale, dz, grad_scale, 1, ws)
        for pnd in pend_z:
            pnd.wait()
        if gram is not None and hasattr(gram[1], "wait"):
            gram[1].wait()
        stages.tiles(z, n_total, row0, code, sums, loss, loss_scale, dz, grad_scale, 2, ws)
        if want_grad:
            ctx.save_for_backward(dz)
        ctx.z_dtype = z_rows.dtype
        ctx.mark_non_differentiable(sums)
        return loss, sums

    @staticmethod
    def backward(ctx, g_loss, _g_sums):
        (dz,) = ctx.saved_tensors
        return ((dz.float() * g_loss.float()).to(ctx.z_dtype),) + (None,) * 10


def staged_layout(latent2d: torch.Tensor, prior_dtype, world: int, strict_fp32: bool, stages, want_grad: bool):
    """(code, layout) when the staged form applies to this call, else None."""
    n_local, dim = latent2d.shape
    if n_local * world * dim < STAGED_MIN_ELEMENTS:
        return None
    dt = torch.bfloat16 if (latent2d.dtype == torch.bfloat16 and prior_dtype == torch.bfloat16) else torch.float32
    if latent2d.dtype != dt:
        return None
    code = F.mmd_code(dt, strict_fp32)
    lay = stages.layout(n_local * world, n_local, dim, want_grad, code)
    return None if lay is None else (code, lay)


def sharded_mmd_reg_loss(latent2d: torch.Tensor, prior_local: Optional[torch.Tensor], batch_size_local: int, n_latents: int,
                         group=None, compensate_ddp_mean: bool = False,
                         row_block_fn: Optional[Callable] = None, strict_fp32: bool = False,
                         prior_all: Optional[torch.Tensor] = None, stages=None) -> torch.Tensor:
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
    n_total = n_local * world
    if row_block_fn is None and (stages is not None or latent2d.is_cuda):
        stages = stages or F.HipShardStages
        want_grad = torch.is_grad_enabled() and latent2d.requires_grad
        prior_dtype = prior_all.dtype if prior_all is not None else prior_local.dtype
        picked = staged_layout(latent2d, prior_dtype, world, strict_fp32, stages, want_grad)
        if picked is not None:
            code, lay = picked
            if prior_all is not None:
                if tuple(prior_all.shape) != (n_total, dim):
                    raise ValueError(f"prior_all must be ({n_total}, {dim}), got {tuple(prior_all.shape)}")
                prior_local = prior_all[rank * n_local:(rank + 1) * n_local]
            loss_scale, grad_scale = F.mmd_scales(batch_size_local * world, n_latents, n_total, dim)
            if compensate_ddp_mean:
                grad_scale *= world
            local, sums = _StagedMMD.apply(latent2d, prior_local, world, rank, loss_scale, grad_scale, code, lay, group, stages,
                                           torch.is_grad_enabled())
            sums = all_reduce_sum(sums.detach().clone(), group)
            reg = ((sums[2] + sums[0] - 2.0 * sums[1]) * loss_scale).to(torch.float32)
            return reg.detach() + (local - local.detach())
    z_all = all_gather_rows(latent2d.detach(), group)
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
