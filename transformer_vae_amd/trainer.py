"""Trainer-side glue for the bottleneck path (SURVEY.md 8f rows f3 and f4's trainer half).

What the reference does around the path, restated without its hard `wandb` import (trainer.py:3):

  * `VAE_Trainer.get_loss_grad` (trainer.py:258-281): mean over DataParallel replicas when
    `n_gpu > 1`, divide by `gradient_accumulation_steps`, backward through the AMP scaler / a
    caller-supplied engine / plain autograd, return the detached loss;
  * `VAE_Trainer.training_step` (trainer.py:378-401) minus the critic / cycle-loss branches
    (outside the path): `model.train()`, forward, `get_loss_grad`;
  * `VAE_Trainer.gradual_interpolation_inputs` / `random_interpolation_inputs`
    (trainer.py:283-301) on the fused `slerp`;
  * `VAE_TrainingArguments` (train.py:33-100) gains the fields that surface the global-batch MMD:
    `mmd_scope` ("local" = each rank's own batch, exactly the reference under DDP; "global" =
    row-sharded MMD over the concatenated batch) and `mmd_compensate_ddp_mean`.

Nothing here computes on the CPU: tensors are only routed to the HIP entry points.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields, make_dataclass
from typing import Any, Callable, Dict, Optional

import torch
from torch import nn

from .autoencoders import EncoderDecoderVAE
from .utils import slerp

MMD_SCOPES = ("local", "global")


@dataclass
class MMDScopeArguments:
    """Mixin for `VAE_TrainingArguments` (train.py:33-100); parsed by `HfArgumentParser` like its other fields."""
    mmd_scope: str = field(
        default="local",
        metadata={"help": "MMD regulariser batch scope under data parallelism. Options: " + ", ".join(MMD_SCOPES)
                          + ". `local` reproduces the reference (each rank regularises its own batch); "
                            "`global` evaluates one MMD over the batch of all ranks (row-sharded)."},
    )
    mmd_compensate_ddp_mean: bool = field(
        default=True,
        metadata={"help": "With --mmd_scope global: scale the local latent gradient by the world size so that "
                          "DDP's gradient averaging reproduces the single-process global-batch gradient."},
    )

    def validate_mmd(self) -> None:
        if self.mmd_scope not in MMD_SCOPES:       # same style as config.py:98-99 (assertIn)
            raise ValueError(f'Unknown `mmd_scope` Got: "{self.mmd_scope}" Expected one of: {list(MMD_SCOPES)}')


def extend_training_arguments(base_cls: type, name: Optional[str] = None) -> type:
    """`VAE_TrainingArguments` + the two MMD fields, as a new dataclass (base class untouched)."""
    extra = [(f.name, f.type, field(default=f.default, metadata=dict(f.metadata))) for f in fields(MMDScopeArguments)]
    return make_dataclass(name or base_cls.__name__, extra, bases=(base_cls,),
                          namespace={"validate_mmd": MMDScopeArguments.validate_mmd})


def configure_vae(vae: EncoderDecoderVAE, args: Any, process_group=None) -> EncoderDecoderVAE:
    """Apply the parsed arguments to `model.vae` (built at model.py:84-88)."""
    scope = getattr(args, "mmd_scope", "local")
    if scope not in MMD_SCOPES:
        raise ValueError(f'Unknown `mmd_scope` Got: "{scope}" Expected one of: {list(MMD_SCOPES)}')
    vae.mmd_scope = scope
    vae.process_group = process_group
    vae.compensate_ddp_mean = bool(getattr(args, "mmd_compensate_ddp_mean", True))
    return vae


def get_loss(outputs, labels=None, label_smoother: Optional[Callable] = None):
    """trainer.py:243-256 (the `past_index` bookkeeping stays with the caller's Trainer)."""
    if labels is not None:
        if label_smoother is None:
            raise ValueError("labels given without a label_smoother")
        return label_smoother(outputs, labels)
    return outputs["loss"] if isinstance(outputs, dict) else outputs[0]


def get_loss_grad(loss: torch.Tensor, n_gpu: int = 1, gradient_accumulation_steps: int = 1, scaler=None,
                  backward: Optional[Callable[[torch.Tensor], None]] = None) -> torch.Tensor:
    """trainer.py:258-281.  `scaler` = a `torch.amp.GradScaler` (the `use_amp` branch); `backward` = an
    engine's backward (the apex / deepspeed branches: `lambda l: engine.backward(l)`)."""
    if n_gpu > 1:
        loss = loss.mean()
    if gradient_accumulation_steps > 1:
        loss = loss / gradient_accumulation_steps
    if scaler is not None:
        scaler.scale(loss).backward()
    elif backward is not None:
        backward(loss)
    else:
        loss.backward()
    return loss.detach()


def training_step(model: nn.Module, inputs: Dict[str, Any], n_gpu: int = 1, gradient_accumulation_steps: int = 1,
                  scaler=None, backward=None, label_smoother=None, autocast_dtype: Optional[torch.dtype] = None
                  ) -> torch.Tensor:
    """trainer.py:378-401 without the critic / cycle-loss side branches."""
    model.train()
    inputs = dict(inputs)
    labels = inputs.pop("labels") if (label_smoother is not None and "labels" in inputs) else None
    if autocast_dtype is not None:
        with torch.autocast("cuda", dtype=autocast_dtype):
            outputs = model(**inputs)
            loss = get_loss(outputs, labels, label_smoother)
    else:
        outputs = model(**inputs)
        loss = get_loss(outputs, labels, label_smoother)
    return get_loss_grad(loss, n_gpu, gradient_accumulation_steps, scaler, backward)


def gradual_interpolation_inputs(latent_start: torch.Tensor, latent_end: torch.Tensor, device=None,
                                 interpolate_all_at_once: bool = False):
    """trainer.py:283-295: 11 ratios 0.0 .. 1.0; returns ((11, n, L) interpolations, ratios)."""
    device = latent_start.device if device is None else device
    ratios = torch.arange(0, 1.1, 0.1, device=device)
    n_tok, dim = latent_start.size(0), latent_start.size(1)
    if interpolate_all_at_once:
        a, b = latent_start.reshape(1, -1), latent_end.reshape(1, -1)
        out = slerp(ratios, a.repeat(11, 1), b.repeat(11, 1))
        return out.view(11, n_tok, dim), ratios
    rep = ratios.repeat(n_tok, 1).T.reshape(-1)          # [0, 0, .1, .1, ...] for two latent tokens
    out = slerp(rep, latent_start.repeat(11, 1), latent_end.repeat(11, 1))
    return out.view(11, n_tok, dim), ratios


def random_interpolation_inputs(latent: torch.Tensor, generator: Optional[torch.Generator] = None):
    """trainer.py:297-301.  The reference writes `latent[::-1]`, which torch rejects (negative step);
    the evident intent — pair sample i with sample B-1-i — is `flip(0)`."""
    bsz = latent.size(0)
    ratios = 0.5 - (torch.rand(bsz, device=latent.device, generator=generator) - 0.5).abs()
    flat = latent.reshape(bsz, -1)
    out = slerp(ratios, flat, flat.flip(0))
    return out.view_as(latent), ratios
