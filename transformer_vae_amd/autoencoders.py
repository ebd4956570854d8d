"""MI355X-native latent bottleneck: drop-in for ``transformer_vae.autoencoders``.

Same class names, constructor signatures, registry keys, parameter names and forward
contract as the reference (citations: transformer_vae/autoencoders.py), so
``Funnel_T5_VAE_Model`` (model.py:84-88, 258-260) and ``VAE_Trainer`` (trainer.py:347) can
use these modules unchanged and checkpoints interchange:

    vae.encoder.token_to_latent.{weight,bias}      (L, d) / (L,)
    vae.decoder.latent_to_token.{weight,bias}      (d, L) / (d,)     absent when L == d
    vae.decoder.norm.{weight,bias}                 (d,)              default decoder only

What differs is where the arithmetic runs: every forward/backward below calls the
hand-written gfx950 kernels through ``functional`` (C ABI in include/tvae_hip.h).
Additions that default to the reference's behaviour:
  * ``true_samples=`` on ``EncoderDecoderVAE.forward`` injects the prior sample (the
    reference draws ``torch.randn(latent.size())`` inside ``_regularliser_loss``, :135);
  * ``mmd_scope="global"`` + ``process_group`` evaluate the MMD of the global batch,
    row-sharded over ranks (distributed.py); ``"local"`` is what the reference does under DDP.
"""
from __future__ import annotations

import logging

import torch
from torch import nn

from . import functional as F
from .model_outputs import BaseVAE_Output

logger = logging.getLogger(__name__)


class LatentEncoderNTokens(nn.Module):
    """Converts the first N hidden tokens into N separate latent codes (autoencoders.py:11-22)."""

    def __init__(self, config):
        super().__init__()
        self.token_to_latent = nn.Linear(config.t5.d_model, config.latent_size)
        self.n_tokens = config.n_latent_tokens
        self.tanh = nn.Tanh()   # kept for module-tree parity; tanh is fused into the projection kernel

    def forward(self, encoding) -> torch.Tensor:
        return F.latent_encode(encoding, self.token_to_latent.weight, self.token_to_latent.bias, self.n_tokens)


# alias for the name used by other revisions of the upstream project
LatentEncoderLargeTanh = LatentEncoderNTokens


class LatentDecoderNTokens(nn.Module):
    """Latent tokens -> full token hidden states (autoencoders.py:25-39).

    When latent_size == d_model the reference installs an identity (and no parameters); so do we."""

    def __init__(self, config):
        super().__init__()
        self.latent_size = config.latent_size
        if self.latent_size == config.t5.d_model:
            logger.warning("Latent decoder is not rescaling the latent code.")
            self.latent_to_token = None
        else:
            self.latent_to_token = nn.Linear(self.latent_size, config.t5.d_model)

    def _linear_params(self):
        if self.latent_to_token is None:
            return None, None
        return self.latent_to_token.weight, self.latent_to_token.bias

    def forward(self, latent) -> torch.Tensor:
        w, b = self._linear_params()
        if w is None:
            return latent          # identity, exactly like the reference's `lambda x: x`
        return F.latent_decode(latent, w, b)


class LatentDecoderFunnelNorm(LatentDecoderNTokens):
    """Linear + nn.LayerNorm(funnel.d_model, funnel.layer_norm_eps), fused (autoencoders.py:58-67)."""

    def __init__(self, config):
        super().__init__(config)
        self.norm = nn.LayerNorm(config.funnel.d_model, config.funnel.layer_norm_eps)

    def forward(self, latent) -> torch.Tensor:
        w, b = self._linear_params()
        return F.latent_decode(latent, w, b, self.norm.weight, self.norm.bias, self.norm.eps)


class LatentDecoderT5Norm(LatentDecoderNTokens):
    """Linear + a stock HF ``T5LayerFF`` block with dropout forced to 0 (autoencoders.py:42-55).

    The feed-forward block is third-party code in the reference too; only the linear runs on
    our kernels."""

    def __init__(self, config):
        super().__init__(config)
        from transformers.models.t5.modeling_t5 import T5LayerFF
        t_config = config.t5
        dropout_rate = t_config.dropout_rate
        t_config.dropout_rate = 0
        self.norm = T5LayerFF(t_config)
        t_config.dropout_rate = dropout_rate

    def forward(self, latent) -> torch.Tensor:
        return self.norm(super().forward(latent))


# alias for the name used in the north-star description
LatentDecoder = LatentDecoderFunnelNorm

VAE_ENCODER_MODELS = {
    "": LatentEncoderNTokens,
}
VAE_DECODER_MODELS = {
    "": LatentDecoderFunnelNorm,
    "t5_norm": LatentDecoderT5Norm,
    "no_norm": LatentDecoderNTokens,
}


class EncoderDecoderVAE(nn.Module):
    """An MMD-VAE bottleneck used between an encoder and a decoder stack (autoencoders.py:80-136)."""

    batch_size = None

    def __init__(self, encoder, decoder, use_reg_loss=True, mmd_scope="local", process_group=None,
                 compensate_ddp_mean=None, mmd_dtype=None, mmd_strict_fp32=False):
        """``encoder, decoder, use_reg_loss`` as the reference (autoencoders.py:88-92).  Additions:

        ``mmd_scope="global"`` + ``process_group``: one MMD over the batch of all ranks, row-sharded.
        ``compensate_ddp_mean`` (default: True with global scope): DistributedDataParallel AVERAGES parameter gradients
        over the R ranks while each rank holds only its rows' share of the global loss gradient; the local latent
        gradient is multiplied by R so that the averaged result equals the single-process global-batch gradient.
        Pass False only when gradients are summed, not averaged, across ranks.
        ``mmd_strict_fp32``: fp32 latents on the fp32-input matrix instruction instead of the default two or three fp16
        matrix-core products per fp32 product (same 1e-5 accuracy class, several times faster)."""
        super().__init__()
        if mmd_scope not in ("local", "global"):
            raise ValueError(f"mmd_scope must be 'local' or 'global', got {mmd_scope!r}")
        if mmd_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError(f"mmd_dtype must be None, torch.float32 or torch.bfloat16, got {mmd_dtype!r}")
        # None: the reference's promotion (bf16 latents against its always-fp32 prior run in fp32).
        # torch.bfloat16 (opt-in): latents and prior are rounded to bf16 and the regulariser runs on the bf16
        # matrix cores (fp32 accumulation, fp32 loss; gradient rounded to bf16) for latent sizes > 192.
        self.mmd_dtype = mmd_dtype
        self.encoder = encoder
        self.decoder = decoder
        self.use_reg_loss = use_reg_loss
        self.mmd_scope = mmd_scope
        self.process_group = process_group
        self.compensate_ddp_mean = (mmd_scope == "global") if compensate_ddp_mean is None else bool(compensate_ddp_mean)
        self.mmd_strict_fp32 = bool(mmd_strict_fp32)
        self._prior_generators = {}
        self._prior_base_seed = None
        # encode -> MMD -> decode as one autograd node when the stock pair is used plainly (same kernels and
        # numbers; less Python autograd bookkeeping per eager step).  False keeps the three separate nodes.
        self.fuse_step = True

    def _can_fuse(self, h, true_samples):
        enc, dec = self.encoder, self.decoder
        if not self.fuse_step or type(enc) is not LatentEncoderNTokens \
                or type(dec) not in (LatentDecoderNTokens, LatentDecoderFunnelNorm):
            return False
        if self.mmd_dtype is not None or torch.is_autocast_enabled() or not torch.is_tensor(h):
            return False
        if not h.is_cuda or h.dim() != 3 or h.dtype not in (torch.float32, torch.bfloat16):
            return False
        if h.shape[0] * min(int(enc.n_tokens), h.shape[1]) == 0 or h.shape[2] != enc.token_to_latent.weight.shape[1]:
            return False
        if dec.latent_to_token is None and not isinstance(dec, LatentDecoderFunnelNorm):
            return False                      # identity decoder: reconstructed_encoding IS the latent tensor
        for m in (enc, dec, enc.token_to_latent, dec.latent_to_token, getattr(dec, "norm", None)):
            if m is not None and (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks
                                  or m._backward_pre_hooks):
                return False
        # the stock pair's parameters, named (the types are exact above; walking self.parameters() costs 11 us per eager step)
        norm = getattr(dec, "norm", None)
        for m in (enc.token_to_latent, dec.latent_to_token, norm):
            if m is not None and (m.weight.dtype != h.dtype or (m.bias is not None and m.bias.dtype != h.dtype)):
                return False
        if self._global_scope_active():
            return False
        return True

    def _fused_forward(self, h, true_samples):
        enc, dec = self.encoder, self.decoder
        n_eff = min(int(enc.n_tokens), h.shape[1])
        L = enc.token_to_latent.weight.shape[0]
        if true_samples is None:
            true_samples = self._draw_prior((h.shape[0] * n_eff, L), h.device)     # fp32, as the reference (:135)
        else:
            true_samples = true_samples.reshape(h.shape[0] * n_eff, L).to(h.device)
        w, b = dec._linear_params()
        norm = getattr(dec, "norm", None)
        latent, recon, reg_loss = F.bottleneck_step(
            h, enc.token_to_latent.weight, enc.token_to_latent.bias, w, b,
            None if norm is None else norm.weight, None if norm is None else norm.bias, true_samples,
            enc.n_tokens, 1e-5 if norm is None else norm.eps, self.mmd_strict_fp32)
        return BaseVAE_Output(latent=latent, reconstructed_encoding=recon, reg_loss=reg_loss)

    def _model_forward(self, encoding, latent=None):
        if latent is None:
            latent = self.encoder(encoding)
        return self.decoder(latent), latent

    def forward(self, input_encoding=None, latent=None, skip_reg_loss=False, true_samples=None):
        if input_encoding is None and latent is None:
            raise ValueError("Both `input_encoding` and `latent` sent to VAE are None.")
        use_reg_loss = self.use_reg_loss and latent is None and skip_reg_loss is False  # don't regularise if given latent
        if use_reg_loss and self._can_fuse(input_encoding, true_samples):
            return self._fused_forward(input_encoding, true_samples)
        recon_encoding, latent = self._model_forward(input_encoding, latent=latent)
        if use_reg_loss:
            # each token encoding is a separate latent code (autoencoders.py:110-112)
            batch_size, n_latents_per_batch, latent_code_dim = latent.size()
            reg_loss = self._regularliser_loss(latent.reshape(-1, latent_code_dim), batch_size, n_latents_per_batch,
                                               true_samples)
        else:
            reg_loss = torch.tensor(0, device=latent.device)
        return BaseVAE_Output(latent=latent, reconstructed_encoding=recon_encoding, reg_loss=reg_loss)

    @staticmethod
    def _compute_kernel(x, y):
        """The (Nx, Ny) matrix exp(-mean((x_i - y_j)^2) / dim) of autoencoders.py:117-126, materialised and differentiable,
        for callers that want the matrix itself.  ``forward`` never forms it: the regulariser regenerates kernel tiles."""
        return F.rbf_kernel(x, y)

    def _compute_mmd(self, x, y):
        """MMD(x = prior samples, y = latents), unscaled (autoencoders.py:128-132); differentiable in y."""
        n, dim = y.shape
        loss, _ = F.mmd_row_block(y, y, x, 0, 1.0 / float(n) ** 2, 4.0 / (float(n) ** 2 * float(dim) ** 2),
                                  self.mmd_strict_fp32)
        return loss

    def _global_scope_active(self):
        if self.mmd_scope != "global":
            return False
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1

    def _draw_prior(self, size, device):
        """``torch.randn(latent.size())`` as the reference (:135), fp32 on the latent's device.

        Global scope: the reference launcher seeds every rank alike (train.py:391), so the default generator would hand
        every rank the SAME block and the gathered prior would be R copies of it.  Each rank's block therefore comes from
        its own stream (see ``_rank_generator``); every rank replays ALL streams in step (``_draw_prior_all_ranks``) and
        keeps its own block, so the streams stay aligned across ranks whichever form of the call drew last."""
        if not self._global_scope_active():
            return torch.randn(size, device=device)
        import torch.distributed as dist
        rank = dist.get_rank(self.process_group)
        n_local = int(size[0])
        return self._draw_prior_all_ranks(size, device)[rank * n_local:(rank + 1) * n_local]

    def _prior_base_seed_all_ranks(self, device):
        """One base seed for the per-rank prior streams, the SAME on every rank: rank 0's ``torch.initial_seed()``,
        broadcast once.  (With per-rank seeding, seed + rank, each rank would otherwise replay different streams and the
        ranks' priors -- hence their losses and gradients -- would silently belong to different problems.)"""
        if self._prior_base_seed is None:
            import torch.distributed as dist
            on_device = dist.get_backend(self.process_group) != "gloo"
            t = torch.tensor([torch.initial_seed() & ((1 << 62) - 1)], dtype=torch.int64, device=device if on_device else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(self.process_group or dist.group.WORLD, 0), group=self.process_group)
            self._prior_base_seed = int(t.item())
        return self._prior_base_seed

    def _rank_generator(self, rank, device):
        key = (str(device), int(rank))
        gen = self._prior_generators.get(key)
        if gen is None:
            gen = torch.Generator(device=device)
            gen.manual_seed(self._prior_base_seed_all_ranks(device) + 1 + int(rank))
            self._prior_generators[key] = gen
        return gen

    def _draw_prior_all_ranks(self, size_local, device):
        """Global scope, no caller prior: every rank replays the streams of ALL ranks (one base seed, broadcast from rank 0;
        advanced in step on every rank), so the (N, D) prior exists everywhere without an all-gather.  Block r is rank
        r's share.  Collective in spirit: every rank must make the same sequence of draws."""
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group)
        return torch.cat([torch.randn(size_local, device=device, generator=self._rank_generator(r, device))
                          for r in range(world)], dim=0)

    def _regularliser_loss(self, latent, batch_size=None, n_latents_per_batch=None, true_samples=None):
        """The reference's one-argument form, ``_regularliser_loss(latent)`` (autoencoders.py:134-136), returns the
        unscaled MMD(prior, latent).  With ``batch_size`` and ``n_latents_per_batch`` the result is
        MMD / batch_size * n_latents (:112) -- the scaling folded into the kernel's output scale, which is how
        ``forward`` calls it."""
        prior_all = None
        if true_samples is None:
            if self._global_scope_active() and batch_size is not None and n_latents_per_batch is not None and latent.numel():
                prior_all = self._draw_prior_all_ranks(latent.size(), latent.device)
            else:
                true_samples = self._draw_prior(latent.size(), latent.device)
        else:
            true_samples = true_samples.reshape(latent.shape).to(latent.device)
        if batch_size is None or n_latents_per_batch is None:
            if latent.numel() == 0:
                return torch.full((), float("nan"), dtype=torch.float32, device=latent.device)
            return self._compute_mmd(true_samples, latent)
        if latent.numel() == 0:
            # the reference takes the mean of empty kernel matrices (:130-131): nan, not an error
            return torch.full((), float("nan"), dtype=torch.float32, device=latent.device)
        if self.mmd_dtype is not None:
            latent = latent.to(self.mmd_dtype)
            true_samples = None if true_samples is None else true_samples.to(self.mmd_dtype)
            prior_all = None if prior_all is None else prior_all.to(self.mmd_dtype)
        if self._global_scope_active():
            from .distributed import sharded_mmd_reg_loss
            return sharded_mmd_reg_loss(latent, true_samples, batch_size, n_latents_per_batch, self.process_group,
                                        self.compensate_ddp_mean, strict_fp32=self.mmd_strict_fp32, prior_all=prior_all)
        return F.mmd_reg_loss(latent, true_samples, batch_size, n_latents_per_batch, self.mmd_strict_fp32)
