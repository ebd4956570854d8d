"""transformer_vae_amd — the latent bottleneck + MMD regulariser of Transformer-VAE on MI355X.

Hand-written gfx950 HIP kernels behind a C ABI (include/tvae_hip.h, built into
libtvae_hip.so) with a Python host that mirrors ``transformer_vae.autoencoders``.
"""
from . import _lib
from .autoencoders import (EncoderDecoderVAE, LatentDecoder, LatentDecoderFunnelNorm, LatentDecoderNTokens,
                           LatentDecoderT5Norm, LatentEncoderLargeTanh, LatentEncoderNTokens, VAE_DECODER_MODELS,
                           VAE_ENCODER_MODELS)
from .model_outputs import BaseVAE_Output
from .utils import slerp, upsample
from . import losses, trainer
from .graph import GraphedStep

build = _lib.build

__all__ = ["EncoderDecoderVAE", "LatentEncoderNTokens", "LatentEncoderLargeTanh", "LatentDecoderNTokens",
           "LatentDecoderFunnelNorm", "LatentDecoderT5Norm", "LatentDecoder", "VAE_ENCODER_MODELS",
           "VAE_DECODER_MODELS", "BaseVAE_Output", "slerp", "upsample", "losses", "trainer", "GraphedStep", "build"]
