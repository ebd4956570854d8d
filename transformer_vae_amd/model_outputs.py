"""Return type of the bottleneck (mirrors transformer_vae/model_outputs.py:8-24)."""
from dataclasses import dataclass
from typing import Optional

import torch
from transformers.utils import ModelOutput


@dataclass
class BaseVAE_Output(ModelOutput):
    """latent (B, n, L); reconstructed_encoding (B, n, d_model); reg_loss 0-dim tensor
    (fp32 when the regulariser ran, an int64 zero otherwise — autoencoders.py:114)."""

    latent: torch.FloatTensor = None
    reconstructed_encoding: torch.FloatTensor = None
    reg_loss: Optional[torch.FloatTensor] = None
