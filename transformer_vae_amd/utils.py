"""Numeric helpers next to the bottleneck (mirrors transformer_vae/utils.py:17-34 and the funnel
`upsample` imported at transformer_vae/model.py:10), running on the gfx950 kernels."""
from .functional import slerp, upsample

__all__ = ["slerp", "upsample"]
