#!/usr/bin/env python3
"""Run the MMD entry point alone a few times (for rocprofv3 --kernel-trace / --pmc passes).

    python tools/prof_mmd.py N D [iters] [rows]
`rows` < N profiles a row block (what one rank of a sharded run executes).
TVAE_PROF_DTYPE=bf16 feeds bf16 latents/priors; the script prints the wall time of the last iteration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import _lib, functional as F  # noqa: E402

if os.environ.get("TVAE_LIB_OVERRIDE"):          # kernel experiments: load a variant build instead
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])

N, D = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = int(sys.argv[4]) if len(sys.argv) > 4 else N
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev))
y = torch.randn((N, D), generator=g, device=dev)
if os.environ.get("TVAE_PROF_DTYPE", "fp32") == "bf16":
    z, y = z.bfloat16(), y.bfloat16()
import time  # noqa: E402
ls, gs = F.mmd_scales(max(1, N // 30), 30, N, D)
for _ in range(iters):
    zr = z[:rows].clone().requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, sums = F.mmd_row_block(zr, z, y, 0, ls, gs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("last iteration %.3f ms -> %.1f TFLOP/s algorithmic" % (dt * 1e3, 10.0 * rows * N * D / dt / 1e12))
print("loss", float(loss), "sums", sums.tolist())
