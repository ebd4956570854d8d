#!/usr/bin/env python3
"""Run the MMD entry point alone a few times (for rocprofv3 --kernel-trace / --pmc passes).

    python tools/prof_mmd.py N D [iters] [rows]
`rows` < N profiles a row block (what one rank of a sharded run executes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import functional as F  # noqa: E402

N, D = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = int(sys.argv[4]) if len(sys.argv) > 4 else N
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev))
y = torch.randn((N, D), generator=g, device=dev)
ls, gs = F.mmd_scales(max(1, N // 30), 30, N, D)
for _ in range(iters):
    zr = z[:rows].clone().requires_grad_(True)
    loss, sums = F.mmd_row_block(zr, z, y, 0, ls, gs)
torch.cuda.synchronize()
print("loss", float(loss), "sums", sums.tolist())
