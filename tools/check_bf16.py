#!/usr/bin/env python3
"""Error of the bf16-matrix-core MMD path against the fp64 oracle on the same bf16-rounded inputs."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import functional as F
from oracle import bottleneck_oracle as orc
dev = torch.device("cuda:0")
for N, D in [(1100, 256), (1536, 300), (1000, 640), (2000, 768), (1040, 1024), (4097, 768)]:
    rs = np.random.RandomState(N + D)
    zt = torch.from_numpy(np.tanh(rs.standard_normal((N, D))).astype(np.float32)).to(dev).bfloat16().requires_grad_(True)
    yt = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32)).to(dev).bfloat16()
    reg = F.mmd_reg_loss(zt, yt, batch_size=1, n_latents=1)
    reg.backward()
    z64, y64 = zt.detach().double().cpu().numpy(), yt.double().cpu().numpy()
    ref = orc.mmd_gram(y64, z64)
    gref = orc.mmd_grad_latent_rows(z64, z64, y64)
    g = zt.grad.float().cpu().numpy()
    print(N, D, "loss rel err %.2e" % (abs(float(reg.detach()) - ref) / abs(ref)),
          "grad rel-to-max err %.2e" % (np.abs(g - gref).max() / np.abs(gref).max()), flush=True)
