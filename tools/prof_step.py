#!/usr/bin/env python3
"""A few eager steps of one BASELINE config (for rocprofv3 --kernel-trace):  python tools/prof_step.py cfg2 [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformer_vae_amd import _lib  # noqa: E402

if os.environ.get("TVAE_LIB_OVERRIDE"):          # kernel experiments: load a variant build instead
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
B, S, d, n, L, cfg_id, _ = bench.CONFIGS[name]
n_eff = min(n, S)
vae = bench.make_vae(d, L, n, dev, torch.float32, "local", 1)
gen = torch.Generator(device=dev).manual_seed(1)
h = torch.randn((B, S, d), generator=gen, device=dev).requires_grad_(True)
prior = torch.randn((B * n_eff, L), generator=gen, device=dev)
g0 = torch.randn((B, n_eff, d), generator=gen, device=dev) / (B * n_eff * d) ** 0.5
for _ in range(steps):
    h.grad = None
    for p in vae.parameters():
        p.grad = None
    out = vae(input_encoding=h, true_samples=prior)
    (out.reg_loss * 0.5 + (out.reconstructed_encoding * g0).sum()).backward()
torch.cuda.synchronize()
print("ok", float(out.reg_loss.detach()))
