#!/usr/bin/env python3
"""Host-side profile (cProfile) of eager steps of one BASELINE config:  python tools/prof_host.py cfg2 [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
B, S, d, n, L, cfg_id, _ = bench.CONFIGS[name]
n_eff = min(n, S)
vae = bench.make_vae(d, L, n, dev, torch.float32, "local", 1)
gen = torch.Generator(device=dev).manual_seed(1)
h = torch.randn((B, S, d), generator=gen, device=dev).requires_grad_(True)
prior = torch.randn((B * n_eff, L), generator=gen, device=dev)
g0 = torch.randn((B, n_eff, d), generator=gen, device=dev) / (B * n_eff * d) ** 0.5
params = list(vae.parameters())


def step():
    h.grad = None
    for p in params:
        p.grad = None
    out = vae(input_encoding=h, true_samples=prior)
    (out.reg_loss * 0.5 + (out.reconstructed_encoding * g0).sum()).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("eager ms/step: %.4f" % ((time.perf_counter() - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
