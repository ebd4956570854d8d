#!/usr/bin/env python3
"""Host + device overhead of the staged (row-sharded) MMD orchestration, on a ONE-rank RCCL group: the same problem through
distributed.sharded_mmd_reg_loss (three C ABI stages, RCCL all-gathers into the workspace, one scalar read-back) and through
the one-call form.       python tools/prof_staged_rccl.py N D [iters]"""
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from transformer_vae_amd import functional as F  # noqa: E402
import transformer_vae_amd.distributed as tdist  # noqa: E402

N, D = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
gen = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=gen, device=dev))
y = torch.randn((N, D), generator=gen, device=dev)


def run(fn):
    zi = z.clone().requires_grad_(True)
    loss = fn(zi)
    loss.backward()
    return float(loss.detach()), zi.grad


def timed(fn):
    run(fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run(fn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


one = lambda zi: F.mmd_reg_loss(zi, y, batch_size=N // 30 or 1, n_latents=30)
staged = lambda zi: tdist.sharded_mmd_reg_loss(zi, y, batch_size_local=N // 30 or 1, n_latents=30)
la, ga = run(one)
lb, gb = run(staged)
print("loss one-call %.9e staged %.9e  grad max diff %.2e of %.2e" % (la, lb, float((ga - gb).abs().max()), float(ga.abs().max())))
print("one-call %.3f ms   staged (1-rank RCCL) %.3f ms" % (timed(one), timed(staged)))
dist.destroy_process_group()
