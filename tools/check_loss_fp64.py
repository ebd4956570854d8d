#!/usr/bin/env python3
"""Loss of the fp32 MMD path against a chunked fp64 evaluation ON THE GPU (torch), at sizes the CPU oracle cannot do.
   python tools/check_loss_fp64.py N D"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import _lib, functional as F
if os.environ.get("TVAE_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])
N, D = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev))
y = torch.randn((N, D), generator=g, device=dev)


def esum(a, b, chunk=4096):
    a64, b64 = a.double(), b.double()
    na, nb = (a64 * a64).sum(1), (b64 * b64).sum(1)
    tot = torch.zeros((), dtype=torch.float64, device=dev)
    for r0 in range(0, a.shape[0], chunk):
        t = (na[r0:r0 + chunk, None] + nb[None, :] - 2.0 * (a64[r0:r0 + chunk] @ b64.T)) / float(D * D)
        tot += torch.expm1(-t.clamp_min(0)).sum()
    return tot


syy, szz, szy = esum(y, y), esum(z, z), esum(z, y)
ref = (syy + szz - 2 * szy) / float(N) ** 2
_, sums = F.mmd_row_block(z.clone().requires_grad_(True), z, y, 0, 1.0, 1.0)
for name, got, want in (("zz", sums[0], szz), ("zy", sums[1], szy), ("yy", sums[2], syy)):
    print("  S%s kernel %.12e fp64 %.12e rel %.2e abs/N^2 %.3e" % (name, float(got), float(want), (float(got) - float(want)) / abs(float(want)), (float(got) - float(want)) / float(N) ** 2))
zi = z.clone().requires_grad_(True)
loss = F.mmd_reg_loss(zi, y, batch_size=1, n_latents=1)
print("N %d D %d  kernel %.9e  fp64 %.9e  rel err %.2e" % (N, D, float(loss.detach()), float(ref), abs(float(loss.detach()) - float(ref)) / abs(float(ref))))
