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


def esum(a, b, chunk=2048):
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

if len(sys.argv) > 3:      # gradient check (closed form in fp64, chunked)
    z64, y64 = z.double(), y.double()
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    shift = z64.sum(0) - y64.sum(0)
    gref = torch.empty_like(z64)
    for r0 in range(0, N, 2048):
        zr = z64[r0:r0 + 2048]
        ezz = torch.expm1(-((nz[r0:r0 + 2048, None] + nz[None, :] - 2 * zr @ z64.T) / float(D * D)).clamp_min(0))
        ezy = torch.expm1(-((nz[r0:r0 + 2048, None] + ny[None, :] - 2 * zr @ y64.T) / float(D * D)).clamp_min(0))
        rs = ezz.sum(1) - ezy.sum(1)
        gref[r0:r0 + 2048] = (-(zr * rs[:, None]) + shift[None, :] + ezz @ z64 - ezy @ y64) * (4.0 / (float(N) ** 2 * D * D))
    loss.backward()
    err = (zi.grad.double() - gref).abs().max() / gref.abs().max()
    print("  gradient max err / max |g| = %.2e   (max |g| %.3e)" % (float(err), float(gref.abs().max())))
