#!/usr/bin/env python3
"""Feasibility check for the next step after the Q-form gradient: second-order closed form for the LOSS sums.

    Q(t) = expm1(-t) + t = t^2/2 + R3(t),   R3 = -t^3/6 + t^4/24 - ...

The sum of t^2 over all pairs is a function of moments (norms, sums, Z^T Z, Y^T Y -- the matrices the Q-form already
forms), so the tile kernels would only have to sum R3, ~t/3 ~ 1e-3 of Q: would a ONE-product Gram (hi x hi fp16, no lo
parts at all) then be enough for the loss?  Emulated here with torch: t from fp16(s x) @ fp16(s y)^T with fp32
accumulation, R3 by its polynomial, sums against fp64.  Prints the error of each R3 sum relative to the Q sum it would
be added to (the loss needs ~1e-9 there; the shipped two-product kernels reach 4e-11 .. 2e-10).

    python tools/r3_proto.py N D"""
import sys

import torch

N, D = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev))
y = torch.randn((N, D), generator=g, device=dev)
amax = float(torch.maximum(z.abs().max(), y.abs().max()))
s = 2.0 ** (12 - int(torch.ceil(torch.log2(torch.tensor(amax)))))      # the kernels' power-of-two scale
zh, yh = (z * s).half(), (y * s).half()
nz, ny = (z.double() ** 2).sum(1), (y.double() ** 2).sum(1)
inv = 1.0 / float(D * D)


def r3_poly(t):
    return t * t * t * (-1.0 / 6 + t * (1.0 / 24 + t * (-1.0 / 120 + t * (1.0 / 720 - t / 5040))))


def sums(a, ah, na, b, bh, nb):
    """(sum R3 from the one-product fp16 Gram in fp32, sum R3 in fp64, sum Q in fp64) over all pairs"""
    lo = torch.zeros((), dtype=torch.float64, device=dev)
    r64 = torch.zeros((), dtype=torch.float64, device=dev)
    q64 = torch.zeros((), dtype=torch.float64, device=dev)
    for r0 in range(0, N, 2048):
        r1 = min(N, r0 + 2048)
        gram = (ah[r0:r1].float() @ bh.float().T) / (s * s)          # fp16 values, fp32 accumulate: the hi x hi product
        t32 = ((na[r0:r1, None].float() + nb[None, :].float() - 2.0 * gram) * inv).clamp_min(0)
        lo += r3_poly(t32).double().sum()
        t64 = ((na[r0:r1, None] + nb[None, :] - 2.0 * (a[r0:r1].double() @ b.double().T)) * inv).clamp_min(0)
        q = torch.expm1(-t64) + t64
        q64 += q.sum()
        r64 += (q - 0.5 * t64 * t64).sum()
    return float(lo), float(r64), float(q64)


for name, args in (("z-z", (z, zh, nz, z, zh, nz)), ("z-y", (z, zh, nz, y, yh, ny)), ("y-y", (y, yh, ny, y, yh, ny))):
    lo, r64, q64 = sums(*args)
    print("%s  sum R3 %.6e (fp64 %.6e)  error / sum Q = %.2e   (R3 / Q = %.2e)" % (name, lo, r64, abs(lo - r64) / abs(q64), abs(r64 / q64)), flush=True)
