#!/usr/bin/env python3
"""Medium-size random row blocks of the default (split-fp16) MMD against torch fp64 on the GPU -- sizes at which the
column splits, the L2 prefetch distance and the padding corrections of the 32-point kernels all come into play:
    python tools/fuzz_big.py [seed] [count]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from transformer_vae_amd import _lib  # noqa: E402
from transformer_vae_amd import functional as F  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rs = np.random.RandomState(seed)
dev = torch.device("cuda:0")
lib = _lib.lib()
fails = 0
for it in range(count):
    N = int(rs.randint(6000, 26000))
    D = int(rs.choice([200, 256, 300, 384, 400, 512, 640, 700, 768, 900, 1024]))
    lo = int(rs.randint(0, N - 1)) if rs.rand() < 0.6 else 0
    hi = int(rs.randint(lo + 1, N + 1)) if rs.rand() < 0.6 else N
    rows = hi - lo
    gen = torch.Generator(device=dev).manual_seed(seed * 1000 + it)
    z = torch.tanh(torch.randn((N, D), generator=gen, device=dev)) * float(rs.choice([1.0, 0.3, 3.0])) + float(rs.choice([0.0, 0.5, -2.0]))
    y = torch.randn((N, D), generator=gen, device=dev) * float(rs.choice([1.0, 2.0]))
    ws = F._workspace(lib.tvae_mmd_workspace_bytes_for(N, rows, D, 1, _lib.TVAE_F32), dev)
    s3 = torch.empty(3, dtype=torch.float64, device=dev)
    dz = torch.empty((rows, D), device=dev)
    _lib.check(lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), N, lo, rows, D, _lib.TVAE_F32, s3.data_ptr(), None, 1.0, dz.data_ptr(), 1.0,
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "mmd")
    z64, y64 = z.double(), y.double()
    inv = 1.0 / float(D * D)
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    ref = torch.zeros(3, dtype=torch.float64, device=dev)
    G = torch.zeros((rows, D), dtype=torch.float64, device=dev)
    R = torch.zeros(rows, dtype=torch.float64, device=dev)
    for r0 in range(lo, hi, 2048):
        sl = slice(r0, min(hi, r0 + 2048))
        o = slice(r0 - lo, min(hi, r0 + 2048) - lo)
        Ezz = torch.expm1(-((nz[sl, None] + nz[None, :] - 2.0 * (z64[sl] @ z64.T)) * inv).clamp_min(0))
        Ezy = torch.expm1(-((nz[sl, None] + ny[None, :] - 2.0 * (z64[sl] @ y64.T)) * inv).clamp_min(0))
        Eyy = torch.expm1(-((ny[sl, None] + ny[None, :] - 2.0 * (y64[sl] @ y64.T)) * inv).clamp_min(0))
        ref += torch.stack([Ezz.sum(), Ezy.sum(), Eyy.sum()])
        G[o] = Ezz @ z64 - Ezy @ y64
        R[o] = Ezz.sum(1) - Ezy.sum(1)
    grad = -(z64[lo:hi] * R[:, None]) + (z64.sum(0) - y64.sum(0))[None, :] + G
    # a row block's y-y share is "its rows x all columns" only without the symmetric split: compare z-z and z-y always, y-y
    # for whole-problem calls, and the gradient always
    e = ((s3 - ref).abs() / ref.abs()).tolist()
    gerr = float((dz.double() - grad).abs().max() / grad.abs().max())
    ok = e[0] < 1e-7 and e[1] < 1e-7 and (e[2] < 1e-7 or not (lo == 0 and hi == N)) and gerr < 1e-5
    fails += 0 if ok else 1
    print("%s N=%d D=%d rows [%d, %d)  sums err %.1e %.1e %.1e  grad err %.2e" % ("ok  " if ok else "FAIL", N, D, lo, hi, e[0], e[1], e[2], gerr), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
