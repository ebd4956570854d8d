#!/usr/bin/env python3
"""Experiment for DESIGN 9 (Q-form gradient): a library built with -DTVAE_QFORM_PROTO makes the split pipelined
gradient kernel accumulate Q = E + t (one fp16 product) instead of E (three); this script adds the linear part in closed
form (torch, fp64 and fp32) and compares with the fp64 gradient.

    TVAE_LIB_OVERRIDE=.../libtvae_qform.so python tools/qform_proto.py N D"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import _lib  # noqa: E402

if os.environ.get("TVAE_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])
N, D = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev) + float(os.environ.get("TVAE_MODES_SHIFT", "0")))
y = torch.randn((N, D), generator=g, device=dev)
lib = _lib.lib()
ls, gs = 1.0 / float(N) ** 2, 4.0 / (float(N) ** 2 * float(D) ** 2)
nbytes = lib.tvae_mmd_workspace_bytes_for(N, N, D, 1, 0)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
sums = torch.zeros(3, dtype=torch.float64, device=dev)
loss = torch.zeros((), dtype=torch.float32, device=dev)
dz = torch.zeros((N, D), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
best = 1e9
for _ in range(3):
    ev[0].record()
    rc = lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), N, 0, N, D, 0, sums.data_ptr(), loss.data_ptr(), ls, dz.data_ptr(), gs,
                              ws.data_ptr(), nbytes, st)
    ev[1].record()
    torch.cuda.synchronize()
    assert rc == 0, lib.tvae_last_error()
    best = min(best, ev[0].elapsed_time(ev[1]))

sample = torch.unique(torch.cat([torch.arange(0, 48), torch.arange(N - 48, N),
                                 torch.randint(0, N, (160,), generator=torch.Generator().manual_seed(2))])).to(dev)


def linear_part(dt):
    zz, yy = z.to(dt), y.to(dt)
    inv = 1.0 / float(D * D)
    nz, ny = (zz * zz).sum(1), (yy * yy).sum(1)
    shift = zz.sum(0) - yy.sum(0)
    m3 = (nz[:, None] * zz).sum(0) - (ny[:, None] * yy).sum(0)
    M2 = zz.T @ zz - yy.T @ yy
    zs = zz[sample]
    r_lin = -inv * ((nz.sum() - ny.sum()) - 2.0 * (zs @ shift))
    g_lin = -inv * (m3[None, :] + nz[sample][:, None] * shift[None, :] - 2.0 * (zs @ M2))
    return (gs * (-(zs * r_lin[:, None]) + g_lin)).double()


z64, y64 = z.double(), y.double()
nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
inv = 1.0 / float(D * D)
zs = z64[sample]


def em1(a, na, b, nb):
    return torch.expm1(-((na[:, None] + nb[None, :] - 2.0 * (a @ b.T)) * inv).clamp_min(0))


ezz, ezy = em1(zs, nz[sample], z64, nz), em1(zs, nz[sample], y64, ny)
rs = ezz.sum(1) - ezy.sum(1)
shift64 = (z64.sum(0) - y64.sum(0))[None, :]
ref = (-(zs * rs[:, None]) + shift64 + ezz @ z64 - ezy @ y64) * gs
for dt in (torch.float64, torch.float32):
    got = dz[sample].double() + linear_part(dt)
    err = float((got - ref).abs().max() / ref.abs().max())
    err_nl = float((got - ref).abs().max() / (ref - shift64 * gs).abs().max())
    print("N=%d D=%d  %.3f ms  closed form in %s: grad err %.2e (%.2e of the kernel part)" % (N, D, best, str(dt)[6:], err, err_nl), flush=True)
