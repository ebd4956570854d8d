#!/usr/bin/env python3
"""Time and cross-check the MMD entry point in its arithmetic modes, through the C ABI (development tool).

    python tools/mmd_modes.py N D [iters] [modes=split,strict,bf16] [rows]

Per mode: HIP-event time of the whole call, loss error against a chunked fp64 evaluation on the GPU (torch), and
the gradient error on a sample of rows against fp64 (torch, on the GPU), both as max|a-b| / max|b|."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import _lib  # noqa: E402

if os.environ.get("TVAE_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])

N, D = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
modes = (sys.argv[4] if len(sys.argv) > 4 else "split,strict,bf16").split(",")
rows = int(sys.argv[5]) if len(sys.argv) > 5 else N
CODES = {"split": 0, "bf16": 1, "strict": 2}
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
z = torch.tanh(torch.randn((N, D), generator=g, device=dev) + float(os.environ.get("TVAE_MODES_SHIFT", "0")))   # shift: latents with a non-zero mean
y = torch.randn((N, D), generator=g, device=dev) * float(os.environ.get("TVAE_MODES_YSCALE", "1"))   # > ~2.2 at D = 768: the E-form fallback of the Q-form calls
lib = _lib.lib()
ls, gs = 1.0 / float(N) ** 2, 4.0 / (float(N) ** 2 * float(D) ** 2)


def fp64_ref(zz, yy, sample):
    z64, y64 = zz.double(), yy.double()
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    inv = 1.0 / float(D * D)

    def esum(a, na, b, nb, r0, r1):
        t = (na[r0:r1, None] + nb[None, :] - 2.0 * (a[r0:r1] @ b.T)) * inv
        return torch.expm1(-t.clamp_min(0))

    tot = torch.zeros(3, dtype=torch.float64, device=dev)
    if N <= 70000:
        for r0 in range(0, N, 2048):
            r1 = min(N, r0 + 2048)
            if r0 < rows:
                rr = min(r1, rows)
                tot[0] += esum(z64, nz, z64, nz, r0, rr).sum()
                tot[1] += esum(z64, nz, y64, ny, r0, rr).sum()
                tot[2] += esum(y64, ny, y64, ny, r0, rr).sum()
    zs = z64[sample]
    ezz = esum(zs, nz[sample], z64, nz, 0, len(sample))
    ezy = esum(zs, nz[sample], y64, ny, 0, len(sample))
    rs = ezz.sum(1) - ezy.sum(1)
    grad = (-(zs * rs[:, None]) + (z64.sum(0) - y64.sum(0))[None, :] + ezz @ z64 - ezy @ y64) * (4.0 * inv / (N * N))
    return tot, grad


sample = torch.unique(torch.cat([torch.arange(0, min(rows, 48)), torch.arange(max(0, rows - 48), rows),
                                 torch.randint(0, rows, (160,), generator=torch.Generator().manual_seed(2))])).to(dev)
for mode in modes:
    code = CODES[mode]
    zi, yi = (z.bfloat16(), y.bfloat16()) if mode == "bf16" else (z, y)
    nbytes = lib.tvae_mmd_workspace_bytes_for(N, rows, D, 1, code)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    sums = torch.zeros(3, dtype=torch.float64, device=dev)
    loss = torch.zeros((), dtype=torch.float32, device=dev)
    dz = torch.zeros((rows, D), dtype=zi.dtype, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = []
    for _ in range(iters):
        ev[0].record()
        rc = lib.tvae_mmd_fwd_bwd(zi.data_ptr(), yi.data_ptr(), N, 0, rows, D, code, sums.data_ptr(), loss.data_ptr(), ls,
                                  dz.data_ptr(), gs, ws.data_ptr(), nbytes, st)
        ev[1].record()
        torch.cuda.synchronize()
        assert rc == 0, lib.tvae_last_error()
        times.append(ev[0].elapsed_time(ev[1]))
    ref_sums, ref_grad = fp64_ref(zi.float(), yi.float(), sample)
    gerr = float((dz[sample].double() - ref_grad).abs().max() / ref_grad.abs().max())
    # the same error measured against the kernel-dependent part of the gradient alone: the mean-shift term
    # gs (sum z - sum y) is exact arithmetic in every mode and dominates the max-norm at large D
    shift = (zi.float().double().sum(0) - yi.float().double().sum(0))[None, :] * (4.0 / (float(D) ** 2 * N * N))
    gerr_nl = float((dz[sample].double() - ref_grad).abs().max() / (ref_grad - shift).abs().max())
    ms = min(times)
    line = "%-6s N=%d D=%d rows=%d  %.3f ms (%.1f TFLOP/s alg.)  ws %.0f MiB  grad err %.2e (%.2e of the kernel part)" % (
        mode, N, D, rows, ms, 10.0 * rows * N * D / ms / 1e9, nbytes / 2 ** 20, gerr, gerr_nl)
    if N <= 70000:
        serr = ((sums - ref_sums).abs() / ref_sums.abs()).tolist()
        ref_loss = float((ref_sums[2] + ref_sums[0] - 2 * ref_sums[1]) * ls)
        line += "  sums err %s  loss err %.2e" % (["%.1e" % v for v in serr], abs(float(loss) - ref_loss) / abs(ref_loss)
                                                  if rows == N else float("nan"))
    print(line, flush=True)
