#!/usr/bin/env python3
"""Achieved HBM rate of the SURVEY 8f row kernels (upsample fwd/bwd, slerp) at BASELINE-sized inputs.

    python tools/bench_aux.py [iters]
Bytes are the algorithmic ones (each input read once, each output written once); time is the mean of `iters`
back-to-back launches between two events on torch's current stream (the stream the C ABI is handed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_vae_amd import _lib, functional as F  # noqa: E402

if os.environ.get("TVAE_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
PEAK = 8.0e12


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def report(name, nbytes, sec):
    print(f"{name:58s} {sec * 1e6:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / sec / 1e12:6.2f} TB/s  "
          f"{nbytes / sec / PEAK:5.2f} of 8 TB/s", flush=True)


# upsample: (B, n, d) -> (B, T, d), stride 4, separate cls (model.py:265-271); cfg3: 237-token memory, cfg4-like
for (B, n, d, stride, T, dt) in [(1024, 60, 768, 4, 237, torch.float32), (8192, 9, 1024, 4, 30, torch.float32),
                                 (1024, 60, 768, 4, 237, torch.bfloat16)]:
    x = torch.randn((B, n, d), device=dev).to(dt)
    out = F.upsample(x, stride, T, True, False)
    esz = x.element_size()
    g = torch.randn_like(out)
    lib = _lib.lib()
    code = F._DTYPES[dt]
    st = torch.cuda.current_stream().cuda_stream
    dx = torch.empty_like(x)
    sec = timed(lambda: lib.tvae_upsample_fwd(x.data_ptr(), out.data_ptr(), B, n, d, stride, T, 1, 0, code, st))
    report(f"upsample_fwd {tuple(x.shape)}->{tuple(out.shape)} {str(dt)[6:]}", (x.numel() + out.numel()) * esz, sec)
    sec = timed(lambda: lib.tvae_upsample_bwd(g.data_ptr(), dx.data_ptr(), B, n, d, stride, T, 1, 0, code, st))
    report(f"upsample_bwd {tuple(out.shape)}->{tuple(x.shape)} {str(dt)[6:]}", (x.numel() + out.numel()) * esz, sec)

# slerp on latent rows: (rows, dim) pairs (trainer.py:283-295 interpolates flattened latents)
for (rows, dim, dt) in [(245760, 768, torch.float32), (65536, 1024, torch.float32), (5120, 77, torch.float32),
                        (245760, 768, torch.bfloat16)]:
    a = torch.randn((rows, dim), device=dev).to(dt)
    b = torch.randn((rows, dim), device=dev).to(dt)
    o = torch.empty_like(a)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    sec = timed(lambda: lib.tvae_slerp(a.data_ptr(), b.data_ptr(), None, 0.3, o.data_ptr(), rows, dim, F._DTYPES[dt], st))
    report(f"slerp ({rows}, {dim}) {str(dt)[6:]}", 3 * a.numel() * a.element_size(), sec)
