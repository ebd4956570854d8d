#!/usr/bin/env python3
"""Reference point for the projection shapes: what the vendor GEMM (torch.mm -> hipBLASLt/rocBLAS) takes, fp32."""
import torch, time
dev = torch.device("cuda:0")
shapes = {"cfg2 enc fwd (350x768 @ 768x77)": (350, 768, 77), "cfg2 dec fwd (350x77 @ 77x768)": (350, 77, 768),
          "cfg2 dWe (77x350 @ 350x768)": (77, 350, 768), "cfg3 enc fwd (5120x768 @ 768x77)": (5120, 768, 77),
          "cfg3 dec fwd (5120x77 @ 77x768)": (5120, 77, 768), "cfg3 dWe (77x5120 @ 5120x768)": (77, 5120, 768)}
for name, (m, k, n) in shapes.items():
    a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev)
    for _ in range(20): torch.mm(a, b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): torch.mm(a, b)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per GEMM (50 back to back in a graph)")
