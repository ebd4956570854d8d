"""One-off: a whole step captured into a HIP graph at a size where the rendezvous counters (a memset node) and the
bf16 column chunks are active; replay must reproduce the eager loss and gradient bit for bit.
    python tools/graph_big.py [B S d n L]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from transformer_vae_amd.graph import GraphedStep
dev = torch.device("cuda:0")
B, S, d, n, L = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1100, 30, 256, 30, 768)))
for mmd_dtype in (None, torch.bfloat16):
    vae = bench.make_vae(d, L, n, dev, torch.float32, "local", 1, mmd_dtype)
    g = torch.Generator(device=dev).manual_seed(5)
    h = torch.randn((B, S, d), generator=g, device=dev).requires_grad_(True)
    prior = torch.randn((B * n, L), generator=g, device=dev)
    g0 = torch.randn((B, n, d), generator=g, device=dev) / (B * n * d) ** 0.5
    params = list(vae.parameters())
    def step(hh, pp):
        out = vae(input_encoding=hh, true_samples=pp)
        (out.reg_loss * 0.5 + (out.reconstructed_encoding * g0).sum()).backward()
        return out.reg_loss.detach()
    for _ in range(2):
        h.grad = None
        for p in params: p.grad = None
        ref_loss = float(step(h, prior))
    ref_grad = h.grad.clone()
    gs = GraphedStep(step, (h, prior), zero_grads=[h] + params)
    for _ in range(3):
        out = gs(h, prior)
    torch.cuda.synchronize()
    print(mmd_dtype, B * n, ref_loss, float(out), float((h.grad - ref_grad).abs().max()), float(ref_grad.abs().max()))
    assert float(out) == ref_loss and torch.equal(h.grad, ref_grad)
print("graph replay ok")
