"""Development check: run-to-run determinism of the MMD call, eager and under HIP-graph replay."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformer_vae_amd as amd
from transformer_vae_amd import functional as F
dev = torch.device("cuda:0")
N, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 768)
gen = torch.Generator(device=dev).manual_seed(99)
DT = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
def draw():
    return torch.tanh(torch.randn((N, D), generator=gen, device=dev)).to(DT), torch.randn((N, D), generator=gen, device=dev).to(DT)
z_buf, y_buf = draw(); z_buf.requires_grad_(True)
def step(z, y):
    loss = F.mmd_reg_loss(z, y, batch_size=1, n_latents=1); loss.backward(); return loss
gs = amd.GraphedStep(step, (z_buf, y_buf), zero_grads=[z_buf])
for it in range(6):
    z_new, y_new = draw()
    loss = gs(z_new, y_new); torch.cuda.synchronize()
    g_graph = z_buf.grad.clone()
    outs = []
    for rep in range(2):
        zt = z_new.clone().requires_grad_(True)
        want = F.mmd_reg_loss(zt, y_new, batch_size=1, n_latents=1); want.backward(); torch.cuda.synchronize()
        outs.append(zt.grad.clone())
    d = (g_graph != outs[0])
    print(it, "loss equal", torch.equal(loss.detach(), want.detach()), "eager==eager", torch.equal(outs[0], outs[1]),
          "graph!=eager elements", int(d.sum()), "rows touched", int(d.any(1).sum()), "cols touched", int(d.any(0).sum()),
          "max rel", float((g_graph.float() - outs[0].float()).abs().max() / outs[0].float().abs().max()))
    if d.any():
        r = d.any(1).nonzero().flatten()
        print("   first rows", r[:10].tolist(), "last rows", r[-5:].tolist())
