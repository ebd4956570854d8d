import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from transformer_vae_amd import functional as F
from oracle import bottleneck_oracle as orc
dev = torch.device("cuda:0")
def run(N, D, lo, hi, strict):
    rs = np.random.RandomState(N * 31 + D)
    z = np.tanh(rs.standard_normal((N, D))).astype(np.float32)
    y = rs.standard_normal((N, D)).astype(np.float32)
    zt, yt = torch.from_numpy(z).to(dev), torch.from_numpy(y).to(dev)
    rows = zt[lo:hi].clone().requires_grad_(True)
    ls, gs = F.mmd_scales(7, 3, N, D)
    loss, sums = F.mmd_row_block(rows, zt, yt, lo, ls, gs, strict)
    loss.backward()
    z64, y64 = z.astype(np.float64), y.astype(np.float64)
    ref = orc.mmd_partial_sums(z64[lo:hi], y64[lo:hi], z64, y64)
    got = sums.cpu().numpy()
    gref = orc.mmd_grad_latent_rows(z64[lo:hi], z64, y64) * (3.0 / 7.0)
    ge = np.abs(rows.grad.cpu().numpy() - gref).max() / np.abs(gref).max()
    print(N, D, lo, hi, "strict" if strict else "split", ["%.2e" % (abs(g - r) / abs(r)) for g, r in zip(got, ref)], "grad %.2e" % ge, flush=True)
for args in [(4672, 31, 513, 1793), (4672, 31, 512, 1792), (4672, 31, 0, 4672), (4672, 31, 513, 1792), (4672, 31, 0, 1280), (4672, 32, 513, 1793),
             (4672, 77, 513, 1793), (4672, 31, 513, 1800), (2000, 31, 513, 1793), (4672, 31, 500, 1780)]:
    run(*args, False)
run(4672, 31, 513, 1793, True)
