#!/usr/bin/env python3
"""Projections (encode, decode + LayerNorm) fwd+bwd at a BASELINE config's full size against torch fp64 on the GPU.
   python tools/check_proj_fp64.py cfg4"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402



def run(name):
    dev = torch.device("cuda:0")
    B, S, d, n, L, cfg_id, _ = bench.CONFIGS[name]
    n_eff = min(n, S)
    vae = bench.make_vae(d, L, n, dev, torch.float32, "local", 1)
    vae.use_reg_loss = False
    gen = torch.Generator(device=dev).manual_seed(3)
    h = torch.randn((B, S, d), generator=gen, device=dev).requires_grad_(True)
    g0 = torch.randn((B, n_eff, d), generator=gen, device=dev) / (B * n_eff * d) ** 0.5
    gz = torch.randn((B, n_eff, L), generator=gen, device=dev) / (B * n_eff * L) ** 0.5   # stands in for the MMD gradient
    out = vae(input_encoding=h)
    ((out.reconstructed_encoding * g0).sum() + (out.latent * gz).sum()).backward()

    We = vae.encoder.token_to_latent.weight.detach().double().requires_grad_(True)
    be = vae.encoder.token_to_latent.bias.detach().double().requires_grad_(True)
    h64 = h.detach().double().requires_grad_(True)
    z = torch.tanh(h64[:, :n_eff] @ We.T + be)
    ref = {"latent": z}
    if hasattr(vae.decoder, "latent_to_token"):
        Wd = vae.decoder.latent_to_token.weight.detach().double().requires_grad_(True)
        bd = vae.decoder.latent_to_token.bias.detach().double().requires_grad_(True)
        r = z @ Wd.T + bd
    else:
        Wd = bd = None
        r = z
    gam = vae.decoder.norm.weight.detach().double().requires_grad_(True)
    bet = vae.decoder.norm.bias.detach().double().requires_grad_(True)
    rec = torch.nn.functional.layer_norm(r, (d,), gam, bet, vae.decoder.norm.eps)
    ((rec * g0.double()).sum() + (z * gz.double()).sum()).backward()


    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max())

    res = {"latent": rel(out.latent.detach(), z.detach()), "recon": rel(out.reconstructed_encoding.detach(), rec.detach()),
           "dh": rel(h.grad[:, :n_eff], h64.grad[:, :n_eff]), "dWe": rel(vae.encoder.token_to_latent.weight.grad, We.grad),
           "dbe": rel(vae.encoder.token_to_latent.bias.grad, be.grad),
           "dgamma": rel(vae.decoder.norm.weight.grad, gam.grad), "dbeta": rel(vae.decoder.norm.bias.grad, bet.grad)}
    if Wd is not None:
        res["dWd"] = rel(vae.decoder.latent_to_token.weight.grad, Wd.grad)
        res["dbd"] = rel(vae.decoder.latent_to_token.bias.grad, bd.grad)
    assert float(h.grad[:, n_eff:].abs().max()) == 0.0 if S > n_eff else True
    return res




if __name__ == "__main__":
    for k, v in run(sys.argv[1]).items():
        print("%-7s %.2e" % (k, v))
