#!/usr/bin/env python3
"""Round-3 additions to the golden vectors, generated from the REAL reference (imported from /root/reference).

Run in the build container only:   python tests/golden/make_golden_r3.py
Writes
  * vae_t5norm_bwd.npz      -- the `t5_norm` decoder (autoencoders.py:42-55): forward AND the gradients of
                               reg_loss * w + <recon, G0> for the hidden states and every parameter, in fp64;
  * compute_kernel_cases.npz -- `EncoderDecoderVAE._compute_kernel(x, y)` (autoencoders.py:117-126), Nx != Ny included,
                               with the gradients of <K, C> for both arguments, in fp64.
The existing fixtures are not touched (make_golden.py)."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts the repo root and /root/reference on sys.path, imports the reference as mg.ref)

orc = mg.orc


def t5norm_case():
    name, B, S, d, n, L, cfg_id = mg.CASES[2]
    h, params, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, "no_norm")
    torch.manual_seed(5)
    vae, cfg = mg.build_reference(d, L, n, "t5_norm", params, 1e-9)
    v64 = copy.deepcopy(vae).double()
    ht = torch.from_numpy(h).double().requires_grad_(True)
    with mg.injected_prior(torch.from_numpy(prior).double()):
        o = v64(input_encoding=ht)
    (o.reg_loss * mg.REG_W + (o.reconstructed_encoding * torch.from_numpy(g0).double()).sum()).backward()
    store = dict(h=h, prior=prior, g0=g0, reg_weight=np.float64(mg.REG_W),
                 latent=o.latent.detach().numpy(), recon=o.reconstructed_encoding.detach().numpy(),
                 reg_loss=o.reg_loss.detach().numpy(), dh=ht.grad.numpy(), shape=np.array([B, S, d, n, L, cfg_id], np.int64),
                 t5=np.array([cfg.t5.d_model, cfg.t5.d_ff, cfg.t5.d_kv, cfg.t5.num_heads], np.int64))
    for k, v in vae.state_dict().items():
        store["sd_" + k] = v.numpy()
    for k, p in v64.named_parameters():
        store["grad_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "vae_t5norm_bwd.npz"), **store)
    print("vae_t5norm_bwd: reg64=%.9e, %d parameter gradients" % (float(o.reg_loss), len(list(v64.named_parameters()))))


def compute_kernel_cases():
    store = {}
    shapes = [(5, 9, 3), (40, 40, 77), (33, 70, 77), (129, 64, 300), (64, 200, 768), (1, 7, 1024)]
    for ci, (nx, ny, dim) in enumerate(shapes):
        rs = np.random.RandomState(600 + ci)
        x = np.tanh(rs.standard_normal((nx, dim))).astype(np.float32)
        y = (rs.standard_normal((ny, dim)) * (1.0 if ci != 3 else 6.0)).astype(np.float32)
        c = rs.standard_normal((nx, ny)).astype(np.float32)
        xt = torch.from_numpy(x).double().requires_grad_(True)
        yt = torch.from_numpy(y).double().requires_grad_(True)
        k = mg.ref.EncoderDecoderVAE._compute_kernel(xt, yt)
        (k * torch.from_numpy(c).double()).sum().backward()
        k32 = mg.ref.EncoderDecoderVAE._compute_kernel(torch.from_numpy(x), torch.from_numpy(y))
        assert np.abs(orc.rbf_kernel_literal(x.astype(np.float64), y.astype(np.float64)) - k.detach().numpy()).max() < 1e-14
        store[f"x{ci}"], store[f"y{ci}"], store[f"c{ci}"] = x, y, c
        store[f"k{ci}"], store[f"dx{ci}"], store[f"dy{ci}"] = k.detach().numpy(), xt.grad.numpy(), yt.grad.numpy()
        store[f"k32err{ci}"] = np.float64(np.abs(k32.numpy() - k.detach().numpy()).max())
    store["count"] = np.int64(len(shapes))
    np.savez_compressed(os.path.join(HERE, "compute_kernel_cases.npz"), **store)
    print("compute_kernel_cases:", shapes)


if __name__ == "__main__":
    t5norm_case()
    compute_kernel_cases()
