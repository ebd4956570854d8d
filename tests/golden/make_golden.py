#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (imported from /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The reference never travels to the GPU box; these vectors do.  For every case the
script (1) runs the reference's EncoderDecoderVAE in fp32 and in fp64 with an injected
prior (torch.randn is patched for the duration of the call so that the reference's own
`_regularliser_loss` draws exactly our tensor), (2) takes autograd gradients of
`reg_loss*w + <recon, G0>`, (3) asserts the numpy oracle and the literal torch port
against those results, and (4) writes inputs + expected outputs.

Large hidden-state tensors are not stored: they are regenerated from numpy's frozen
RandomState stream by oracle.bottleneck_oracle.synth_inputs and verified by checksum.
"""
import contextlib
import copy
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from transformers import FunnelConfig, T5Config  # noqa: E402
import transformer_vae.autoencoders as ref  # noqa: E402  (the reference)

from oracle import bottleneck_oracle as orc  # noqa: E402
from oracle import literal_torch as lit  # noqa: E402

REG_W = 0.5
BIG = 100_000  # tensors above this many elements are stored as float32 / regenerated

CASES = [
    # name       B   S   d    n   L   cfg_id
    ("cfg1",     8,  60, 512, 5,  77, 1),
    ("cfg2",     70, 30, 768, 5,  77, 2),
    ("tiny",     6,  9,  32,  3,  11, 10),
    ("n_gt_s",   4,  16, 64,  30, 7,  11),
    ("ident",    4,  8,  32,  2,  32, 12),
    ("ragged",   3,  5,  40,  4,  19, 13),
]
DECODERS = ["", "no_norm"]


@contextlib.contextmanager
def injected_prior(prior_t):
    """Make the reference's torch.randn(latent.size(), device=...) return `prior_t`."""
    real = torch.randn

    def fake(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        assert shape == tuple(prior_t.shape), (shape, prior_t.shape)
        return prior_t.clone()

    torch.randn = fake
    try:
        yield
    finally:
        torch.randn = real


def build_reference(d, L, n, decoder, params, eps_default):
    cfg = SimpleNamespace(t5=T5Config(d_model=d, d_ff=2 * d, d_kv=8, num_heads=2),
                          funnel=FunnelConfig(d_model=d), latent_size=L, n_latent_tokens=n)
    vae = ref.EncoderDecoderVAE(ref.VAE_ENCODER_MODELS[""](cfg), ref.VAE_DECODER_MODELS[decoder](cfg), True)
    sd = vae.state_dict()
    sd["encoder.token_to_latent.weight"] = torch.from_numpy(params["We"])
    sd["encoder.token_to_latent.bias"] = torch.from_numpy(params["be"])
    if "Wd" in params:
        sd["decoder.latent_to_token.weight"] = torch.from_numpy(params["Wd"])
        sd["decoder.latent_to_token.bias"] = torch.from_numpy(params["bd"])
    if "gamma" in params:
        sd["decoder.norm.weight"] = torch.from_numpy(params["gamma"])
        sd["decoder.norm.bias"] = torch.from_numpy(params["beta"])
    vae.load_state_dict(sd)
    assert cfg.funnel.layer_norm_eps == eps_default
    return vae, cfg


def run_reference(vae, h, prior, g0, dtype):
    vae = copy.deepcopy(vae).to(dtype)
    for p in vae.parameters():
        p.grad = None
    ht = torch.from_numpy(h).to(dtype).requires_grad_(True)
    # fp32 run: the prior is fp32 exactly as the reference draws it.  fp64 run: the same values as
    # float64, so that the prior-prior kernel is not silently evaluated in fp32 (ground truth).
    with injected_prior(torch.from_numpy(prior).to(dtype)):
        out = vae(input_encoding=ht)
    loss = out.reg_loss * REG_W + (out.reconstructed_encoding * torch.from_numpy(g0).to(dtype)).sum()
    loss.backward()
    res = dict(latent=out.latent, recon=out.reconstructed_encoding, reg_loss=out.reg_loss, dh=ht.grad)
    names = {"encoder.token_to_latent.weight": "dWe", "encoder.token_to_latent.bias": "dbe",
             "decoder.latent_to_token.weight": "dWd", "decoder.latent_to_token.bias": "dbd",
             "decoder.norm.weight": "dgamma", "decoder.norm.bias": "dbeta"}
    for k, p in vae.named_parameters():
        res[names[k]] = p.grad
    return {k: v.detach().numpy() for k, v in res.items()}


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def check_oracle(h, params, n, prior, g0, eps, r64):
    """The numpy oracle (fp64, Gram form + analytic grads) must equal the reference's fp64 autograd."""
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    h64, y64, g64 = h.astype(np.float64), prior.astype(np.float64), g0.astype(np.float64)
    fwd = orc.bottleneck_forward(h64, p64, n, y64, eps)
    bwd = orc.bottleneck_backward(h64, p64, n, fwd, g64, REG_W, y64)
    pairs = [("latent", fwd["latent"]), ("recon", fwd["recon"]), ("dh", bwd["h"]), ("dWe", bwd["We"]),
             ("dbe", bwd["be"])]
    if "Wd" in params:
        pairs += [("dWd", bwd["Wd"]), ("dbd", bwd["bd"])]
    if "gamma" in params:
        pairs += [("dgamma", bwd["gamma"]), ("dbeta", bwd["beta"])]
    worst = abs(fwd["reg_loss"] - float(r64["reg_loss"])) / abs(float(r64["reg_loss"]))
    assert worst < 1e-9, ("reg_loss", worst)
    for name, val in pairs:
        e = relerr(val, r64[name])
        assert e < 1e-10, (name, e)
        worst = max(worst, e)
    return worst


def check_literal_port(h, params, n, prior, g0, eps, r32):
    """oracle/literal_torch.py must reproduce the fp32 reference (same ops, same order)."""
    pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    ht = torch.from_numpy(h).clone().requires_grad_(True)
    lit.step_literal(ht, pt, n, torch.from_numpy(prior), torch.from_numpy(g0), REG_W, eps)
    assert relerr(ht.grad.numpy(), r32["dh"]) < 1e-6
    assert relerr(pt["We"].grad.numpy(), r32["dWe"]) < 1e-6
    _, _, reg = lit.forward_literal(ht, pt, n, torch.from_numpy(prior), eps)
    assert abs(float(reg) - float(r32["reg_loss"])) <= 1e-6 * abs(float(r32["reg_loss"])) + 1e-9


def pack(name, arr, store):
    arr = np.asarray(arr)
    if arr.size > BIG:
        store[name] = arr.astype(np.float32)
    else:
        store[name] = arr


def main():
    os.makedirs(HERE, exist_ok=True)
    eps = 1e-9
    summary = []
    for (name, B, S, d, n, L, cfg_id) in CASES:
        for dec in DECODERS:
            h, params, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, dec)
            vae, _ = build_reference(d, L, n, dec, params, eps)
            r32 = run_reference(vae, h, prior, g0, torch.float32)
            r64 = run_reference(vae, h, prior, g0, torch.float64)
            n_eff = min(n, S)
            assert r64["latent"].shape == (B, n_eff, L)
            assert np.all(r64["dh"][:, n_eff:, :] == 0) and np.all(r32["dh"][:, n_eff:, :] == 0)
            worst = check_oracle(h, params, n, prior, g0, eps, r64)
            if h.size <= 400_000:
                check_literal_port(h, params, n, prior, g0, eps, r32)
            store = dict(shape=np.array([B, S, d, n, L, cfg_id], np.int64), eps=np.float64(eps),
                         reg_weight=np.float64(REG_W), decoder=np.array(dec))
            store["h_checksum"] = np.array([h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum()])
            if h.size <= BIG:
                store["h"] = h
            for k, v in params.items():
                store["p_" + k] = v
            store["prior"] = prior
            store["g0"] = g0
            for k, v in r64.items():
                v = v[:, :n_eff, :] if k == "dh" else v
                pack("f64_" + k, v, store)
            # the fp32 reference's own distance from fp64 (its noise floor) + its loss value
            store["f32_reg_loss"] = r32["reg_loss"]
            for k in r64:
                store["ref32_err_" + k] = np.float64(relerr(r32[k], r64[k]))
            tag = f"{name}_{dec or 'funnel'}"
            np.savez_compressed(os.path.join(HERE, f"vae_{tag}.npz"), **store)
            summary.append((tag, float(r64["reg_loss"]), float(store["ref32_err_reg_loss"]), worst))
            print(f"{tag:18s} reg64={float(r64['reg_loss']):.9e} ref32_relerr(reg)={float(store['ref32_err_reg_loss']):.2e} "
                  f"oracle_vs_ref64={worst:.1e}")

    # ---- latent-given and skip_reg_loss cases (int64 zero reg_loss; autoencoders.py:107,114) ----
    name, B, S, d, n, L, cfg_id = CASES[2]
    h, params, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, "")
    vae, _ = build_reference(d, L, n, "", params, eps)
    lat_in = np.tanh(np.random.RandomState(7).standard_normal((B, n, L))).astype(np.float32)
    with torch.no_grad():
        o1 = vae(latent=torch.from_numpy(lat_in))
        o2 = vae(input_encoding=torch.from_numpy(h), skip_reg_loss=True)
    assert o1.reg_loss.dtype == torch.int64 and o2.reg_loss.dtype == torch.int64
    assert o1.reg_loss.dim() == 0 and int(o1.reg_loss) == 0 and int(o2.reg_loss) == 0
    try:
        vae()
        raise AssertionError("expected ValueError")
    except ValueError as e:
        msg = str(e)
    store = dict(h=h, latent_in=lat_in, recon_from_latent=o1.reconstructed_encoding.numpy(),
                 latent_echo=o1.latent.numpy(), skip_latent=o2.latent.numpy(),
                 skip_recon=o2.reconstructed_encoding.numpy(), value_error=np.array(msg),
                 shape=np.array([B, S, d, n, L, cfg_id], np.int64))
    for k, v in params.items():
        store["p_" + k] = v
    np.savez_compressed(os.path.join(HERE, "vae_modes_tiny.npz"), **store)

    # ---- t5_norm decoder (kept as stock HF T5LayerFF on top of our linear; autoencoders.py:42-55) ----
    h, params, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, "no_norm")
    torch.manual_seed(5)
    vae, cfg = build_reference(d, L, n, "t5_norm", params, eps)
    with injected_prior(torch.from_numpy(prior)):
        o = vae(input_encoding=torch.from_numpy(h))
    sd = {k: v.numpy() for k, v in vae.state_dict().items()}
    store = dict(h=h, prior=prior, latent=o.latent.detach().numpy(), recon=o.reconstructed_encoding.detach().numpy(),
                 reg_loss=o.reg_loss.detach().numpy(), shape=np.array([B, S, d, n, L, cfg_id], np.int64),
                 t5=np.array([cfg.t5.d_model, cfg.t5.d_ff, cfg.t5.d_kv, cfg.t5.num_heads], np.int64))
    for k, v in sd.items():
        store["sd_" + k] = v
    np.savez_compressed(os.path.join(HERE, "vae_t5norm_tiny.npz"), **store)

    # ---- MMD-only medium case, fp64, literal reference kernel (N=1024, D=77) ----
    N, D = 1024, 77
    z = np.tanh(np.random.RandomState(21).standard_normal((N, D))).astype(np.float32)
    y = np.random.RandomState(22).standard_normal((N, D)).astype(np.float32)
    zt = torch.from_numpy(z).double().requires_grad_(True)
    obj = ref.EncoderDecoderVAE(torch.nn.Identity(), torch.nn.Identity())
    mmd = obj._compute_mmd(torch.from_numpy(y).double(), zt)
    mmd.backward()
    mmd32 = obj._compute_mmd(torch.from_numpy(y), torch.from_numpy(z))
    g = orc.mmd_grad_latent_rows(z.astype(np.float64), z.astype(np.float64), y.astype(np.float64))
    assert abs(orc.mmd_gram(y.astype(np.float64), z.astype(np.float64)) - float(mmd)) < 1e-12 * abs(float(mmd)) + 1e-18
    assert relerr(g, zt.grad.numpy()) < 1e-10
    np.savez_compressed(os.path.join(HERE, "mmd_n1024_d77.npz"), z=z, y=y, mmd_f64=mmd.detach().numpy(),
                        mmd_ref32=mmd32.numpy(), dz_f64=zt.grad.numpy())
    print("mmd_n1024_d77 mmd64=%.12e ref32=%.9e" % (float(mmd), float(mmd32)))

    # ---- next-to-the-path rows: funnel upsample (model.py:265-271) and slerp (utils.py:17-34) ----
    from transformers.models.funnel.modeling_funnel import upsample as hf_upsample
    from transformer_vae.utils import slerp as ref_slerp
    up = {}
    rs = np.random.RandomState(31)
    x = rs.standard_normal((3, 5, 24)).astype(np.float32)
    g_up = {}
    cases = [(2, 9, True, False), (4, 17, True, True), (4, 12, True, True), (2, 7, False, False), (3, 40, True, False),
             (8, 16, False, False), (1, 5, True, False)]
    for ci, (stride, tlen, sep, trunc) in enumerate(cases):
        xt = torch.from_numpy(x).double().requires_grad_(True)
        o = hf_upsample(xt, stride, tlen, sep, trunc)
        cot = np.random.RandomState(40 + ci).standard_normal(tuple(o.shape))
        (o * torch.from_numpy(cot)).sum().backward()
        mine = orc.upsample(x.astype(np.float64), stride, tlen, sep, trunc)
        assert mine.shape == tuple(o.shape) and np.array_equal(mine, o.detach().numpy())
        assert np.allclose(orc.upsample_backward(cot, 5, stride, tlen, sep, trunc), xt.grad.numpy(), rtol=0, atol=1e-12)
        up[f"case{ci}_args"] = np.array([stride, tlen, int(sep), int(trunc)], np.int64)
        up[f"case{ci}_out"] = o.detach().numpy().astype(np.float32)
        up[f"case{ci}_cot"] = cot.astype(np.float32)
        up[f"case{ci}_dx"] = xt.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "upsample_cases.npz"), x=x, n_cases=np.int64(len(cases)), **up)
    t1 = rs.standard_normal((22, 77)).astype(np.float32)
    t2 = rs.standard_normal((22, 77)).astype(np.float32)
    ratios = np.repeat(np.arange(0, 1.1, 0.1, dtype=np.float32), 2)                       # trainer.py:285,293
    s_vec = ref_slerp(torch.from_numpy(ratios).double(), torch.from_numpy(t1).double(), torch.from_numpy(t2).double()).numpy()
    s_sc = ref_slerp(0.3, torch.from_numpy(t1).double(), torch.from_numpy(t2).double()).numpy()
    s32 = ref_slerp(torch.from_numpy(ratios), torch.from_numpy(t1), torch.from_numpy(t2)).numpy()
    assert relerr(orc.slerp(ratios.astype(np.float64), t1.astype(np.float64), t2.astype(np.float64)), s_vec) < 1e-12
    assert relerr(orc.slerp(0.3, t1.astype(np.float64), t2.astype(np.float64)), s_sc) < 1e-12
    np.savez_compressed(os.path.join(HERE, "slerp_cases.npz"), t1=t1, t2=t2, ratios=ratios, out_vec_f64=s_vec,
                        out_scalar03_f64=s_sc, out_vec_ref32=s32)
    print("upsample + slerp fixtures written")

    # ---- loss-weight schedule (model.py:103-109) restated; a handful of known answers ----
    steps = np.array([0, 1000, 2500, 5000, 10000])
    w = np.array([float(torch.sigmoid(torch.tensor(s * 0.0025 - 6.25)).item()) for s in steps])
    assert np.allclose(w, [orc.reg_loss_weight(int(s)) for s in steps], rtol=1e-6)
    np.savez(os.path.join(HERE, "reg_weight_schedule.npz"), steps=steps, w=w)
    print("done;", len(summary), "vae cases")


if __name__ == "__main__":
    main()
