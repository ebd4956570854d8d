"""CPU: the numpy oracle and the literal torch port against the committed reference vectors.

tests/golden/*.npz hold outputs of the real reference (see tests/golden/make_golden.py).
Tolerances: the oracle runs in fp64 -> 1e-9 relative against the fp64 reference (big tensors
are stored as float32, so those compare at 2e-7)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import bottleneck_oracle as orc
from oracle import literal_torch as lit

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VAE_FILES = sorted(glob.glob(os.path.join(GOLD, "vae_*_funnel.npz")) + glob.glob(os.path.join(GOLD, "vae_*_no_norm.npz")))


def load_case(path):
    g = np.load(path)
    B, S, d, n, L, cfg_id = (int(v) for v in g["shape"])
    dec = str(g["decoder"])
    h, params, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, dec)
    chk = np.array([h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum()])
    assert np.allclose(chk, g["h_checksum"], rtol=0, atol=0), "RandomState stream changed"
    for k, v in params.items():
        assert np.array_equal(v, g["p_" + k])
    assert np.array_equal(prior, g["prior"]) and np.array_equal(g0, g["g0"])
    return g, (B, S, d, n, L), h, params, prior, g0


def tol_for(arr):
    return 2e-7 if arr.dtype == np.float32 else 1e-9


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("path", VAE_FILES, ids=[os.path.basename(p)[4:-4] for p in VAE_FILES])
def test_oracle_matches_reference_fp64(path):
    g, (B, S, d, n, L), h, params, prior, g0 = load_case(path)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    h64 = h.astype(np.float64)
    fwd = orc.bottleneck_forward(h64, p64, n, prior.astype(np.float64), float(g["eps"]))
    bwd = orc.bottleneck_backward(h64, p64, n, fwd, g0.astype(np.float64), float(g["reg_weight"]),
                                  prior.astype(np.float64))
    n_eff = min(n, S)
    assert fwd["latent"].shape == (B, n_eff, L)
    assert abs(fwd["reg_loss"] - float(g["f64_reg_loss"])) <= 1e-9 * abs(float(g["f64_reg_loss"]))
    got = dict(latent=fwd["latent"], recon=fwd["recon"], dh=bwd["h"][:, :n_eff], dWe=bwd["We"], dbe=bwd["be"])
    for k in ("Wd", "bd", "gamma", "beta"):
        if k in bwd:
            got["d" + k] = bwd[k]
    for k, v in got.items():
        ref = g["f64_" + k]
        assert relerr(v, ref) <= tol_for(ref), k
    assert np.all(bwd["h"][:, n_eff:] == 0)


@pytest.mark.parametrize("name", ["tiny_funnel", "ragged_no_norm", "n_gt_s_funnel", "ident_funnel"])
def test_literal_forms_agree(name):
    """literal (tile/sub/square/mean/exp) == Gram/expm1 form, numpy fp64 and torch fp32 port."""
    g, (B, S, d, n, L), h, params, prior, g0 = load_case(os.path.join(GOLD, f"vae_{name}.npz"))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    a = orc.bottleneck_forward(h.astype(np.float64), p64, n, prior.astype(np.float64), literal=True)
    b = orc.bottleneck_forward(h.astype(np.float64), p64, n, prior.astype(np.float64), literal=False)
    assert abs(a["reg_loss"] - b["reg_loss"]) <= 1e-10 * abs(a["reg_loss"])
    assert relerr(a["latent"], b["latent"]) < 1e-14
    pt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    ht = torch.from_numpy(h).clone().requires_grad_(True)
    lit.step_literal(ht, pt, n, torch.from_numpy(prior), torch.from_numpy(g0), float(g["reg_weight"]))
    assert relerr(ht.grad.numpy()[:, :min(n, S)], g["f64_dh"]) < 5e-5
    _, _, reg = lit.forward_literal(ht, pt, n, torch.from_numpy(prior))
    assert abs(float(reg.detach()) - float(g["f32_reg_loss"])) <= 1e-6 * abs(float(g["f32_reg_loss"]))


def test_mmd_medium_fp64():
    g = np.load(os.path.join(GOLD, "mmd_n1024_d77.npz"))
    z, y = g["z"].astype(np.float64), g["y"].astype(np.float64)
    assert abs(orc.mmd_gram(y, z) - float(g["mmd_f64"])) <= 1e-11 * abs(float(g["mmd_f64"]))
    dz = orc.mmd_grad_latent_rows(z, z, y, chunk=300)
    assert relerr(dz, g["dz_f64"]) < 1e-10
    # row-sharded partial sums add up to the global sums (what each rank would compute)
    tot = np.zeros(3)
    for r in range(4):
        sl = slice(r * 256, (r + 1) * 256)
        tot += np.array(orc.mmd_partial_sums(z[sl], y[sl], z, y))
    n = z.shape[0]
    mmd = (tot[2] + tot[0] - 2 * tot[1]) / (n * n)
    assert abs(mmd - float(g["mmd_f64"])) <= 1e-11 * abs(float(g["mmd_f64"]))


def test_mmd_fp32_gram_form_is_at_least_as_accurate_as_reference_fp32():
    """The expm1/Gram formulation in fp32 must not be further from fp64 than the reference's own fp32."""
    g = np.load(os.path.join(GOLD, "mmd_n1024_d77.npz"))
    ref_err = abs(float(g["mmd_ref32"]) - float(g["mmd_f64"]))
    ours = orc.mmd_gram(g["y"], g["z"])
    assert abs(ours - float(g["mmd_f64"])) <= ref_err


def test_modes_and_schedule():
    g = np.load(os.path.join(GOLD, "vae_modes_tiny.npz"))
    B, S, d, n, L, _ = (int(v) for v in g["shape"])
    params = {k[2:]: g[k].astype(np.float64) for k in g.files if k.startswith("p_")}
    out = orc.bottleneck_forward(g["h"].astype(np.float64), params, n, None, use_reg_loss=False)
    assert out["reg_loss"] == 0
    assert relerr(out["latent"], g["skip_latent"]) < 1e-6 and relerr(out["recon"], g["skip_recon"]) < 1e-5
    r = orc.latent_decode(g["latent_in"].astype(np.float64), params["Wd"], params["bd"])
    o, _, _ = orc.layer_norm(r, params["gamma"], params["beta"], 1e-9)
    assert relerr(o, g["recon_from_latent"]) < 1e-5
    s = np.load(os.path.join(GOLD, "reg_weight_schedule.npz"))
    for step, w in zip(s["steps"], s["w"]):
        assert abs(orc.reg_loss_weight(int(step)) - w) <= 1e-6 * w
    assert orc.combine_loss(2.0, 0.5, 0.25) == 2.125
    assert orc.trainer_reduce([1.0, 3.0], 2) == 1.0


def test_upsample_and_slerp_oracle_vs_reference_vectors():
    """HF funnel upsample (model.py:265-271) and slerp (utils.py:17-34)."""
    g = np.load(os.path.join(GOLD, "upsample_cases.npz"))
    x = g["x"].astype(np.float64)
    for ci in range(int(g["n_cases"])):
        stride, tlen, sep, trunc = (int(v) for v in g[f"case{ci}_args"])
        out = orc.upsample(x, stride, tlen, bool(sep), bool(trunc))
        assert out.shape == g[f"case{ci}_out"].shape and np.array_equal(out.astype(np.float32), g[f"case{ci}_out"])
        dx = orc.upsample_backward(g[f"case{ci}_cot"].astype(np.float64), x.shape[1], stride, tlen, bool(sep), bool(trunc))
        assert relerr(dx, g[f"case{ci}_dx"]) < 1e-6
    s = np.load(os.path.join(GOLD, "slerp_cases.npz"))
    t1, t2 = s["t1"].astype(np.float64), s["t2"].astype(np.float64)
    assert relerr(orc.slerp(s["ratios"].astype(np.float64), t1, t2), s["out_vec_f64"]) < 1e-12
    assert relerr(orc.slerp(0.3, t1, t2), s["out_scalar03_f64"]) < 1e-12
