"""CPU oracle for the latent-bottleneck + MMD hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the arithmetic in the reference's
``transformer_vae/autoencoders.py`` (citations are file:line under the reference
checkout).  It exists to *check* the HIP path; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product package (``transformer_vae_amd``) never does, and fails loudly when
the HIP library is missing.

Pinning: the reference has no numeric assertions of its own for this path
(SURVEY.md section 4), so the pin is ``tests/golden/*.npz`` — outputs of the real
reference imported from its checkout by ``tests/golden/make_golden.py`` (committed),
which also asserts this oracle against the import while it generates them.

Two formulations are kept side by side:
  * ``*_literal``  — op for op what the reference executes (tile, subtract, square,
    mean, /dim, exp, mean).  O(N^2 D) memory.  Used for small cases and for the
    literal CPU baseline.
  * Gram / expm1 form with an analytic gradient — O(N D) memory, row-chunked, runs
    at every BASELINE size.  Verified equal to the literal form in fp64.
All functions work in the dtype of their inputs (use float64 for ground truth).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- #
# a1  LatentEncoderNTokens            autoencoders.py:15-22
# --------------------------------------------------------------------------- #

def latent_encode_literal(h, We, be, n_tokens):
    """tanh(Linear(h)) over ALL S tokens, then keep the first n (autoencoders.py:22)."""
    pre = h @ We.T + be
    return np.tanh(pre)[:, :n_tokens, :]


def latent_encode(h, We, be, n_tokens):
    """Slice first, then project: identical values, S/n times less work.

    n_tokens > S silently yields S tokens, as the reference's slice does
    (autoencoders.py:22)."""
    hs = h[:, :n_tokens, :]
    return np.tanh(hs @ We.T + be)


# --------------------------------------------------------------------------- #
# a2/a3  LatentDecoderNTokens / LatentDecoderFunnelNorm   autoencoders.py:29-39, 58-67
# --------------------------------------------------------------------------- #

def latent_decode(z, Wd=None, bd=None):
    """Linear latent_size -> d_model; identity when the sizes are equal
    (autoencoders.py:32-36: no parameters exist then)."""
    if Wd is None:
        return z
    return z @ Wd.T + bd


def layer_norm(r, gamma, beta, eps):
    """nn.LayerNorm over the last axis, biased variance (autoencoders.py:64-67)."""
    mu = r.mean(axis=-1, keepdims=True)
    xc = r - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = xc * rstd
    return xhat * gamma + beta, xhat, rstd


# --------------------------------------------------------------------------- #
# a7/a8/a9  _compute_kernel / _compute_mmd / _regularliser_loss  autoencoders.py:117-136
# --------------------------------------------------------------------------- #

def rbf_kernel_literal(x, y):
    """K[i,j] = exp(-mean_k((x_i-y_j)^2) / dim)   (autoencoders.py:118-126).

    Both inputs are tiled to (Nx, Ny, D) exactly as the reference does."""
    nx, dim = x.shape
    ny = y.shape[0]
    tx = np.repeat(x.reshape(nx, 1, dim), ny, axis=1)
    ty = np.repeat(y.reshape(1, ny, dim), nx, axis=0)
    return np.exp(-((tx - ty) ** 2).mean(axis=2) / dim)


def mmd_literal(x, y):
    """mean(Kxx) + mean(Kyy) - 2 mean(Kxy)   (autoencoders.py:128-132)."""
    return rbf_kernel_literal(x, x).mean() + rbf_kernel_literal(y, y).mean() \
        - 2.0 * rbf_kernel_literal(x, y).mean()


def _em1_block(a, b, na, nb, inv_d2):
    """expm1(-t) with t = ||a_i - b_j||^2 / D^2 in Gram form, clamped at t >= 0."""
    t = (na[:, None] + nb[None, :] - 2.0 * (a @ b.T)) * inv_d2
    np.maximum(t, 0, out=t)
    return np.expm1(-t)


def mmd_partial_sums(z_rows, y_rows, z_all, y_all, chunk=1024):
    """Sums of (K-1) over row-block x all columns for the three kernels.

    Returns (S_zz, S_zy, S_yy) as float64.  Summing the results over a partition of
    the rows gives the global sums; the '+1's of K = 1 + expm1(-t) cancel exactly in
    mean(Kyy)+mean(Kzz)-2mean(Kzy) because every mean has weight one."""
    dim = z_all.shape[1]
    inv_d2 = z_all.dtype.type(1.0) / z_all.dtype.type(dim * dim)
    nz_all = (z_all * z_all).sum(1)
    ny_all = (y_all * y_all).sum(1)
    s_zz = s_zy = s_yy = 0.0
    for r0 in range(0, z_rows.shape[0], chunk):
        zr = z_rows[r0:r0 + chunk]
        nzr = (zr * zr).sum(1)
        s_zz += float(_em1_block(zr, z_all, nzr, nz_all, inv_d2).sum(dtype=np.float64))
        s_zy += float(_em1_block(zr, y_all, nzr, ny_all, inv_d2).sum(dtype=np.float64))
    for r0 in range(0, y_rows.shape[0], chunk):
        yr = y_rows[r0:r0 + chunk]
        nyr = (yr * yr).sum(1)
        s_yy += float(_em1_block(yr, y_all, nyr, ny_all, inv_d2).sum(dtype=np.float64))
    return s_zz, s_zy, s_yy


def mmd_gram(prior, latent, chunk=1024):
    """Same value as ``mmd_literal(prior, latent)`` without (N,N,D) temporaries."""
    s_zz, s_zy, s_yy = mmd_partial_sums(latent, prior, latent, prior, chunk)
    nz, ny = latent.shape[0], prior.shape[0]
    return s_yy / (ny * ny) + s_zz / (nz * nz) - 2.0 * s_zy / (nz * ny)


def mmd_grad_latent_rows(z_rows, z_all, y_all, chunk=1024):
    """d MMD / d z_k for the given rows (prior and latent have equal counts N).

    Closed form (derived from autoencoders.py:126-132):
      (4 / (N^2 D^2)) [ sum_i K(y_i,z_k)(z_k - y_i) - sum_j K(z_k,z_j)(z_k - z_j) ]
    written with E = K - 1 so that no O(1) terms have to cancel numerically."""
    n, dim = z_all.shape
    assert y_all.shape == z_all.shape
    inv_d2 = z_all.dtype.type(1.0) / z_all.dtype.type(dim * dim)
    nz_all = (z_all * z_all).sum(1)
    ny_all = (y_all * y_all).sum(1)
    mean_shift = z_all.sum(0, dtype=np.float64) - y_all.sum(0, dtype=np.float64)
    out = np.empty_like(z_rows)
    for r0 in range(0, z_rows.shape[0], chunk):
        zr = z_rows[r0:r0 + chunk]
        nzr = (zr * zr).sum(1)
        ezz = _em1_block(zr, z_all, nzr, nz_all, inv_d2)
        ezy = _em1_block(zr, y_all, nzr, ny_all, inv_d2)
        rs = ezz.sum(1, dtype=np.float64) - ezy.sum(1, dtype=np.float64)
        g = ezz @ z_all - ezy @ y_all
        out[r0:r0 + chunk] = (-(zr * rs[:, None]) + mean_shift[None, :] + g) * (4.0 * inv_d2 / (n * n))
    return out


# --------------------------------------------------------------------------- #
# a6  EncoderDecoderVAE.forward + analytic backward     autoencoders.py:99-115
# --------------------------------------------------------------------------- #

def bottleneck_forward(h, params, n_tokens, prior=None, eps=1e-9, use_reg_loss=True,
                       literal=False):
    """h (B,S,d) -> dict(latent, recon, reg_loss, ...).

    params: We (L,d), be (L,), optional Wd (d,L), bd (d,), optional gamma/beta (d,).
    reg_loss = MMD(prior, latent.reshape(-1, L)) / B * n   (autoencoders.py:111-112:
    left-to-right precedence, i.e. MMD * n / B)."""
    enc = latent_encode_literal if literal else latent_encode
    z = enc(h, params["We"], params["be"], n_tokens)
    r = latent_decode(z, params.get("Wd"), params.get("bd"))
    out = dict(latent=z, pre_norm=r)
    if params.get("gamma") is not None:
        o, xhat, rstd = layer_norm(r, params["gamma"], params["beta"], eps)
        out.update(recon=o, xhat=xhat, rstd=rstd)
    else:
        out["recon"] = r
    if use_reg_loss and prior is not None:
        bsz, n_eff, lat = z.shape
        z2 = z.reshape(-1, lat)
        mmd = mmd_literal(prior, z2) if literal else mmd_gram(prior, z2)
        out["reg_loss"] = mmd / bsz * n_eff
    else:
        out["reg_loss"] = 0  # the reference returns an int64 zero tensor here (:114)
    return out


def bottleneck_backward(h, params, n_tokens, fwd, g_recon, reg_weight=0.0, prior=None,
                        g_latent=None):
    """Gradients of  reg_weight*reg_loss + <recon, g_recon> (+ <latent, g_latent>)."""
    z = fwd["latent"]
    bsz, n_eff, lat = z.shape
    d_model = h.shape[2]
    z2 = z.reshape(-1, lat)
    grads = {}
    dout = g_recon.reshape(-1, g_recon.shape[-1])
    if params.get("gamma") is not None:
        xhat = fwd["xhat"].reshape(dout.shape)
        rstd = fwd["rstd"].reshape(-1, 1)
        grads["gamma"] = (dout * xhat).sum(0)
        grads["beta"] = dout.sum(0)
        dxh = dout * params["gamma"]
        dr = rstd * (dxh - dxh.mean(1, keepdims=True) - xhat * (dxh * xhat).mean(1, keepdims=True))
    else:
        dr = dout
    if params.get("Wd") is not None:
        grads["Wd"] = dr.T @ z2
        grads["bd"] = dr.sum(0)
        dz = dr @ params["Wd"]
    else:
        dz = dr.copy()
    if prior is not None and reg_weight != 0.0:
        dz = dz + reg_weight * (n_eff / bsz) * mmd_grad_latent_rows(z2, z2, prior)
    if g_latent is not None:
        dz = dz + g_latent.reshape(dz.shape)
    dpre = dz * (1.0 - z2 * z2)
    hs = h[:, :n_eff, :].reshape(-1, d_model)
    grads["We"] = dpre.T @ hs
    grads["be"] = dpre.sum(0)
    dh = np.zeros_like(h)
    dh[:, :n_eff, :] = (dpre @ params["We"]).reshape(bsz, n_eff, d_model)
    grads["h"] = dh
    grads["latent_total"] = dz.reshape(z.shape)
    return grads


# --------------------------------------------------------------------------- #
# a11/a12  loss weighting and trainer reduction     model.py:103-109,305-306; trainer.py:258-281
# --------------------------------------------------------------------------- #

def reg_loss_weight(global_step, k=0.0025, b=6.25):
    """sigmoid(step*k - b) as a python float (model.py:103-109; defaults config.py:83-84)."""
    return float(1.0 / (1.0 + np.exp(-(global_step * k - b))))


def combine_loss(decoder_ce, reg_loss, w):
    """loss = decoder_ce + reg_loss * w   (model.py:305-306)."""
    return decoder_ce + reg_loss * w


def trainer_reduce(loss_per_replica, grad_accum_steps=1):
    """mean over DataParallel replicas, then / grad_accum (trainer.py:265-269)."""
    loss = float(np.mean(loss_per_replica))
    return loss / grad_accum_steps if grad_accum_steps > 1 else loss


# --------------------------------------------------------------------------- #
# f1  upsample of reconstructed_encoding        model.py:265-271 (HF modeling_funnel.upsample)
# f4  slerp of latent rows                      utils.py:17-34
# --------------------------------------------------------------------------- #

def upsample(x, stride, target_len, separate_cls=True, truncate_seq=False):
    """Repeat tokens `stride` times along axis 1 (the first one stays single when separate_cls,
    truncate_seq appends stride-1 zero rows before the cut) and keep target_len positions."""
    if stride == 1:
        return x
    if separate_cls:
        body = np.repeat(x[:, 1:], stride, axis=1)
        if truncate_seq:
            body = np.concatenate([body, np.zeros((x.shape[0], stride - 1, x.shape[2]), x.dtype)], axis=1)
        return np.concatenate([x[:, :1], body[:, :max(target_len - 1, 0)]], axis=1)
    return np.repeat(x, stride, axis=1)[:, :target_len]


def upsample_backward(g, n, stride, target_len, separate_cls=True, truncate_seq=False):
    """Adjoint of `upsample`: every input row collects the cotangents of its copies."""
    if stride == 1:
        return g
    B, out_len, d = g.shape
    dx = np.zeros((B, n, d), g.dtype)
    for j in range(out_len):
        if separate_cls:
            src = 0 if j == 0 else 1 + (j - 1) // stride
            if src > n - 1:
                continue            # zero padding row
        else:
            src = j // stride
        dx[:, src] += g[:, j]
    return dx


def slerp(ratio, t1, t2):
    """sin((1-r)w)/sin(w) t1 + sin(r w)/sin(w) t2, w = acos(<t1/|t1|, t2/|t2|>) per row (utils.py:29-33)."""
    lo = t1 / np.linalg.norm(t1, axis=1, keepdims=True)
    hi = t2 / np.linalg.norm(t2, axis=1, keepdims=True)
    omega = np.arccos((lo * hi).sum(1))
    so = np.sin(omega)
    ratio = np.asarray(ratio, dtype=t1.dtype)
    return (np.sin((1.0 - ratio) * omega) / so)[:, None] * t1 + (np.sin(ratio * omega) / so)[:, None] * t2


# --------------------------------------------------------------------------- #
# synthetic inputs shared by tests, bench and golden generation
# --------------------------------------------------------------------------- #

def synth_inputs(B, S, d, n, L, cfg_id=0, decoder="", dtype=np.float32):
    """Deterministic inputs from numpy's frozen RandomState streams.

    h ~ N(0,1) seed 1234+cfg; nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights seed 0;
    prior ~ N(0,1) seed 4321+cfg; recon cotangent G0 ~ N(0,1)/sqrt(N d) seed 99."""
    n_eff = min(n, S)
    N = B * n_eff
    h = np.random.RandomState(1234 + cfg_id).standard_normal((B, S, d)).astype(dtype)
    rs = np.random.RandomState(0)
    p = {}
    k = 1.0 / np.sqrt(d)
    p["We"] = rs.uniform(-k, k, (L, d)).astype(dtype)
    p["be"] = rs.uniform(-k, k, (L,)).astype(dtype)
    if L != d:
        k = 1.0 / np.sqrt(L)
        p["Wd"] = rs.uniform(-k, k, (d, L)).astype(dtype)
        p["bd"] = rs.uniform(-k, k, (d,)).astype(dtype)
    if decoder == "":
        p["gamma"] = (1.0 + 0.1 * rs.standard_normal(d)).astype(dtype)
        p["beta"] = (0.1 * rs.standard_normal(d)).astype(dtype)
    prior = np.random.RandomState(4321 + cfg_id).standard_normal((N, L)).astype(dtype)
    g0 = (np.random.RandomState(99).standard_normal((B, n_eff, d)) / np.sqrt(N * d)).astype(dtype)
    return h, p, prior, g0
