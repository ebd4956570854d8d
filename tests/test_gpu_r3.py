"""Round-3 GPU parity additions.

* Adversarial input STRUCTURE for the split-fp16 MMD (the default for fp32 latents): the accuracy argument of the
  one-/two-product Gram is statistical (column roundings independent and zero-mean), so these feed it inputs whose
  roundings are coherent -- duplicated rows, latents quantised to a few levels, a collapsed posterior, a prior with a
  common offset, one outlier row that crushes the fp16 scale for everybody else -- against fp64 on the GPU.
  Bars: loss 1e-6, gradient 1e-5 of its maximum, and the error relative to the kernel-dependent part of the gradient
  (the exact mean-shift term removed) within 2x of the strict fp32 kernel's own.
* `_compute_kernel` as a materialised matrix and the `t5_norm` decoder's backward, against fixtures generated from the
  imported reference (tests/golden/make_golden_r3.py)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    return torch.device("cuda:0")


def _fp64_truth(z, y):
    """loss (unscaled MMD), full gradient d MMD / d z and its mean-shift part, torch fp64 on the GPU."""
    z64, y64 = z.double(), y.double()
    N, D = z64.shape
    inv = 1.0 / float(D * D)
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    tot = torch.zeros(3, dtype=torch.float64, device=z.device)
    G = torch.zeros_like(z64)
    R = torch.zeros(N, dtype=torch.float64, device=z.device)
    for r0 in range(0, N, 2048):
        sl = slice(r0, r0 + 2048)
        Ezz = torch.expm1(-((nz[sl, None] + nz[None, :] - 2.0 * (z64[sl] @ z64.T)) * inv).clamp_min(0))
        Ezy = torch.expm1(-((nz[sl, None] + ny[None, :] - 2.0 * (z64[sl] @ y64.T)) * inv).clamp_min(0))
        Eyy = torch.expm1(-((ny[sl, None] + ny[None, :] - 2.0 * (y64[sl] @ y64.T)) * inv).clamp_min(0))
        tot += torch.stack([Ezz.sum(), Ezy.sum(), Eyy.sum()])
        G[sl] = Ezz @ z64 - Ezy @ y64
        R[sl] = Ezz.sum(1) - Ezy.sum(1)
    shift = (z64.sum(0) - y64.sum(0))[None, :]
    c = 4.0 * inv / float(N * N)
    grad = c * (-(z64 * R[:, None]) + shift + G)
    return float((tot[2] + tot[0] - 2.0 * tot[1]) / float(N * N)), grad, c * shift


def _structured(case, N, D):
    gen = torch.Generator(device=dev()).manual_seed(1000 + N + D)
    z = torch.tanh(torch.randn((N, D), generator=gen, device=dev()))
    y = torch.randn((N, D), generator=gen, device=dev())
    if case == "dup64":
        z = z[: N // 64].repeat_interleave(64, 0)
        y = y[: N // 64].repeat_interleave(64, 0)
    elif case == "tern":
        z = torch.round(z)
    elif case == "lev8":
        z = torch.round(z * 4) / 4
    elif case == "collapsed":
        z = z[:1].repeat(N, 1)
    elif case == "prior+3":
        y = y + 3.0
    elif case == "outlier50":
        z = z / z.norm(dim=1, keepdim=True)
        y = y / y.norm(dim=1, keepdim=True)
        z[17] *= 50.0
    elif case == "both+5":
        z, y = z + 5.0, y + 5.0
    elif case == "both+50":
        z, y = z + 50.0, y + 50.0
    return z.contiguous(), y.contiguous()


def _run(z, y, strict):
    from transformer_vae_amd import functional as F
    zt = z.clone().requires_grad_(True)
    loss = F.mmd_reg_loss(zt, y, batch_size=1, n_latents=1, strict_fp32=strict)
    loss.backward()
    return float(loss.detach()), zt.grad.double()


# "both+5" / "both+50": every coordinate of both sets shifted -- the split modes centre on the pooled mean (mmd.hip, stage B)
CASES = ["dup64", "tern", "lev8", "collapsed", "prior+3", "outlier50", "both+5", "both+50"]


@pytest.mark.parametrize("D", [77, 256, 768, 1024])
@pytest.mark.parametrize("case", CASES)
def test_mmd_split_on_adversarially_structured_inputs(case, D):
    N = 8192
    z, y = _structured(case, N, D)
    ref_loss, ref_grad, shift_part = _fp64_truth(z, y)
    loss, grad = _run(z, y, strict=False)
    loss_s, grad_s = _run(z, y, strict=True)
    gmax = float(ref_grad.abs().max())
    kmax = float((ref_grad - shift_part).abs().max())
    err = float((grad - ref_grad).abs().max())
    err_s = float((grad_s - ref_grad).abs().max())
    info = dict(case=case, D=D, loss=loss, ref_loss=ref_loss, loss_err=abs(loss - ref_loss) / max(abs(ref_loss), 1e-300),
                grad_err=err / gmax, kern_err=err / max(kmax, 1e-300), strict_kern_err=err_s / max(kmax, 1e-300),
                strict_loss_err=abs(loss_s - ref_loss) / max(abs(ref_loss), 1e-300),
                err_row=int((grad - ref_grad).abs().max(1).values.argmax()), gmax_row=int(ref_grad.abs().max(1).values.argmax()))
    print(info)
    # a collapsed posterior against itself has MMD(z, z) terms that are exactly 0; the loss is still O(1e-3)
    assert abs(loss - ref_loss) <= 1e-6 * abs(ref_loss) + 1e-12, info
    assert err <= 1e-5 * gmax, info
    if kmax > 1e-3 * gmax:      # the finer yardstick only where the kernel-dependent part is not itself rounding noise
        # within 2x of the strict fp32 kernel's own error, or 4e-5 of the kernel-dependent part where the strict kernel is far
        # better than that (an outlier row at D = 77: fp16 pairs carry 22 bits against fp32's 24 -- measured 1.5e-5 .. 2.3e-5
        # against 1.4e-6, moving with the rounding pattern; the 1e-5-of-the-gradient bar above is the binding one)
        assert err <= max(2.0 * err_s, 4e-5 * kmax), info


def test_compute_kernel_matches_the_reference_matrix_and_gradients():
    import transformer_vae_amd as amd
    g = np.load(os.path.join(GOLD, "compute_kernel_cases.npz"))
    for ci in range(int(g["count"])):
        x = torch.from_numpy(g[f"x{ci}"]).to(dev()).requires_grad_(True)
        y = torch.from_numpy(g[f"y{ci}"]).to(dev()).requires_grad_(True)
        K = amd.EncoderDecoderVAE._compute_kernel(x, y)
        assert tuple(K.shape) == g[f"k{ci}"].shape and K.dtype == torch.float32
        (K * torch.from_numpy(g[f"c{ci}"]).to(dev())).sum().backward()
        kerr = np.abs(K.detach().cpu().numpy() - g[f"k{ci}"]).max()
        assert kerr <= max(2e-6, 2.0 * float(g[f"k32err{ci}"])), (ci, kerr)      # values in (0, 1]; the reference's own fp32 run beside it
        for got, want in ((x.grad, g[f"dx{ci}"]), (y.grad, g[f"dy{ci}"])):
            e = np.abs(got.cpu().numpy() - want).max()
            assert e <= 1e-5 * np.abs(want).max() + 1e-12, (ci, e, np.abs(want).max())
    # the instance form the reference's _compute_mmd uses, and a CPU tensor fails loudly (no fallback)
    cfg = SimpleNamespace(t5=SimpleNamespace(d_model=8), funnel=SimpleNamespace(d_model=8, layer_norm_eps=1e-9), latent_size=3,
                          n_latent_tokens=2)
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](cfg), amd.VAE_DECODER_MODELS[""](cfg))
    k2 = vae._compute_kernel(torch.from_numpy(g["x0"]).to(dev()), torch.from_numpy(g["y0"]).to(dev()))
    assert np.abs(k2.cpu().numpy() - g["k0"]).max() <= 2e-6
    with pytest.raises(RuntimeError):
        vae._compute_kernel(torch.from_numpy(g["x0"]), torch.from_numpy(g["y0"]))


def test_t5_norm_decoder_backward_through_our_linear():
    import transformer_vae_amd as amd
    from transformers import T5Config
    g = np.load(os.path.join(GOLD, "vae_t5norm_bwd.npz"))
    B, S, d, n, L, _ = (int(v) for v in g["shape"])
    dm, dff, dkv, nh = (int(v) for v in g["t5"])
    cfg = SimpleNamespace(t5=T5Config(d_model=dm, d_ff=dff, d_kv=dkv, num_heads=nh),
                          funnel=SimpleNamespace(d_model=d, layer_norm_eps=1e-9), latent_size=L, n_latent_tokens=n)
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](cfg), amd.VAE_DECODER_MODELS["t5_norm"](cfg))
    vae.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    vae = vae.to(dev())
    h = torch.from_numpy(g["h"]).to(dev()).requires_grad_(True)
    out = vae(input_encoding=h, true_samples=torch.from_numpy(g["prior"]).to(dev()))
    (out.reg_loss * float(g["reg_weight"]) + (out.reconstructed_encoding * torch.from_numpy(g["g0"]).to(dev())).sum()).backward()

    def rel(a, b):
        return float(np.abs(a.detach().cpu().numpy().astype(np.float64) - b).max() / np.abs(b).max())

    assert rel(out.latent, g["latent"]) <= 1e-5
    assert rel(out.reconstructed_encoding, g["recon"]) <= 1e-4          # stock HF block in fp32 against the fp64 fixture
    assert rel(h.grad, g["dh"]) <= 1e-4
    for name, p in vae.named_parameters():
        # every gradient passes through the stock fp32 T5 block (1e-4 class, as its forward); our linear and encoder add 1e-5
        assert rel(p.grad, g["grad_" + name]) <= 1e-4, name


# ---- the three-stage (row-sharded) evaluation through the C ABI -------------------------------------------------------
def _staged_ranks(z, y, world, code, want_grad=True):
    """What `world` ranks would run (tvae_mmd_stats_local / _prep_local / _tiles), one after the other on this GPU, with the
    collectives done by hand between the stages.  Returns (sums, dz rows, per-rank local gram copies, layout)."""
    from transformer_vae_amd import functional as F
    n_total, D = z.shape
    n_loc = n_total // world
    st = F.HipShardStages
    lay = st.layout(n_total, n_loc, D, want_grad, code)
    wss = [st.workspace(lay[0], z.device) for _ in range(world)]
    zs = [z[r * n_loc:(r + 1) * n_loc].contiguous() for r in range(world)]
    ys = [y[r * n_loc:(r + 1) * n_loc].contiguous() for r in range(world)]

    def reg(ws, off, nbytes, dt=torch.uint8):
        return ws[off:off + nbytes].view(dt)

    def combine(recs, mw):
        t = torch.stack(recs).sum(0)
        t[mw] = torch.stack(recs)[:, mw].max()
        return t

    if lay[7] >= 0:
        for r in range(world):
            st.stats(zs[r], ys[r], n_total, r * n_loc, code, want_grad, wss[r])
        tot = combine([reg(wss[r], lay[7], lay[8] * 8, torch.float64).clone() for r in range(world)], lay[8] - 2)
        for r in range(world):
            reg(wss[r], lay[7], lay[8] * 8, torch.float64).copy_(tot)
    for r in range(world):
        st.prep(zs[r], ys[r], n_total, r * n_loc, code, want_grad, wss[r])
    tot = combine([reg(wss[r], lay[10], lay[11] * 8, torch.float64).clone() for r in range(world)], lay[11] - 2)
    grams = []
    if lay[12] >= 0:
        grams = [reg(wss[r], lay[12], lay[13] * 4, torch.float32).clone() for r in range(world)]
        gsum = torch.stack(grams).sum(0)
    regions = [(lay[5], 4), (lay[6], 4), (lay[1], lay[3]), (lay[2], lay[3])]
    if lay[4] > 0:
        regions += [(lay[1] + lay[4], lay[3]), (lay[2] + lay[4], lay[3])]
    for r in range(world):
        reg(wss[r], lay[9], lay[11] * 8, torch.float64).copy_(tot)
        if lay[12] >= 0:
            reg(wss[r], lay[12], lay[13] * 4, torch.float32).copy_(gsum)
        for off, rb in regions:                      # the all-gather: every rank's own rows into everybody's workspace
            for q in range(world):
                if q != r:
                    a, b = off + q * n_loc * rb, off + (q + 1) * n_loc * rb
                    wss[r][a:b].copy_(wss[q][a:b])
    sums = torch.zeros(3, dtype=torch.float64, device=z.device)
    dzs = []
    for r in range(world):
        s3 = torch.empty(3, dtype=torch.float64, device=z.device)
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        dz = torch.empty((n_loc, D), dtype=z.dtype, device=z.device) if want_grad else None
        st.tiles(zs[r], n_total, r * n_loc, code, s3, loss, 1.0, dz, 1.0, 1, wss[r])
        st.tiles(zs[r], n_total, r * n_loc, code, s3, loss, 1.0, dz, 1.0, 2, wss[r])
        sums += s3
        dzs.append(dz)
    return sums, (torch.cat(dzs) if want_grad else None), grams, lay


@pytest.mark.parametrize("shape", [(2, 2000, 768, "f32"), (4, 4096, 768, "f32"), (2, 2048, 1024, "f32"), (2, 3000, 77, "f32"),
                                   (2, 1200, 300, "f32"), (2, 2048, 768, "bf16"), (2, 2048, 512, "strict"), (4, 1024, 64, "strict")])
def test_staged_entry_points_equal_the_one_call_form(shape):
    from transformer_vae_amd import _lib
    from transformer_vae_amd import functional as F
    world, n_total, D, kind = shape
    gen = torch.Generator(device=dev()).manual_seed(world * 1000 + D)
    z = torch.tanh(torch.randn((n_total, D), generator=gen, device=dev())) + 0.25
    y = torch.randn((n_total, D), generator=gen, device=dev())
    if kind == "bf16":
        z, y = z.bfloat16(), y.bfloat16()
    code = {"f32": _lib.TVAE_F32, "bf16": _lib.TVAE_BF16, "strict": _lib.TVAE_F32_STRICT}[kind]
    sums, dz, grams, lay = _staged_ranks(z, y, world, code)
    lib = _lib.lib()
    ws = F._workspace(lib.tvae_mmd_workspace_bytes_for(n_total, n_total, D, 1, code), dev())
    s1 = torch.empty(3, dtype=torch.float64, device=dev())
    l1 = torch.empty((), dtype=torch.float32, device=dev())
    d1 = torch.empty_like(z)
    _lib.check(lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), n_total, 0, n_total, D, code, s1.data_ptr(), l1.data_ptr(), 1.0,
                                    d1.data_ptr(), 1.0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "mmd")
    mmd = lambda s: float(s[2] + s[0] - 2.0 * s[1])
    tol = 1e-2 if kind == "bf16" else 1e-6
    assert abs(mmd(sums) - mmd(s1)) <= tol * abs(mmd(s1)), (mmd(sums), mmd(s1))
    assert float((dz.double() - d1.double()).abs().max()) <= tol * float(d1.double().abs().max())
    # against fp64 as well (the staged form is not merely self-consistent)
    ref_loss, ref_grad, _ = _fp64_truth(z.float(), y.float())
    tol64 = 1e-2 if kind == "bf16" else 1e-5
    assert abs(mmd(sums) / n_total ** 2 - ref_loss) <= tol64 * abs(ref_loss)
    c = 4.0 / (float(D) ** 2 * float(n_total) ** 2)
    assert float((dz.double() * c - ref_grad).abs().max()) <= tol64 * float(ref_grad.abs().max())
    if grams:
        # the closed-form D x D products ran over the rank's OWN rows only (K = n_local): each rank's matrix is the
        # Gram matrix of its centred rows, not of everybody's
        n_loc, ldq = n_total // world, lay[13] // (2 * D)
        mu = (z.double().sum(0) + y.double().sum(0)) / (2.0 * n_total)
        for r, g in enumerate(grams):
            zc = z[r * n_loc:(r + 1) * n_loc].double() - mu
            yc = y[r * n_loc:(r + 1) * n_loc].double() - mu
            gz = g[:D * ldq].view(D, ldq)[:, :D].double()
            gy = g[D * ldq:].view(D, ldq)[:, :D].double()
            for got, want in ((gz, zc.T @ zc), (gy, yc.T @ yc)):
                assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("n_loc,D,kind", [(752, 192, "strict"), (752, 512, "strict"), (736, 768, "f32"), (736, 1024, "f32"), (752, 77, "f32"),
                                          (736, 512, "bf16"), (100, 64, "strict"), (1000, 300, "f32")])
def test_row_blocks_of_an_equal_ragged_partition_add_up(n_loc, D, kind):
    """Three equal row blocks whose size is not a whole number of 32- / 64-row blocks while the LAST block's offset is: the
    symmetric y-y enumeration splits pairs of row blocks between the calls that own them, so all calls must decide alike
    whether to use it (until round 3 the last block did and the others did not: y-y sum off by 6e-7, tools/fuzz_gpu.py)."""
    from transformer_vae_amd import _lib
    from transformer_vae_amd import functional as F
    world, n_total = 3, 3 * n_loc
    gen = torch.Generator(device=dev()).manual_seed(n_loc + D)
    z = torch.tanh(torch.randn((n_total, D), generator=gen, device=dev())) + 0.25
    y = torch.randn((n_total, D), generator=gen, device=dev())
    if kind == "bf16":
        z, y = z.bfloat16(), y.bfloat16()
    code = {"f32": _lib.TVAE_F32, "bf16": _lib.TVAE_BF16, "strict": _lib.TVAE_F32_STRICT}[kind]
    lib = _lib.lib()

    def call(lo, rows):
        ws = F._workspace(lib.tvae_mmd_workspace_bytes_for(n_total, rows, D, 0, code), dev())
        s3 = torch.empty(3, dtype=torch.float64, device=dev())
        _lib.check(lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), n_total, lo, rows, D, code, s3.data_ptr(), None, 1.0, None, 1.0,
                                        ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "mmd")
        return s3

    whole = call(0, n_total)
    parts = sum(call(r * n_loc, n_loc) for r in range(world))
    z64, y64 = z.double(), y.double()
    inv = 1.0 / float(D * D)
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    E = lambda a, na, b, nb: torch.expm1(-((na[:, None] + nb[None, :] - 2.0 * (a @ b.T)) * inv).clamp_min(0)).sum()
    ref = torch.stack([E(z64, nz, z64, nz), E(z64, nz, y64, ny), E(y64, ny, y64, ny)])
    tol = 1e-3 if kind == "bf16" else 2e-8          # of each sum (the MMD is a 1e-3-class difference of them)
    for got in (whole, parts):
        assert float(((got - ref).abs() / ref.abs()).max()) <= tol, (got.tolist(), ref.tolist())


@pytest.mark.parametrize("world,n_loc,D,grad", [(1, 32837, 768, True), (1, 16500, 384, False), (2, 16512, 512, True), (2, 16500, 512, True),
                                                (2, 16448, 640, True), (1, 33000, 256, False),
                                                # D > 768: the same kernel on two step buffers, beside the 32-row gradient launch
                                                (1, 16517, 1024, True), (1, 16500, 896, False), (2, 16512, 1024, True), (2, 16480, 1024, True)])
def test_sum_only_launches_on_128_row_workgroups(world, n_loc, D, grad):
    """Sizes at which the sum-only launches of the split mode run mmd_q32_sum3_kernel (128 rows per workgroup, each wave
    its 32 rows against the whole step): ragged ends, the two-problem launch of a call without gradient, row blocks that are whole
    128-row blocks (symmetric enumeration over 128-point blocks), ragged equal blocks (plain enumeration), and blocks that are
    whole 64- but not 128-row blocks (the 64-row kernel keeps them).  The three sums against torch fp64."""
    from transformer_vae_amd import _lib
    from transformer_vae_amd import functional as F
    n_total = world * n_loc
    gen = torch.Generator(device=dev()).manual_seed(world + n_loc + D)
    z = torch.tanh(torch.randn((n_total, D), generator=gen, device=dev())) * 0.8 + 0.1
    y = torch.randn((n_total, D), generator=gen, device=dev())
    lib = _lib.lib()
    parts = torch.zeros(3, dtype=torch.float64, device=dev())
    for r in range(world):
        ws = F._workspace(lib.tvae_mmd_workspace_bytes_for(n_total, n_loc, D, int(grad), _lib.TVAE_F32), dev())
        s3 = torch.empty(3, dtype=torch.float64, device=dev())
        dz = torch.empty((n_loc, D), device=dev()) if grad else None
        _lib.check(lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), n_total, r * n_loc, n_loc, D, _lib.TVAE_F32, s3.data_ptr(), None, 1.0,
                                        dz.data_ptr() if grad else None, 1.0, ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream().cuda_stream), "mmd")
        parts += s3
        del ws
    z64, y64 = z.double(), y.double()
    inv = 1.0 / float(D * D)
    nz, ny = (z64 * z64).sum(1), (y64 * y64).sum(1)
    ref = torch.zeros(3, dtype=torch.float64, device=dev())
    for r0 in range(0, n_total, 4096):
        sl = slice(r0, r0 + 4096)
        E = lambda a, na, b, nb: torch.expm1(-((na[:, None] + nb[None, :] - 2.0 * (a @ b.T)) * inv).clamp_min(0)).sum()
        ref += torch.stack([E(z64[sl], nz[sl], z64, nz), E(z64[sl], nz[sl], y64, ny), E(y64[sl], ny[sl], y64, ny)])
    err = ((parts - ref).abs() / ref.abs()).tolist()
    assert max(err) <= 2e-8, (err, parts.tolist(), ref.tolist())
    mmd, mmd_ref = float(parts[2] + parts[0] - 2.0 * parts[1]), float(ref[2] + ref[0] - 2.0 * ref[1])
    assert abs(mmd - mmd_ref) <= 1e-6 * abs(mmd_ref), (mmd, mmd_ref)
