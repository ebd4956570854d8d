"""BASELINE.json's configurations at FULL size, as whole steps through the module interface.

cfg3 (B 1024, S 237, d 768, n 5, L 77), cfg4 (global batch 8192, S = n = 30, d 1024, L 768: N = 245 760 MMD rows) and
cfg5 (global batch 65 536, S = n = 1, d = L = 1024: identity decoder + LayerNorm) run `out = vae(h, true_samples=...)`,
`loss = 0.5 * reg_loss + <recon, G0>`, `loss.backward()` and are checked against

  * the oracle (oracle/bottleneck_oracle.py, fp64) on a SAMPLE of whole sequences: latent, reconstruction and dh rows,
    with the regulariser's share from `mmd_grad_latent_rows` of those rows against ALL N columns -- 1e-5 relative;
  * a chunked fp64 evaluation of the MMD on the GPU (torch) for the scalar loss at sizes the CPU cannot reach
    (N^2 kernel values) -- 1e-6 relative.

Everything else about these shapes (all rows of the projections and of the weight gradients) is
test_gpu_parity.py::test_full_size_projections_against_fp64."""
import numpy as np
import pytest
import torch

from oracle import bottleneck_oracle as orc
from test_gpu_parity import F32_TOL, _mmd_fp64_on_gpu, build_vae, dev, relerr

pytestmark = pytest.mark.gpu


def _whole_step(B, S, d, n, L, cfg_id, seed, n_seq_sample, reg_w=0.5):
    _, params, _, _ = orc.synth_inputs(2, 2, d, n, L, cfg_id, "")
    vae = build_vae(d, L, n, "", params)
    n_eff = min(n, S)
    N = B * n_eff
    gen = torch.Generator(device=dev()).manual_seed(seed)
    h = torch.randn((B, S, d), generator=gen, device=dev()).requires_grad_(True)
    prior = torch.randn((N, L), generator=gen, device=dev())
    g0 = torch.randn((B, n_eff, d), generator=gen, device=dev()) / float(N * d) ** 0.5
    out = vae(input_encoding=h, true_samples=prior)
    (out.reg_loss * reg_w + (out.reconstructed_encoding * g0).sum()).backward()
    torch.cuda.synchronize()

    # ---- scalar loss: chunked fp64 on the GPU, from the latents the kernels produced
    z_all = out.latent.detach().reshape(N, L)
    ref_loss = _mmd_fp64_on_gpu(z_all, prior, chunk=2048) * n_eff / B
    got_loss = float(out.reg_loss.detach())
    assert abs(got_loss - ref_loss) <= 1e-6 * abs(ref_loss), (got_loss, ref_loss)

    # ---- latents of every row against an fp64 encode (torch, on the GPU)
    We = vae.encoder.token_to_latent.weight.detach().double()
    be = vae.encoder.token_to_latent.bias.detach().double()
    z64_all = torch.empty((N, L), dtype=torch.float64, device=dev())
    hs = h.detach()[:, :n_eff]
    step = max(1, 8192 // n_eff)
    for b0 in range(0, B, step):
        z64_all[b0 * n_eff:(b0 + step) * n_eff] = torch.tanh(hs[b0:b0 + step].double().reshape(-1, d) @ We.T + be)
    assert float((z_all.double() - z64_all).abs().max() / z64_all.abs().max()) <= F32_TOL
    assert float(h.grad[:, n_eff:].abs().max()) == 0.0 if S > n_eff else True

    # ---- a sample of whole sequences through the oracle in fp64
    rs = np.random.RandomState(seed)
    seqs = np.unique(np.concatenate([[0, 1, B // 2, B - 2, B - 1], rs.randint(0, B, n_seq_sample)]))
    rows = (seqs[:, None] * n_eff + np.arange(n_eff)[None, :]).reshape(-1)
    assert rows.size >= 200
    idx = torch.from_numpy(seqs).to(dev())
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    h_sub = h.detach()[idx].double().cpu().numpy()
    fwd = orc.bottleneck_forward(h_sub, p64, n, None, vae.decoder.norm.eps)
    z64, y64 = z64_all.cpu().numpy(), prior.double().cpu().numpy()
    g_mmd = orc.mmd_grad_latent_rows(z64[rows], z64, y64) * (n_eff / B)          # d reg_loss / d latent rows
    bwd = orc.bottleneck_backward(h_sub, p64, n, fwd, g0[idx].double().cpu().numpy(), 0.0, None,
                                  g_latent=reg_w * g_mmd)
    assert relerr(out.latent.detach()[idx].cpu().numpy(), fwd["latent"]) <= F32_TOL
    assert relerr(out.reconstructed_encoding.detach()[idx].cpu().numpy(), fwd["recon"]) <= F32_TOL
    assert relerr(h.grad[idx].cpu().numpy(), bwd["h"]) <= F32_TOL
    # the regulariser's share alone is orders of magnitude below the decoder's at these sizes: check it by itself too
    h.grad = None
    for p in vae.parameters():
        p.grad = None
    out2 = vae(input_encoding=h, true_samples=prior)
    out2.reg_loss.backward()
    bwd_reg = orc.bottleneck_backward(h_sub, p64, n, fwd, np.zeros_like(fwd["recon"]), 0.0, None, g_latent=g_mmd)
    assert relerr(h.grad[idx].cpu().numpy(), bwd_reg["h"]) <= F32_TOL
    return got_loss


def test_cfg4_whole_step_full_size():
    """large_model (d_model 1024) full_30_tkn, global batch 8192 on one GPU: N = 245 760, D = 768."""
    _whole_step(8192, 30, 1024, 30, 768, 4, 20244, 4)


def test_cfg5_whole_step_full_size():
    """MMD stress: global batch 65 536, latent dim 1024 = d_model (identity decoder + LayerNorm), one token."""
    _whole_step(65536, 1, 1024, 1, 1024, 5, 20245, 240)


def test_cfg3_whole_step_full_size():
    """mnist_base funnel_small window200: B 1024, 237-token hidden states, 5 latent tokens of 77."""
    _whole_step(1024, 237, 768, 5, 77, 3, 20243, 48)


@pytest.mark.parametrize("strict", [True, False])
def test_graph_replay_of_a_large_mmd_call(strict):
    """A captured MMD call at N = 32 768, D = 768.  strict: the fp32-input kernel, whose gradient launch runs its
    per-XCD soft rendezvous at this size (arrival counters zeroed by a memset node inside the graph, three epochs).
    Default: the split-fp16 kernels, whose operand scale is recomputed from the inputs on every replay.  Replays with
    new contents must reproduce the eager results bit for bit, three times in a row."""
    import transformer_vae_amd as amd
    from transformer_vae_amd import functional as F
    N, D = 32768, 768
    gen = torch.Generator(device=dev()).manual_seed(99)

    def draw():
        return (torch.tanh(torch.randn((N, D), generator=gen, device=dev())), torch.randn((N, D), generator=gen, device=dev()))

    z_buf, y_buf = draw()
    z_buf.requires_grad_(True)

    def step(z, y):
        loss = F.mmd_reg_loss(z, y, batch_size=1, n_latents=1, strict_fp32=strict)
        loss.backward()
        return loss

    gs = amd.GraphedStep(step, (z_buf, y_buf), zero_grads=[z_buf])
    for _ in range(3):
        z_new, y_new = draw()
        loss = gs(z_new, y_new)
        torch.cuda.synchronize()
        zt = z_new.clone().requires_grad_(True)
        want = F.mmd_reg_loss(zt, y_new, batch_size=1, n_latents=1, strict_fp32=strict)
        want.backward()
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), want.detach())
        assert torch.equal(z_buf.grad, zt.grad)
    rows = np.arange(0, N, 997)
    gref = orc.mmd_grad_latent_rows(z_new.double().cpu().numpy()[rows], z_new.double().cpu().numpy(),
                                    y_new.double().cpu().numpy())
    assert relerr(z_buf.grad.cpu().numpy()[rows], gref) <= F32_TOL


def test_graph_replay_switches_gradient_form_on_the_device():
    """The Q-form / E-form choice of the split kernels (192 < D <= 1024) is made on the device from the inputs' norms, so
    ONE captured graph serves both: replays alternate between ordinary priors (Q-form) and priors scaled so that
    4 |y|^2 / D^2 passes 1/16 (E-form) and must reproduce the eager results bit for bit, and the oracle's to 1e-5."""
    import transformer_vae_amd as amd
    from transformer_vae_amd import functional as F
    N, D = 4096, 512
    gen = torch.Generator(device=dev()).manual_seed(7)

    def draw(scale):
        return (torch.tanh(torch.randn((N, D), generator=gen, device=dev())),
                torch.randn((N, D), generator=gen, device=dev()) * scale)

    z_buf, y_buf = draw(1.0)
    z_buf.requires_grad_(True)

    def step(z, y):
        loss = F.mmd_reg_loss(z, y, batch_size=1, n_latents=1)
        loss.backward()
        return loss

    gs = amd.GraphedStep(step, (z_buf, y_buf), zero_grads=[z_buf])
    rows = np.arange(0, N, 97)
    for scale in (1.0, 5.0, 1.0, 5.0):
        z_new, y_new = draw(scale)
        loss = gs(z_new, y_new)
        torch.cuda.synchronize()
        zt = z_new.clone().requires_grad_(True)
        want = F.mmd_reg_loss(zt, y_new, batch_size=1, n_latents=1)
        want.backward()
        torch.cuda.synchronize()
        assert torch.equal(loss.detach(), want.detach())
        assert torch.equal(z_buf.grad, zt.grad)
        z64, y64 = z_new.double().cpu().numpy(), y_new.double().cpu().numpy()
        ref = orc.mmd_gram(y64, z64)
        assert abs(float(loss.detach()) - ref) <= F32_TOL * abs(ref)
        assert relerr(z_buf.grad.cpu().numpy()[rows], orc.mmd_grad_latent_rows(z64[rows], z64, y64)) <= F32_TOL
