"""Two processes on ONE GPU (gloo control plane, the real HIP kernels): the row-sharded global-batch step of
``EncoderDecoderVAE(mmd_scope="global")`` against the same global batch in a single process.

What one rank of an N-GPU run executes -- encode its rows, all-gather latents (and prior), its row block of the MMD
against all columns with the complete gradient of its own rows, 3-double all-reduce, decode -- with gloo standing in for
RCCL (the collectives move through host memory; the kernels and their row_offset / n_rows arguments are the product's).
reg_loss must be the SAME fp32 number on every rank and equal to the single-process one; gradients agree to 1e-6
(the column splits of a row block differ from those of the whole batch, so the fp32 summation order does)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # name: (B_global, S, d, n, L)
    "narrow_D77": (64, 12, 96, 5, 77),
    "narrow_D77_split": (300, 6, 64, 5, 77),      # N = 1 500: the split-fp16 kernel for narrow latents, ragged row blocks
    "wide_D768_split": (96, 8, 128, 8, 768),
    "wide_D300_ragged": (50, 6, 64, 5, 300),
}


def _build(d, L, n, scope, seed=5):
    from types import SimpleNamespace
    import transformer_vae_amd as amd
    cfg = SimpleNamespace(t5=SimpleNamespace(d_model=d), funnel=SimpleNamespace(d_model=d, layer_norm_eps=1e-9),
                          latent_size=L, n_latent_tokens=n)
    torch.manual_seed(seed)
    return amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](cfg), amd.VAE_DECODER_MODELS[""](cfg), True, mmd_scope=scope,
                                 compensate_ddp_mean=False).to("cuda:0")


def _inputs(case):
    B, S, d, n, L = CASES[case]
    g = torch.Generator().manual_seed(77)
    h = torch.randn((B, S, d), generator=g)
    prior = torch.randn((B * min(n, S), L), generator=g)
    g0 = torch.randn((B, min(n, S), d), generator=g) / float(B * n * d) ** 0.5
    return h, prior, g0


def _step(vae, h, prior, g0):
    h = h.to("cuda:0").requires_grad_(True)
    out = vae(input_encoding=h, true_samples=prior.to("cuda:0"))
    (out.reg_loss * 0.5 + (out.reconstructed_encoding * g0.to("cuda:0")).sum()).backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in vae.named_parameters()}
    return out.reg_loss.detach().cpu().numpy(), h.grad.cpu().numpy(), out.latent.detach().cpu().numpy(), grads


def _worker(rank, world, port, case, out, staged=False):
    import torch.distributed as dist
    if staged:      # the three-stage form (sharded preparation) at any size; read when transformer_vae_amd.distributed is imported
        os.environ["TVAE_STAGED_MIN_ELEMENTS"] = "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import traceback
    # a rank that dies must not leave the other one (and the test) waiting: short collective timeout, and every failure
    # travels to the parent through the queue
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        B, S, d, n, L = CASES[case]
        h, prior, g0 = _inputs(case)
        Bl, n_eff = B // world, min(n, S)
        vae = _build(d, L, n, "global")
        sl = slice(rank * Bl, (rank + 1) * Bl)
        reg, dh, z, grads = _step(vae, h[sl], prior[rank * Bl * n_eff:(rank + 1) * Bl * n_eff], g0[sl])
        # parameter gradients are SUMMED over ranks here (compensate_ddp_mean=False): each holds its rows' share
        for k in sorted(grads):
            t = torch.from_numpy(grads[k])
            dist.all_reduce(t)
            grads[k] = t.numpy()
        out.put((rank, reg, dh, z, grads))
    except BaseException:
        out.put((rank, "error", traceback.format_exc(), None, None))
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("staged", [False, True], ids=["gathered", "staged"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_two_ranks_on_one_gpu_equal_the_single_process_step(case, staged):
    """``staged``: the preparation is sharded too (distributed._StagedMMD: tvae_mmd_stats_local / _prep_local / _tiles with
    the collectives between them), which production uses from n_total * D = 2^22 on."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q, staged)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    for r in res:
        assert not (isinstance(r[1], str) and r[1] == "error"), f"rank {r[0]} failed:\n{r[2]}"
    for p in procs:
        assert p.exitcode == 0
    B, S, d, n, L = CASES[case]
    h, prior, g0 = _inputs(case)
    ref_reg, ref_dh, ref_z, ref_grads = _step(_build(d, L, n, "local"), h, prior, g0)
    (_, reg0, dh0, z0, gr0), (_, reg1, dh1, z1, _) = res
    assert reg0.tobytes() == reg1.tobytes()                    # the same fp32 number on every rank
    assert np.array_equal(np.concatenate([z0, z1]), ref_z)     # the projections do not depend on the sharding
    assert abs(float(reg0) - float(ref_reg)) <= 1e-6 * abs(float(ref_reg))
    dh = np.concatenate([dh0, dh1])
    assert np.abs(dh - ref_dh).max() <= 1e-6 * np.abs(ref_dh).max()
    for k, g in ref_grads.items():
        assert np.abs(gr0[k] - g).max() <= 2e-6 * np.abs(g).max(), k
