"""CPU, world_size 2 over gloo: the collective logic of the row-sharded global-batch MMD
(all-gather of latent/prior row blocks, per-rank row-block evaluation, 3-double all-reduce,
gradient without exchange).  The row-block compute is injected from the oracle HERE ONLY — the
package default is the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bottleneck_oracle as orc


class OracleRowBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_rows, z_all, y_all, row_offset, loss_scale, grad_scale):
        zr, za, ya = (t.detach().double().numpy() for t in (z_rows, z_all, y_all))
        n, dim = za.shape
        yr = ya[row_offset:row_offset + zr.shape[0]]
        sums = np.array(orc.mmd_partial_sums(zr, yr, za, ya))
        dz = orc.mmd_grad_latent_rows(zr, za, ya) * (grad_scale * n * n * dim * dim / 4.0)
        ctx.save_for_backward(torch.from_numpy(dz))
        ctx.mark_non_differentiable(s := torch.from_numpy(sums))
        return torch.tensor(loss_scale * (sums[2] + sums[0] - 2 * sums[1]), dtype=torch.float32), s

    @staticmethod
    def backward(ctx, g, _):
        (dz,) = ctx.saved_tensors
        return (dz * g.double()).to(torch.float32), None, None, None, None, None


def _worker(rank, world, port, compensate, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from transformer_vae_amd.distributed import all_gather_rows, sharded_mmd_reg_loss
        B, n, D = 6, 3, 7                       # per rank
        rs = np.random.RandomState(100 + rank)
        z = torch.from_numpy(np.tanh(rs.standard_normal((B * n, D))).astype(np.float32)).requires_grad_(True)
        y = torch.from_numpy(rs.standard_normal((B * n, D)).astype(np.float32))
        reg = sharded_mmd_reg_loss(z, y, B, n, None, compensate, OracleRowBlock.apply)
        (reg * 0.5).backward()
        g = all_gather_rows(z.grad)
        if rank == 0:
            out.put((float(reg), g.numpy(), all_gather_rows(z.detach()).numpy(), all_gather_rows(y).numpy()))
        else:
            all_gather_rows(z.detach()); all_gather_rows(y)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("compensate", [False, True])
def test_sharded_global_mmd_equals_single_process(compensate):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, compensate, q)) for port in [_free_port()] for r in range(world)]
    for p in procs:
        p.start()
    reg, grad, z_all, y_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch (rank order), fp64 oracle
    B, n = 6 * world, 3
    z64, y64 = z_all.astype(np.float64), y_all.astype(np.float64)
    ref = orc.mmd_gram(y64, z64) * n / B                                   # autoencoders.py:112 on the global batch
    assert abs(reg - ref) <= 1e-6 * abs(ref)
    gref = 0.5 * (n / B) * orc.mmd_grad_latent_rows(z64, z64, y64) * (world if compensate else 1)
    assert np.abs(grad - gref).max() <= 1e-6 * np.abs(gref).max()


# ---------------------------------------------------------------------------------------------------------------
# DistributedDataParallel: DDP AVERAGES parameter gradients over ranks; with compensate_ddp_mean (the default under
# global scope) the averaged gradients must equal those of ONE process holding the global batch (trainer.py:265-266
# is the reference's only multi-GPU line; it never measured a curve).
# ---------------------------------------------------------------------------------------------------------------
class _TinyEncoder(torch.nn.Module):
    def __init__(self, d, D):
        super().__init__()
        self.lin = torch.nn.Linear(d, D)

    def forward(self, h):
        return torch.tanh(self.lin(h))


def _ddp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from transformer_vae_amd.distributed import sharded_mmd_reg_loss
        B, n, d, D = 4, 3, 5, 7                 # per rank
        torch.manual_seed(3)
        enc = _TinyEncoder(d, D)
        ddp = DDP(enc)
        rs = np.random.RandomState(200 + rank)
        h = torch.from_numpy(rs.standard_normal((B * n, d)).astype(np.float32))
        y = torch.from_numpy(rs.standard_normal((B * n, D)).astype(np.float32))
        z = ddp(h)
        reg = sharded_mmd_reg_loss(z, y, B, n, None, True, OracleRowBlock.apply)     # compensate_ddp_mean = True
        reg.backward()                                                               # DDP all-reduces and divides by world
        hs, ys = [torch.empty_like(h) for _ in range(world)], [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(hs, h)
        dist.all_gather(ys, y)
        if rank == 0:
            out.put((float(reg), enc.lin.weight.grad.numpy(), enc.lin.bias.grad.numpy(), torch.cat(hs).numpy(), torch.cat(ys).numpy()))
    finally:
        dist.destroy_process_group()


def test_ddp_averaged_parameter_grads_equal_the_global_batch_grads():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    reg, gw, gb, h_all, y_all = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one process, the global batch: autograd through tanh(Linear) with the oracle's latent gradient
    B, n = 4 * world, 3
    torch.manual_seed(3)
    enc = _TinyEncoder(5, 7).double()
    z = enc(torch.from_numpy(h_all).double())
    ref = orc.mmd_gram(y_all.astype(np.float64), z.detach().numpy()) * n / B
    gz = (n / B) * orc.mmd_grad_latent_rows(z.detach().numpy(), z.detach().numpy(), y_all.astype(np.float64))
    z.backward(torch.from_numpy(gz))
    assert abs(reg - ref) <= 1e-6 * abs(ref)
    assert np.abs(gw - enc.lin.weight.grad.numpy()).max() <= 2e-6 * np.abs(enc.lin.weight.grad.numpy()).max()
    assert np.abs(gb - enc.lin.bias.grad.numpy()).max() <= 2e-6 * np.abs(enc.lin.bias.grad.numpy()).max()


# ---------------------------------------------------------------------------------------------------------------
# Global scope draws its prior per rank: the reference launcher seeds all ranks alike (train.py:391), so the default
# generator would make the gathered prior R copies of one block.
# ---------------------------------------------------------------------------------------------------------------
def _prior_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        import transformer_vae_amd as amd
        cfg = SimpleNamespace(t5=SimpleNamespace(d_model=8), funnel=SimpleNamespace(d_model=8, layer_norm_eps=1e-9),
                              latent_size=3, n_latent_tokens=2)
        torch.manual_seed(11 + 5 * rank)                        # per-rank seeding (seed + rank): the base seed must still agree
        mk = lambda: amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](cfg), amd.VAE_DECODER_MODELS[""](cfg), True, mmd_scope="global")
        cpu = torch.device("cpu")
        a = mk()
        own1, own2 = a._draw_prior((4, 3), cpu), a._draw_prior((4, 3), cpu)
        b = mk()
        all1, all2 = b._draw_prior_all_ranks((4, 3), cpu), b._draw_prior_all_ranks((4, 3), cpu)
        c = mk()                                                # the one-argument form first, then the replay: streams stay aligned
        c._draw_prior((4, 3), cpu)
        mixed = c._draw_prior_all_ranks((4, 3), cpu)
        blocks = [torch.empty_like(own1) for _ in range(world)]
        dist.all_gather(blocks, own1)
        out.put((rank, own1.numpy(), own2.numpy(), all1.numpy(), all2.numpy(), torch.cat(blocks).numpy(), mixed.numpy()))
    finally:
        dist.destroy_process_group()


def test_global_scope_prior_streams_differ_per_rank_and_replay_everywhere():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prior_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, o0, o0b, a0, a0b, g0, m0), (_, o1, o1b, a1, a1b, g1, m1) = res
    assert np.array_equal(m0, m1) and np.array_equal(m0, a0b)                  # same second draw on every rank, whichever form drew first
    assert not np.array_equal(o0, o1) and not np.array_equal(o0, o0b)          # ranks differ, draws advance
    assert np.array_equal(g0, np.concatenate([o0, o1])) and np.array_equal(g0, g1)
    # the all-rank replay is the gathered prior, on every rank, draw after draw: no all-gather needed
    assert np.array_equal(a0, g0) and np.array_equal(a1, g0)
    assert np.array_equal(a0b, np.concatenate([o0b, o1b])) and np.array_equal(a1b, a0b)


# ---- the staged form (sharded preparation): its collectives rehearsed with a CPU stand-in for the three C ABI stages -----------
class OracleStages:
    """Same four methods as functional.HipShardStages, same layout contract (include/tvae_hip.h, tvae_mmd_shard_layout), numpy
    inside: fp32 "images" of the centred rows, no lo image.  `tiles` reads only the workspace -- every column it sees arrived
    through the collectives under test -- and checks the combined records and the summed D x D matrices against the images."""
    calls = []

    @staticmethod
    def layout(n_total, n_rows, dim, want_grad, code):
        off, lay = 0, [0] * 16

        def take(nbytes):
            nonlocal off
            o, off = off, (off + nbytes + 255) // 256 * 256
            return o

        lay[1], lay[2] = take(n_total * dim * 4), take(n_total * dim * 4)
        lay[3], lay[4] = dim * 4, 0
        lay[5], lay[6] = take(n_total * 4), take(n_total * 4)
        lay[7], lay[8] = take((2 * dim + 2) * 8), 2 * dim + 2
        lay[9], lay[10], lay[11] = take((4 * dim + 4) * 8), take((4 * dim + 4) * 8), 4 * dim + 4
        lay[12], lay[13] = take(2 * dim * dim * 4), 2 * dim * dim
        lay[14], lay[15] = 0, dim
        lay[0] = off
        return lay

    @staticmethod
    def workspace(nbytes, device):
        return torch.zeros(nbytes, dtype=torch.uint8)

    @staticmethod
    def _views(ws, n_total, n_rows, dim):
        lay = OracleStages.layout(n_total, n_rows, dim, True, 0)
        v = lambda off, count, dt: ws[off:off + count * torch.empty((), dtype=dt).element_size()].view(dt).numpy()
        return dict(zi=v(lay[1], n_total * dim, torch.float32).reshape(n_total, dim), yi=v(lay[2], n_total * dim, torch.float32).reshape(n_total, dim),
                    zn=v(lay[5], n_total, torch.float32), yn=v(lay[6], n_total, torch.float32), ra=v(lay[7], lay[8], torch.float64),
                    rb=v(lay[9], lay[11], torch.float64), rbl=v(lay[10], lay[11], torch.float64),
                    gram=v(lay[12], lay[13], torch.float32).reshape(2, dim, dim))

    @staticmethod
    def stats(z, y, n_total, row_offset, code, want_grad, ws):
        OracleStages.calls.append("stats")
        n_rows, dim = z.shape
        w = OracleStages._views(ws, n_total, n_rows, dim)
        zz, yy = z.double().numpy(), y.double().numpy()
        w["ra"][:dim], w["ra"][dim:2 * dim] = zz.sum(0), yy.sum(0)
        w["ra"][2 * dim], w["ra"][2 * dim + 1] = max(np.abs(zz).max(), np.abs(yy).max()), 0.0

    @staticmethod
    def prep(z, y, n_total, row_offset, code, want_grad, ws):
        OracleStages.calls.append("prep")
        n_rows, dim = z.shape
        w = OracleStages._views(ws, n_total, n_rows, dim)
        mu = ((w["ra"][:dim] + w["ra"][dim:2 * dim]) / (2.0 * n_total)).astype(np.float32)
        sl = slice(row_offset, row_offset + n_rows)
        for img, nrm, src, k in ((w["zi"], w["zn"], z, 0), (w["yi"], w["yn"], y, 1)):
            c = src.numpy() - mu
            img[sl], nrm[sl] = c, (c.astype(np.float64) ** 2).sum(1)
            c64 = c.astype(np.float64)
            w["rbl"][k * dim:(k + 1) * dim] = c64.sum(0)
            w["rbl"][(2 + k) * dim:(3 + k) * dim] = ((c64 ** 2).sum(1)[:, None] * c64).sum(0)
            w["rbl"][4 * dim + k] = (c64 ** 2).sum()
            w["gram"][k] = (c64.T @ c64).astype(np.float32)
        w["rbl"][4 * dim + 2] = max(w["zn"][sl].max(), w["yn"][sl].max())
        w["rbl"][4 * dim + 3] = 0.0
        w["rb"][:] = w["rbl"]

    @staticmethod
    def tiles(z, n_total, row_offset, code, sums, loss, loss_scale, dz, grad_scale, phase, ws):
        OracleStages.calls.append(f"tiles{phase}")
        n_rows, dim = z.shape
        w = OracleStages._views(ws, n_total, n_rows, dim)
        sl = slice(row_offset, row_offset + n_rows)
        ya = w["yi"].astype(np.float64)
        if phase & 1:       # only the prior's image, its norms and the records are promised here
            assert np.allclose(w["rb"][dim:2 * dim], ya.sum(0), rtol=1e-9, atol=1e-9)
            assert np.isclose(w["rb"][4 * dim + 1], (ya ** 2).sum(), rtol=1e-6)
            assert np.allclose(w["yn"], (ya ** 2).sum(1), rtol=1e-5)
            OracleStages.syy = orc.mmd_partial_sums(ya[sl], ya[sl], ya, ya)[2]
        if phase & 2:
            za = w["zi"].astype(np.float64)
            assert np.allclose(w["rb"][:dim], za.sum(0), rtol=1e-9, atol=1e-9)
            assert np.isclose(w["rb"][4 * dim + 2], max((za ** 2).sum(1).max(), (ya ** 2).sum(1).max()), rtol=1e-5)     # MAX, not SUM
            assert np.allclose(w["gram"][0], za.T @ za, rtol=1e-4, atol=1e-4) and np.allclose(w["gram"][1], ya.T @ ya, rtol=1e-4, atol=1e-4)
            assert np.allclose(za[sl], z.double().numpy() - (w["ra"][:dim] + w["ra"][dim:2 * dim]) / (2.0 * n_total), atol=1e-6)
            s = np.array(orc.mmd_partial_sums(za[sl], ya[sl], za, ya))
            s[2] = OracleStages.syy
            sums.copy_(torch.from_numpy(s))
            loss.copy_(torch.tensor(loss_scale * (s[2] + s[0] - 2 * s[1]), dtype=torch.float32))
            if dz is not None:
                g = orc.mmd_grad_latent_rows(za[sl], za, ya) * (grad_scale * n_total * n_total * dim * dim / 4.0)
                dz.copy_(torch.from_numpy(g).to(dz.dtype))


def _staged_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TVAE_STAGED_MIN_ELEMENTS"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from transformer_vae_amd.distributed import all_gather_rows, sharded_mmd_reg_loss
        B, n, D = 6, 3, 7                       # per rank
        rs = np.random.RandomState(100 + rank)
        z = torch.from_numpy((np.tanh(rs.standard_normal((B * n, D))) + 3.0).astype(np.float32)).requires_grad_(True)
        y = torch.from_numpy((rs.standard_normal((B * n, D)) + 3.0).astype(np.float32))
        reg = sharded_mmd_reg_loss(z, y, B, n, None, False, None, stages=OracleStages)
        (reg * 0.5).backward()
        assert OracleStages.calls == ["stats", "prep", "tiles1", "tiles2"], OracleStages.calls
        g = all_gather_rows(z.grad)
        za, ya = all_gather_rows(z.detach()), all_gather_rows(y)
        if rank == 0:
            out.put((float(reg), g.numpy(), za.numpy(), ya.numpy()))
    finally:
        dist.destroy_process_group()


def test_staged_sharded_mmd_collectives_with_a_cpu_stand_in():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_staged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    reg, grad, z_all, y_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, n = 6 * world, 3
    z64, y64 = z_all.astype(np.float64), y_all.astype(np.float64)
    ref = orc.mmd_gram(y64, z64) * n / B
    assert abs(reg - ref) <= 1e-6 * abs(ref)
    gref = 0.5 * (n / B) * orc.mmd_grad_latent_rows(z64, z64, y64)
    assert np.abs(grad - gref).max() <= 1e-6 * np.abs(gref).max()
