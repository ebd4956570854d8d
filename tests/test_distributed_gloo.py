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
