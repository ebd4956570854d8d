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
