"""tvae_mmd_fwd_bwd_static (the allocation-free entry point) against tvae_mmd_fwd_bwd, through the C ABI.

Both run the same kernels under the same plan, so the bar is bit-exact on the three sums, the loss and the
gradient.  Two calls in a row share the GPU's built-in scratch array; the second is smaller, so stale contents of
the first (padded copies, slabs, rendezvous counters) would show.  Sizes beyond the built-in array are rejected.
Collected last on purpose (file name): the round-end run is `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import bottleneck_oracle as orc

pytestmark = pytest.mark.gpu


def _call(lib, static, z, y, row_offset, n_rows, loss_scale, grad_scale, want_grad=True):
    from transformer_vae_amd import _lib
    n_total, D = z.shape
    code = _lib.TVAE_BF16 if z.dtype == torch.bfloat16 else _lib.TVAE_F32
    sums = torch.full((3,), float("nan"), dtype=torch.float64, device=z.device)
    loss = torch.full((), float("nan"), dtype=torch.float32, device=z.device)
    dz = torch.full((n_rows, D), float("nan"), dtype=z.dtype, device=z.device) if want_grad else None
    stream = torch.cuda.current_stream(z.device).cuda_stream
    dzp = dz.data_ptr() if want_grad else None
    if static:
        st = lib.tvae_mmd_fwd_bwd_static(z.data_ptr(), y.data_ptr(), n_total, row_offset, n_rows, D, code,
                                         sums.data_ptr(), loss.data_ptr(), loss_scale, dzp, grad_scale, stream)
    else:
        nbytes = lib.tvae_mmd_workspace_bytes_for(n_total, n_rows, D, int(want_grad), code)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
        st = lib.tvae_mmd_fwd_bwd(z.data_ptr(), y.data_ptr(), n_total, row_offset, n_rows, D, code, sums.data_ptr(),
                                  loss.data_ptr(), loss_scale, dzp, grad_scale, ws.data_ptr(), nbytes, stream)
    return st, sums, loss, dz


def _inputs(n, D, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    z = torch.tanh(torch.randn(n, D, generator=g)).to(dtype).cuda()
    y = torch.randn(n, D, generator=g).to(dtype).cuda()
    return z, y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_entry_matches_workspace_entry_on_two_calls(dtype):
    from transformer_vae_amd import _lib
    lib = _lib.lib()
    # first: BASELINE cfg3's MMD (N = 5120, D = 77, all rows); second: smaller, ragged, a row block with an offset
    for (n, D, off, rows, seed) in [(5120, 77, 0, 5120, 11), (1000, 50, 96, 517, 12)]:
        z, y = _inputs(n, D, seed, dtype)
        ls, gs = 5.0 / (8.0 * n * n), 4.0 * 5.0 / (8.0 * n * n * D * D)
        st_a, sums_a, loss_a, dz_a = _call(lib, False, z, y, off, rows, ls, gs)
        st_b, sums_b, loss_b, dz_b = _call(lib, True, z, y, off, rows, ls, gs)
        torch.cuda.synchronize()
        assert st_a == 0, lib.tvae_last_error()
        assert st_b == 0, lib.tvae_last_error()
        assert torch.equal(sums_a, sums_b), (sums_a, sums_b)
        assert torch.equal(loss_a, loss_b)
        assert torch.equal(dz_a, dz_b)
        assert bool(torch.isfinite(dz_b.float()).all()) and bool(torch.isfinite(sums_b).all())
        if dtype == torch.float32:       # and the pair is right, not merely equal: fp64 oracle on the same inputs
            z64, y64 = z.double().cpu().numpy(), y.double().cpu().numpy()
            want = orc.mmd_partial_sums(z64[off:off + rows], y64[off:off + rows], z64, y64)
            np.testing.assert_allclose(sums_b.cpu().numpy(), np.array(want), rtol=1e-5)
            gwant = orc.mmd_grad_latent_rows(z64[off:off + rows], z64, y64) * (gs * D * D * n * n / 4.0)
            err = np.abs(dz_b.double().cpu().numpy() - gwant).max() / np.abs(gwant).max()
            assert err <= 1e-5, err
    # sum-only form (no gradient) after the gradient calls
    z, y = _inputs(640, 33, 13, dtype)
    st_a, sums_a, loss_a, _ = _call(lib, False, z, y, 0, 640, 1.0, 1.0, want_grad=False)
    st_b, sums_b, loss_b, _ = _call(lib, True, z, y, 0, 640, 1.0, 1.0, want_grad=False)
    torch.cuda.synchronize()
    assert st_a == 0 and st_b == 0, lib.tvae_last_error()
    assert torch.equal(sums_a, sums_b) and torch.equal(loss_a, loss_b)


def test_static_entry_rejects_oversize_and_bad_arguments():
    from transformer_vae_amd import _lib
    lib = _lib.lib()
    n, D = 32768, 768
    limit = 64 << 20                                             # TVAE_MMD_STATIC_SCRATCH_BYTES (include/tvae_hip.h)
    assert lib.tvae_mmd_workspace_bytes_for(n, n, D, 1, _lib.TVAE_F32) > limit
    z = torch.zeros(16, dtype=torch.float32, device="cuda")     # never read: the size check comes first
    sums = torch.zeros(3, dtype=torch.float64, device="cuda")
    st = lib.tvae_mmd_fwd_bwd_static(z.data_ptr(), z.data_ptr(), n, 0, n, D, _lib.TVAE_F32, sums.data_ptr(), None, 1.0,
                                     z.data_ptr(), 1.0, None)
    assert st == 3, st                                           # TVAE_ERR_WORKSPACE
    msg = lib.tvae_last_error().decode()
    assert "built-in buffer" in msg and str(limit) in msg, msg
    st = lib.tvae_mmd_fwd_bwd_static(z.data_ptr(), z.data_ptr(), 0, 0, 0, 8, _lib.TVAE_F32, sums.data_ptr(), None, 1.0,
                                     None, 1.0, None)
    assert st == 1, st                                           # TVAE_ERR_ARG


def test_static_entry_under_graph_capture():
    """The caller it is for: capture (no allocation inside) and replay with new contents."""
    from transformer_vae_amd import _lib
    lib = _lib.lib()
    n, D = 350, 77
    z, y = _inputs(n, D, 21, torch.float32)
    sums = torch.zeros(3, dtype=torch.float64, device="cuda")
    loss = torch.zeros((), dtype=torch.float32, device="cuda")
    dz = torch.zeros(n, D, dtype=torch.float32, device="cuda")
    ls, gs = 5.0 / (70.0 * n * n), 4.0 * 5.0 / (70.0 * n * n * D * D)

    def call():
        st = lib.tvae_mmd_fwd_bwd_static(z.data_ptr(), y.data_ptr(), n, 0, n, D, _lib.TVAE_F32, sums.data_ptr(),
                                         loss.data_ptr(), ls, dz.data_ptr(), gs,
                                         torch.cuda.current_stream().cuda_stream)
        assert st == 0, lib.tvae_last_error()

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()                       # warm-up outside capture: symbol look-up, function attributes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    z2, y2 = _inputs(n, D, 22, torch.float32)
    z.copy_(z2); y.copy_(y2)
    graph.replay()
    torch.cuda.synchronize()
    _, sums_ref, loss_ref, dz_ref = _call(lib, False, z2, y2, 0, n, ls, gs)
    torch.cuda.synchronize()
    assert torch.equal(sums, sums_ref) and torch.equal(loss, loss_ref) and torch.equal(dz, dz_ref)
