#!/usr/bin/env python3
"""Per-rank time of the three-stage (row-sharded) MMD evaluation, measured on ONE GPU.

usage: python tools/prof_staged.py N D [world ...]        (default worlds: 1 2 4 8)

For each world size R the script prepares the whole problem the way R ranks would (every rank's stage A / B into its own
rows of one workspace, records and D x D matrices combined by hand -- the collectives' job), then times what RANK 0
executes, stage by stage, with HIP events: tvae_mmd_stats_local, tvae_mmd_prep_local (image, norms, records, its share
of Z^T Z and Y^T Y: K = N / R) and tvae_mmd_tiles (closed-form kernel, P = Z_rows [M2 | shift], tile kernels, finalize).
The bytes a rank receives between the stages are printed beside them (fp16 image rows and norms of the other ranks; the
lo images only when the device would leave the Q-form; records and matrices are a few MB).  DESIGN.md section 6."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from transformer_vae_amd import _lib  # noqa: E402
from transformer_vae_amd import functional as F  # noqa: E402


def timed(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    N, D = int(sys.argv[1]), int(sys.argv[2])
    worlds = [int(v) for v in sys.argv[3:]] or [1, 2, 4, 8]
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(3)
    z = torch.tanh(torch.randn((N, D), generator=gen, device=dev))
    y = torch.randn((N, D), generator=gen, device=dev)
    code, st = _lib.TVAE_F32, F.HipShardStages
    print(f"N = {N}, D = {D}, fp32 latents (split-fp16 kernels); per-rank milliseconds of rank 0")
    print("%5s %9s %9s %9s %9s %9s %12s" % ("ranks", "rows", "stats", "prep", "tiles", "total", "recv MB/rank"))
    for world in worlds:
        n_loc = N // world
        lay = st.layout(N, n_loc, D, True, code)
        ws = st.workspace(lay[0], dev)
        reg = lambda off, nbytes, dt=torch.uint8: ws[off:off + nbytes].view(dt)
        zs = [z[r * n_loc:(r + 1) * n_loc] for r in range(world)]
        ys = [y[r * n_loc:(r + 1) * n_loc] for r in range(world)]

        def combine(recs, mw):
            t = torch.stack(recs).sum(0)
            t[mw] = torch.stack(recs)[:, mw].max()
            return t

        recs = []
        for r in range(world):
            st.stats(zs[r], ys[r], N, r * n_loc, code, True, ws)
            recs.append(reg(lay[7], lay[8] * 8, torch.float64).clone())
        rec_a = combine(recs, lay[8] - 2)
        reg(lay[7], lay[8] * 8, torch.float64).copy_(rec_a)
        recs, grams = [], []
        for r in reversed(range(world)):          # rank 0 last: its local record stays in the workspace
            st.prep(zs[r], ys[r], N, r * n_loc, code, True, ws)
            recs.append(reg(lay[10], lay[11] * 8, torch.float64).clone())
            if lay[12] >= 0:
                grams.append(reg(lay[12], lay[13] * 4, torch.float32).clone())
        rec_b = combine(recs, lay[11] - 2)
        gsum = torch.stack(grams).sum(0) if grams else None
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dz = torch.empty((n_loc, D), dtype=torch.float32, device=dev)

        def stage_c():
            reg(lay[9], lay[11] * 8, torch.float64).copy_(rec_b)
            if gsum is not None:
                reg(lay[12], lay[13] * 4, torch.float32).copy_(gsum)       # (tvae_mmd_tiles turns it into M2 in place)
            st.tiles(zs[0], N, 0, code, sums, loss, 1.0 / N ** 2, dz, 1.0, 3, ws)

        def stage_a():
            st.stats(zs[0], ys[0], N, 0, code, True, ws)
            reg(lay[7], lay[8] * 8, torch.float64).copy_(rec_a)

        t_a = timed(stage_a)
        t_b = timed(lambda: st.prep(zs[0], ys[0], N, 0, code, True, ws))
        t_c = timed(stage_c, iters=2)
        need_lo = lay[4] > 0 and not (lay[14] and 4.0 * float(rec_b[lay[11] - 2]) / D ** 2 < 0.9 / 16)
        recv = (world - 1) * n_loc * (2 * (lay[3] * (2 if need_lo else 1) + 4)) + (world - 1) * (lay[8] + lay[11]) * 8 + \
            (2 * lay[13] * 4 * (world - 1) // world if world > 1 else 0)
        print("%5d %9d %9.3f %9.3f %9.3f %9.3f %12.1f" % (world, n_loc, t_a, t_b, t_c, t_a + t_b + t_c, recv / 1e6))
        del ws
    print("loss of the last run: %.9e" % float(loss))


if __name__ == "__main__":
    main()
