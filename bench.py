#!/usr/bin/env python3
"""bench.py — bottleneck+MMD forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4] [--dtype fp32]

A step = one pass of the hot path over one batch of synthetic hidden states:
    out  = vae(h, true_samples=prior)                        (encode -> MMD -> decode)
    loss = out.reg_loss * 0.5 + <out.reconstructed_encoding, G0>
    loss.backward()                                          (grads for h and every parameter)
Default workload: BASELINE.json configs[3] — large_model (d_model 1024) full_30_tkn, GLOBAL batch
8192 (N = 245 760 latent rows, D = 768), the configuration the metric's 1/2/4/8-GPU and
%-of-MFMA-roofline figures are quoted on; it fits one GPU, so N=1 runs the whole global batch and
N>1 row-shards the same batch (strong scaling: all-gather of latents + 3-double all-reduce).
Rank 0 prints ONE JSON line.  See DESIGN.md section "Measurement".
"""
import argparse
import ctypes
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name: (B_global, S, d, n, L, cfg_id, BASELINE.json configs[] label)
CONFIGS = {
    "cfg1": (8, 60, 512, 5, 77, 1, "configs[0] t5-small mnist_small+tenth_5_tkn batch 8"),
    "cfg2": (70, 30, 768, 5, 77, 2, "configs[1] syntax t5-base 30Seq 5 latent tokens batch_large"),
    "cfg3": (1024, 237, 768, 5, 77, 3, "configs[2] mnist_base funnel_small window200 237-seq batch 1024"),
    "cfg4": (8192, 30, 1024, 30, 768, 4, "configs[3] large_model d_model 1024 full_30_tkn global batch 8192"),
    "cfg5": (65536, 1, 1024, 1, 1024, 5, "configs[4] MMD stress global batch 65536 latent dim 1024"),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak (= vector fp32 peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 / fp16 dense MFMA peak (~2.5 PFLOP/s)
PEAK_HBM_TBS = 8.0                # MI355X_MICROARCH.md: HBM3E ~8 TB/s
REG_W = 0.5


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class HipEvents:
    """Raw hipEvent_t pairs (ctypes on libamdhip64) handed to tvae_profile_next_mmd."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.pairs = []

    def new_pair(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
        self.pairs.append((a, b))
        return a, b

    def durations_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                out.append(ms.value)
        return out


def mmd_mode(L, mmd_bf16, strict):
    """The arithmetic the MMD kernels of a (latent width, options) pair run in (transformer_vae_amd/csrc/mmd.hip, mmd_mode)."""
    if L > 1024:
        return "fp32"
    if L <= 192:      # narrow latents: fp32 inputs take the split kernel of mmd_n16.hip, bf16 inputs are widened to fp32
        return "fp32" if (strict or mmd_bf16) else "split"
    return "bf16" if mmd_bf16 else ("fp32" if strict else "split")


def mmd_roofline(mode, L, achieved, kern_ms, launches, traffic=None):
    """roofline object of the MMD tile kernels of one call (one gradient launch + one sum-only launch, bracketed by the
    HIP events of tvae_profile_next_mmd).  `achieved` counts ALGORITHMIC flops, 10 N_rows N D."""
    pipe = 192 < L <= 768
    if mode == "split":
        # fp16 products per fp32 product.  192 < L <= 768 (mmd_q32.hip): ONE for the Gram and ONE for the gradient -- both sums
        # carry Q = E + t, the part linear in t is closed form, so executed = algorithmic.  768 < L <= 1024:
        # the same on the 32-row kernel's Q-form instances.  Narrow latents (mmd_n16.hip): three and three.  6 of the 10 algorithmic flops are Gram, 4 gradient
        if L > 192:
            gram_p, grad_p = 1, 1
        else:
            gram_p, grad_p = 3, 3
        split_exec = (6 * gram_p + 4 * grad_p) / 10.0
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_MFMA_TFLOPS if achieved else None, "traffic": traffic,
                "kernel": ("mmd_n16_kernel<KBN,RT,*>" if L <= 192 else
                           "mmd_q32_kernel<KB> (gradient launch) + mmd_q32_sum3_kernel<KB> (y-y sums)" if pipe else "mmd_bf16_kernel<KB,2,*,true,*> (Q-form gradient launch) + mmd_q32_sum3_kernel<KB> (y-y sums)"),
                "kernel_ms": kern_ms, "launches_timed": launches,
                "products_per_fp32_product": {"gram": gram_p, "gradient": grad_p},
                "executed_per_algorithmic": split_exec,
                "executed_tflops": split_exec * achieved if achieved else None,
                "frac_executed": split_exec * achieved / PEAK_BF16_MFMA_TFLOPS if achieved else None,
                "peak_note": "fp16 MFMA dense NOMINAL peak (v_mfma_f32_32x32x16_f16 / 16x16x32_f16; the bare instruction with random operands sustains 1 690 TFLOP/s = 0.68 of it at 1.61 GHz on this part: profiles/r03_mfma_power.txt).  fp32 operands are centred, scaled fp16 pairs "
                             "hi + lo.  192 < D <= 768: Gram and gradient are ONE fp16 product each (hi x hi): every sum the tile kernels "
                             "accumulate carries Q = E + t (second order in t ~ 2/D), and the part linear in t is closed form from "
                             "third moments and Z^T Z - Y^T Y (DESIGN 4.1g/4.1h; a device flag falls back to the three-product E-form "
                             "when t can leave the polynomial range) -- executed = algorithmic (768 < D <= 1024: the same, on the 32-row kernels)."
                             "  D <= 192: three and three.  Of the 10 N^2 D algorithmic flops 6 are Gram and 4 gradient; "
                             "`frac` is algorithmic / peak, `frac_executed` counts the executed products (upper bound: the symmetric "
                             "y-y launch executes half of its share; the fp32 GEMMs of the closed form are outside the timed kernels)"}
    if mode == "bf16":
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_MFMA_TFLOPS if achieved else None, "traffic": None,
                "kernel": ("mmd_pipe_kernel<KB,false> + mmd_pipe_sum_kernel<KB,false>" if pipe else "mmd_bf16_kernel"),
                "kernel_ms": kern_ms, "launches_timed": launches,
                "peak_note": "bf16 MFMA (v_mfma_f32_16x16x32_bf16) dense peak; opt-in mode, latents and prior rounded to bf16; "
                             "algorithmic flops (one matrix product per product)"}
    return {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F32_MFMA_TFLOPS if achieved else None, "traffic": traffic,
            "kernel": "mmd_tile_kernel", "kernel_ms": kern_ms, "launches_timed": launches,
            "peak_note": "fp32-input MFMA (v_mfma_f32_16x16x4_f32) dense peak; the Gram and gradient products run in exact fp32; "
                         "algorithmic flops (the symmetric y-y launch executes half of its share)"}


def make_vae(d, L, n, device, dtype, scope, world, mmd_dtype=None, strict=False, compensate=False):
    import transformer_vae_amd as amd
    cfg = SimpleNamespace(t5=SimpleNamespace(d_model=d), funnel=SimpleNamespace(d_model=d, layer_norm_eps=1e-9),
                          latent_size=L, n_latent_tokens=n)
    torch.manual_seed(0)      # default nn.Linear / LayerNorm init under seed 0
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](cfg), amd.VAE_DECODER_MODELS[""](cfg), True,
                                mmd_scope=scope, compensate_ddp_mean=compensate, mmd_dtype=mmd_dtype, mmd_strict_fp32=strict)
    return vae.to(device).to(dtype)


def cpu_baseline(cfg_name, world):
    """The oracle (a CPU port of the reference arithmetic) timed on this host, on a bounded sample.

    cfg1/cfg2: the literal op sequence of the reference (tile/sub/square/mean/exp, autograd) on the
    whole batch.  cfg3-5: the literal form cannot allocate its (N,N,D) temporaries, so the Gram-form
    restatement is timed on a sample of latent rows against ALL N columns and scaled by N/rows."""
    from oracle import bottleneck_oracle as orc
    from oracle import literal_torch as lit
    B, S, d, n, L, cfg_id, _ = CONFIGS[cfg_name]
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    n_eff = min(n, S)
    N = B * n_eff
    if cfg_name in ("cfg1", "cfg2"):
        h, p, prior, g0 = orc.synth_inputs(B, S, d, n, L, cfg_id, "")
        pt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in p.items()}
        ht = torch.from_numpy(h).requires_grad_(True)
        pr, g0t = torch.from_numpy(prior), torch.from_numpy(g0)
        for _ in range(2):
            lit.step_literal(ht, pt, n, pr, g0t, REG_W)
        ts = []
        t_end = time.perf_counter() + 15.0
        while len(ts) < 20 and time.perf_counter() < t_end:
            t0 = time.perf_counter()
            lit.step_literal(ht, pt, n, pr, g0t, REG_W)
            ts.append(time.perf_counter() - t0)
        sec = float(np.median(ts))
        return {"value": B / sec, "unit": "samples/s", "cores": threads, "kind": "port",
                "sample": f"literal reference op sequence (oracle/literal_torch.py), whole batch B={B}, "
                          f"median of {len(ts)} steps, torch CPU fp32"}
    # Gram-form restatement, row sample: a 256-row probe sizes the sample for 10-20 s of CPU work (bounded by N)
    rs = np.random.RandomState(4321 + cfg_id)
    z_all = np.tanh(rs.standard_normal((N, L))).astype(np.float32)
    y_all = rs.standard_normal((N, L)).astype(np.float32)

    def mmd_sample(rows):
        t0 = time.perf_counter()
        orc.mmd_partial_sums(z_all[:rows], y_all[:rows], z_all, y_all, chunk=min(rows, 512))
        orc.mmd_grad_latent_rows(z_all[:rows], z_all, y_all, chunk=min(rows, 512))
        return time.perf_counter() - t0

    rows = min(N, 256)
    t_probe = mmd_sample(rows)
    want = min(N, int(rows * 20.0 / max(t_probe, 1e-3)) // 256 * 256)
    if want > rows:
        rows = want
        t_probe = mmd_sample(rows)
    t_mmd = t_probe * (N / rows)
    # projections on the same share of the batch (slice-first restatement), fwd + analytic bwd
    bs = max(1, min(B, rows // n_eff))
    h, p, prior, g0 = orc.synth_inputs(bs, min(S, n_eff), d, n, L, cfg_id, "")
    t0 = time.perf_counter()
    fwd = orc.bottleneck_forward(h, p, n, None, use_reg_loss=False)
    orc.bottleneck_backward(h, p, n, fwd, g0)
    t_proj = (time.perf_counter() - t0) * (B / bs)
    sec = t_mmd + t_proj
    return {"value": B / sec, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"Gram-form restatement (oracle/bottleneck_oracle.py, numpy fp32; the literal reference needs "
                      f"{N * N * L * 4 / 2**30:.0f} GiB per temporary): {rows} of {N} latent rows against all {N} columns "
                      f"(sums + gradient) and {bs} of {B} sequences through the projections, scaled to the full step"}


def side_config(cfg_name, device, graphed, lib=None):
    """A bounded run of one of the small BASELINE configs on this GPU (rank 0, N=1 only): the headline stays
    on the default workload, these ride along in `other_configs`.  `graphed`: the whole step (forward, loss,
    backward) captured once into a HIP graph and replayed -- the launch-bound configs spend their time in
    ~30 kernel launches and the autograd bookkeeping around them."""
    B, S, d, n, L, cfg_id, label = CONFIGS[cfg_name]
    n_eff = min(n, S)
    N = B * n_eff
    vae = make_vae(d, L, n, device, torch.float32, "local", 1)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + cfg_id)
    h = torch.randn((B, S, d), generator=gen, device=device).requires_grad_(True)
    gen.manual_seed(4321 + cfg_id)
    prior = torch.randn((N, L), generator=gen, device=device)
    gen.manual_seed(99)
    g0 = torch.randn((B, n_eff, d), generator=gen, device=device) / (N * d) ** 0.5
    params = list(vae.parameters())

    def step(hh, pp):
        out = vae(input_encoding=hh, true_samples=pp)
        loss = out.reg_loss * REG_W + (out.reconstructed_encoding * g0).sum()
        loss.backward()
        return out.reg_loss

    def clear():
        h.grad = None
        for p in params:
            p.grad = None

    for _ in range(3):
        clear()
        step(h, prior)
    torch.cuda.synchronize(device)
    # the launch-bound configurations are host-bound: 50 warm-up steps (the first hundred steps of a fresh process read 0.29 ms
    # against 0.19 ms for the same loop in tools/prof_host.py) and 400 timed ones
    steps = 400 if cfg_name in ("cfg1", "cfg2") else (40 if cfg_name == "cfg3" else 5)
    events = HipEvents() if (lib is not None and not graphed) else None
    if graphed:
        from transformer_vae_amd.graph import GraphedStep
        gs = GraphedStep(step, (h, prior), zero_grads=[h] + params)

        def run():
            return gs(h, prior)
    else:
        def run(timed=False):
            clear()
            if timed and events is not None:
                lib.tvae_profile_next_mmd(*events.new_pair())
            return step(h, prior)
    for _ in range(50 if steps >= 400 else (5 if steps > 5 else 1)):
        run()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        run(True) if events is not None else run()
    torch.cuda.synchronize(device)
    sec = (time.perf_counter() - t0) / steps
    assert h.grad is not None and bool(torch.isfinite(h.grad).all())
    res = {"workload": f"{cfg_name}: {label}", "value": B / sec, "unit": "samples/s", "ms_per_step": sec * 1e3,
           "steps": steps, "hip_graph": bool(graphed)}
    if events is not None and L > 192:
        # MFMA-bound configuration: roofline of its MMD tile kernels, HIP events inside the timed steps (as the headline's)
        k_ms = float(np.mean(events.durations_ms()))
        res["roofline"] = mmd_roofline(mmd_mode(L, False, False), L, 10.0 * N * N * L / (k_ms * 1e-3) / 1e12, k_ms, steps)
    if cfg_name == "cfg3":
        # SURVEY.md 8d, algorithmic HBM bytes of one fp32 step: h[:, :n] read + dh[:, :n] written + recon written + G0 read
        # (4 N d), the zero-fill of dh[:, n:] (B (S - n) d), z written + re-read (~4 N D), the prior read twice (2 N D),
        # weights and their gradients (4 d D); 4 bytes each
        nbytes = 4.0 * (4 * N * d + B * (S - n_eff) * d + 4 * N * L + 2 * N * L + 4 * d * L)
        ach = nbytes / sec / 1e12
        res["roofline"] = {"bound": "hbm", "bytes": nbytes, "achieved": ach, "peak": PEAK_HBM_TBS, "unit": "TB/s",
                           "frac": ach / PEAK_HBM_TBS, "traffic": None,
                           "note": "whole step (all kernels) against the algorithmic bytes of SURVEY.md 8d; "
                                   "the MMD of this configuration (N = 5 120, D = 77) is matrix-core work inside the same step"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--mmd-dtype", default="ref", choices=["ref", "bf16"],
                    help="ref: the reference's promotion (fp32 against its fp32 prior); bf16: opt-in bf16 matrix-core MMD")
    ap.add_argument("--mmd-strict-fp32", action="store_true",
                    help="fp32 inputs: keep the fp32-input MFMA kernels instead of the default split-fp16 form (same 1e-5 class)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the bounded cfg1/cfg2/cfg3 runs reported under other_configs (N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # one process per GPU; TVAE_DIST_BACKEND=gloo lets several ranks share one card for a rehearsal
    backend = os.environ.get("TVAE_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if backend == "nccl":
            # create the RCCL communicator and load its kernels OUTSIDE the timed region, whatever --warmup is
            w8 = torch.zeros(3, dtype=torch.float64, device=device)
            dist.all_reduce(w8)
            wg = torch.empty((world * 4, 8), dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(wg, torch.zeros((4, 8), dtype=torch.float32, device=device))
            torch.cuda.synchronize(device)

    B, S, d, n, L, cfg_id, label = CONFIGS[args.config]
    if B % world:
        raise SystemExit(f"global batch {B} not divisible by {world} ranks")
    dtype = torch.float32 if args.dtype == "fp32" else torch.bfloat16
    Bl = B // world
    n_eff = min(n, S)
    N, Nl = B * n_eff, Bl * n_eff

    from transformer_vae_amd import _lib
    lib = _lib.lib()
    # The launch-bound side configs run FIRST, in a fresh process state (measured: after the cfg4 workload and the
    # CPU baseline's worker threads the same eager loop reads 0.37 ms instead of 0.26 ms per step).
    others = None
    if rank == 0 and world == 1 and not args.no_side_configs:
        others = []
        for name, graphed in (("cfg1", False), ("cfg1", True), ("cfg2", False), ("cfg2", True), ("cfg3", False), ("cfg3", True),
                              ("cfg5", False)):
            if name == args.config:
                continue
            try:
                others.append(side_config(name, device, graphed, lib if name == "cfg5" else None))
            except Exception as e:
                others.append({"workload": name, "hip_graph": graphed, "value": None, "error": repr(e)[:200]})
        torch.cuda.synchronize(device)
        torch.cuda.empty_cache()
    mmd_bf16 = args.mmd_dtype == "bf16"
    # N > 1: global-batch MMD with the DDP compensation on (each rank's latent gradient x R), and the parameter gradients
    # averaged over the ranks inside the timed step, as DistributedDataParallel does for the reference's trainer
    # (trainer.py:258-281; HF Trainer wraps the model in DDP): ONE flat all-reduce of all six bottleneck gradients
    vae = make_vae(d, L, n, device, dtype, "global" if world > 1 else "local", world,
                   torch.bfloat16 if mmd_bf16 else None, args.mmd_strict_fp32, compensate=world > 1)
    # rank r takes rows [r B/R, (r+1) B/R) of the SAME global tensors (generated on every rank from
    # the same seeds, sliced, the rest released), so reg_loss is the same number at every N -- a built-in check
    # of the sharded path against the single-GPU one.
    gen = torch.Generator(device=device)

    def global_rows(shape, seed, lo, hi, div=None):
        gen.manual_seed(seed)
        full = torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
        if div is not None:
            return full[lo:hi] / div
        return full if world == 1 else full[lo:hi].clone()

    h = global_rows((B, S, d), 1234 + cfg_id, rank * Bl, (rank + 1) * Bl).to(dtype).requires_grad_(True)
    prior = global_rows((N, L), 4321 + cfg_id, rank * Nl, (rank + 1) * Nl)
    g0 = global_rows((B, n_eff, d), 99, rank * Bl, (rank + 1) * Bl, (N * d) ** 0.5).to(dtype)
    if world > 1:
        torch.cuda.empty_cache()
    params = list(vae.parameters())

    events = HipEvents()

    def step(timed):
        h.grad = None
        for p in params:
            p.grad = None
        if timed:
            a, b = events.new_pair()
            lib.tvae_profile_next_mmd(a, b)
        out = vae(input_encoding=h, true_samples=prior)
        loss = out.reg_loss * REG_W + (out.reconstructed_encoding * g0).sum()
        loss.backward()
        if world > 1:
            from transformer_vae_amd.distributed import all_reduce_sum
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            all_reduce_sum(flat).div_(world)
            off = 0
            for p in params:
                p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                off += p.numel()
        return out

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    reg = float(out.reg_loss.detach())
    assert np.isfinite(reg) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    assert h.grad is not None and h.grad.shape == h.shape

    ms_per_step = elapsed / args.steps * 1e3
    value = B / (elapsed / args.steps)

    # ---- roofline of the dominant kernels (the MMD tile kernels of the call), measured live with HIP events ----
    tile_ms = events.durations_ms()
    kern_ms = float(np.mean(tile_ms)) if tile_ms else float("nan")
    # algorithmic flops of one launch on this rank: 10 * N_local_rows * N * D (3 Grams fwd
    # 6 N^2 D + 2 gradient products 4 N^2 D, no symmetry credit, recompute not counted)
    flops = 10.0 * Nl * N * L
    achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms == kern_ms else None
    mode = mmd_mode(L, mmd_bf16 or args.dtype == "bf16", args.mmd_strict_fp32)
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get(f"{args.config}_{args.dtype}_{mode}_gpus{world}")
        except Exception:
            traffic = None
    roofline = mmd_roofline(mode, L, achieved, kern_ms, len(tile_ms), traffic if mode != "bf16" else None)

    line = {
        "metric": "bottleneck+MMD fwd+bwd samples/sec", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32" if args.dtype == "fp32" else "bf16") + ("+bf16-mmd" if mmd_bf16 else "") +
                 (" (MMD products as split fp16 pairs, fp32 accumulate)" if mode == "split" else ""), "data": "synthetic",
        "config": {"workload": f"{args.config}: {label}", "global_batch": B, "seq_len": S, "d_model": d,
                   "n_latent_tokens": n, "latent_size": L, "mmd_rows": N,
                   "parallelism": f"row-sharded global-batch MMD x{world}" if world > 1 else "single GPU"},
        "reg_loss": reg, "roofline": roofline,
        "ranks_seen": dist.get_world_size() if world > 1 else 1,
        "param_grad_allreduce": world > 1,
    }
    if rank == 0 and world == 1 and others is not None and not mmd_bf16 and args.dtype == "fp32" and 192 < L <= 1024:
        # the other two arithmetic modes of the MMD on the SAME inputs, reported beside the headline, never instead of it:
        #   strict fp32 (mmd_strict_fp32=True: fp32-input MFMA, the round-1 headline) and the opt-in bf16 matrix-core MMD
        #   (mmd_dtype=torch.bfloat16: latents and prior rounded to bf16 for the regulariser only, 1e-2 parity class)
        for tag, kw, m2 in ((("strict fp32 MMD (fp32-input MFMA)", dict(strict=True), "fp32"),) if not args.mmd_strict_fp32 else ()) + \
                (("opt-in bf16 matrix-core MMD", dict(mmd_dtype=torch.bfloat16), "bf16"),):
            try:
                vae2 = make_vae(d, L, n, device, dtype, "local", 1, **kw)
                p2 = list(vae2.parameters())
                ev2 = HipEvents()

                def step2(timed):
                    h.grad = None
                    for p in p2:
                        p.grad = None
                    if timed:
                        lib.tvae_profile_next_mmd(*ev2.new_pair())
                    o = vae2(input_encoding=h, true_samples=prior)
                    (o.reg_loss * REG_W + (o.reconstructed_encoding * g0).sum()).backward()
                    return o

                step2(False)
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    o2 = step2(True)
                fence()
                sec2 = (time.perf_counter() - t1) / args.steps
                k2 = float(np.mean(ev2.durations_ms()))
                others.append({"workload": f"{args.config}: {label} -- {tag}", "value": B / sec2,
                               "unit": "samples/s", "ms_per_step": sec2 * 1e3, "steps": args.steps, "hip_graph": False,
                               "dtype": "f32" if m2 == "fp32" else "f32+bf16-mmd", "reg_loss": float(o2.reg_loss.detach()),
                               "roofline": mmd_roofline(m2, L, flops / (k2 * 1e-3) / 1e12, k2, args.steps)})
                del vae2, p2
            except Exception as e:
                others.append({"workload": f"{args.config} {tag}", "hip_graph": False, "value": None, "error": repr(e)[:200]})
    if rank == 0:
        if others is not None:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.config, world)
            except Exception as e:      # the baseline must never take the bench line down
                line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
