"""torch.autograd.Functions over the C ABI (include/tvae_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream and autograd's graph.
Every number is produced by the hand-written kernels in csrc/.  Inputs must live on a
ROCm device; anything else raises (no CPU path).
"""
from __future__ import annotations

import functools
from typing import Optional

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.TVAE_F32, torch.bfloat16: _lib.TVAE_BF16}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _on_tensor_device(fn):
    """Run ``fn`` with the device of its first ROCm tensor argument current.

    HIP resolves a kernel's code object for the CURRENT device, while the C ABI launches on the stream it is
    given: a module living on cuda:1 called while cuda:0 is current would otherwise launch cuda:0's code object
    on cuda:1's queue.  PyTorch's own ops guard the same way.  Free when the device is already current."""
    @functools.wraps(fn)
    def guarded(*args):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args)
                break
        return fn(*args)
    return guarded


def _workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    if nbytes < 0:
        raise RuntimeError("libtvae_hip: unsupported problem size (workspace query failed)")
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the bottleneck kernels run on MI355X only "
                           "(there is no CPU fallback)")


def _compute_dtype(t: torch.Tensor) -> torch.dtype:
    """fp32 stays fp32; bf16 (or an active bf16 autocast region) computes on bf16 storage."""
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.bfloat16 if t.dtype == torch.bfloat16 else torch.float32


class _LatentEncode(torch.autograd.Function):
    """latent = tanh(h[:, :n] @ We^T + be)      (reference: autoencoders.py:21-22)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, h, We, be, n_tokens: int):
        _require_device(h, "latent encoder")
        if h.dim() != 3:
            raise ValueError(f"encoder input must be (batch, seq, d_model), got {tuple(h.shape)}")
        out_dtype = h.dtype
        dt = _compute_dtype(h)
        hc = h.to(dt)
        if hc.stride(2) != 1:
            hc = hc.contiguous()
        Wc, bc = We.to(dt).contiguous(), be.to(dt).contiguous()
        B, S, d = hc.shape
        L = Wc.shape[0]
        if Wc.shape[1] != d:
            raise ValueError(f"token_to_latent expects d_model={Wc.shape[1]}, got hidden size {d}")
        n_eff = min(int(n_tokens), S)          # the reference's slice silently truncates (:22)
        z = torch.empty((B, n_eff, L), dtype=dt, device=h.device)
        if B * n_eff > 0:
            lib = _lib.lib()
            ws = _workspace(lib.tvae_encode_fwd_workspace_bytes(B, n_eff, d, L), h.device)
            _lib.check(lib.tvae_encode_fwd(_ptr(hc), hc.stride(0), hc.stride(1), _ptr(Wc), _ptr(bc), _ptr(z),
                                           B, n_eff, d, L, _DTYPES[dt], _ptr(ws), ws.numel(), _stream(hc)),
                       "tvae_encode_fwd")
        ctx.save_for_backward(hc, Wc, z)
        ctx.meta = (B, S, d, L, n_eff, dt, h.dtype, We.dtype, be.dtype)
        return z if dt == out_dtype or torch.is_autocast_enabled() else z.to(out_dtype)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dz):
        hc, Wc, z = ctx.saved_tensors
        B, S, d, L, n_eff, dt, h_dtype, w_dtype, b_dtype = ctx.meta
        need_h, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dzc = dz.to(dt).contiguous()
        dh = torch.empty((B, S, d), dtype=dt, device=hc.device) if need_h else None
        dWe = torch.empty((L, d), dtype=dt, device=hc.device) if need_w else None
        dbe = torch.empty((L,), dtype=dt, device=hc.device) if need_b else None
        if B * n_eff == 0:
            for t in (dh, dWe, dbe):
                if t is not None:
                    t.zero_()
        else:
            lib = _lib.lib()
            ws = _workspace(lib.tvae_encode_bwd_workspace_bytes(B, n_eff, d, L), hc.device)
            _lib.check(lib.tvae_encode_bwd(_ptr(dzc), _ptr(z), _ptr(hc), hc.stride(0), hc.stride(1), _ptr(Wc),
                                           _ptr(dh), _ptr(dWe), _ptr(dbe), B, S, n_eff, d, L, _DTYPES[dt],
                                           _ptr(ws), ws.numel(), _stream(hc)), "tvae_encode_bwd")
        cast = lambda t, to: None if t is None else (t if t.dtype == to else t.to(to))
        return cast(dh, h_dtype), cast(dWe, w_dtype), cast(dbe, b_dtype), None


class _LatentDecode(torch.autograd.Function):
    """recon = [LayerNorm]( latent @ Wd^T + bd )    (reference: autoencoders.py:38-39, 66-67)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, z, Wd, bd, gamma, beta, eps: float):
        _require_device(z, "latent decoder")
        out_dtype = z.dtype
        dt = _compute_dtype(z)
        lead = z.shape[:-1]
        L = z.shape[-1]
        zc = z.to(dt).reshape(-1, L).contiguous()
        N = zc.shape[0]
        has_lin, has_ln = Wd is not None, gamma is not None
        d = Wd.shape[0] if has_lin else L
        if has_lin and Wd.shape[1] != L:
            raise ValueError(f"latent_to_token expects latent_size={Wd.shape[1]}, got {L}")
        Wc = Wd.to(dt).contiguous() if has_lin else None
        bc = bd.to(dt).contiguous() if has_lin else None
        gc = gamma.to(dt).contiguous() if has_ln else None
        btc = beta.to(dt).contiguous() if has_ln else None
        dev = z.device
        recon = torch.empty((N, d), dtype=dt, device=dev)
        pre = torch.empty((N, d), dtype=dt, device=dev) if (has_lin and has_ln) else None
        mean = torch.empty((N,), dtype=torch.float32, device=dev) if has_ln else None
        rstd = torch.empty((N,), dtype=torch.float32, device=dev) if has_ln else None
        if N > 0:
            lib = _lib.lib()
            ws = _workspace(lib.tvae_decode_fwd_workspace_bytes(N, L, d), dev)
            _lib.check(lib.tvae_decode_fwd(_ptr(zc), _ptr(Wc), _ptr(bc), _ptr(gc), _ptr(btc), float(eps), _ptr(recon),
                                           _ptr(pre), _ptr(mean), _ptr(rstd), N, L, d, _DTYPES[dt], _ptr(ws),
                                           ws.numel(), _stream(zc)), "tvae_decode_fwd")
        ctx.save_for_backward(zc, Wc, gc, pre, mean, rstd)
        ctx.meta = (N, L, d, dt, tuple(lead), z.dtype,
                    None if Wd is None else Wd.dtype, None if gamma is None else gamma.dtype)
        out = recon.reshape(*lead, d)
        return out if dt == out_dtype or torch.is_autocast_enabled() else out.to(out_dtype)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dout):
        zc, Wc, gc, pre, mean, rstd = ctx.saved_tensors
        N, L, d, dt, lead, z_dtype, w_dtype, g_dtype = ctx.meta
        need = ctx.needs_input_grad
        dev = zc.device
        dc = dout.to(dt).reshape(N, d).contiguous()
        has_lin, has_ln = Wc is not None, gc is not None
        dz = torch.empty((N, L), dtype=dt, device=dev) if need[0] else None
        dWd = torch.empty((d, L), dtype=dt, device=dev) if (has_lin and need[1]) else None
        dbd = torch.empty((d,), dtype=dt, device=dev) if (has_lin and need[2]) else None
        want_ln = has_ln and (need[3] or need[4])
        dgamma = torch.empty((d,), dtype=dt, device=dev) if want_ln else None
        dbeta = torch.empty((d,), dtype=dt, device=dev) if want_ln else None
        if N == 0:
            for t in (dWd, dbd, dgamma, dbeta):
                if t is not None:
                    t.zero_()
        else:
            lib = _lib.lib()
            ws = _workspace(lib.tvae_decode_bwd_workspace_bytes(N, L, d), dev)
            _lib.check(lib.tvae_decode_bwd(_ptr(dc), _ptr(zc), _ptr(Wc), _ptr(gc), _ptr(pre), _ptr(mean), _ptr(rstd),
                                           _ptr(dz), _ptr(dWd), _ptr(dbd), _ptr(dgamma), _ptr(dbeta), N, L, d,
                                           _DTYPES[dt], _ptr(ws), ws.numel(), _stream(zc)), "tvae_decode_bwd")
        cast = lambda t, to: None if t is None else (t if t.dtype == to else t.to(to))
        dz_out = None if dz is None else cast(dz.reshape(*lead, L), z_dtype)
        return dz_out, cast(dWd, w_dtype), cast(dbd, w_dtype), cast(dgamma, g_dtype), cast(dbeta, g_dtype), None


def _mmd_code(dt: torch.dtype, strict_fp32: bool) -> int:
    """dtype code of a tvae_mmd_* call: fp32 inputs run the split-fp16 products (two or three matrix-core products per fp32
    product, same 1e-5 accuracy class) unless ``strict_fp32`` asks for the fp32-input matrix instruction."""
    if dt == torch.float32 and strict_fp32:
        return _lib.TVAE_F32_STRICT
    return _DTYPES[dt]


# the four-launch whole step of csrc/small_step.hip for the reference's own batch sizes (False: always the general path).
# Measured crossover (graph-replayed step, MI355X): N = 40 latent rows (BASELINE cfg1) 74 us against 94 us on the general
# kernels; N = 350 (cfg2) 140 us against 115 us -- one workgroup per row pair streams every weight through one CU, the
# general path tiles the same products over the whole chip.  The C ABI takes up to 512 rows; the module uses it up to:
SMALL_STEP = True
SMALL_STEP_MAX_ROWS = 128


class _MMDRegulariser(torch.autograd.Function):
    """reg = MMD(prior, latent) * scale, forward and gradient in one device pass
    (reference: autoencoders.py:117-136; closed-form gradient SURVEY.md 8a).

    ``row_offset``/``z_all``/``y_all`` implement the row-sharded global-batch form: the
    caller has already all-gathered latents and priors; this rank evaluates its row block
    and returns the three partial sums, which the caller all-reduces."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, z_rows, z_all, y_all, row_offset: int, loss_scale: float, grad_scale: float, strict_fp32: bool = False,
                grad_enabled: bool = True):
        _require_device(z_all, "MMD regulariser")
        dt = torch.bfloat16 if (z_all.dtype == torch.bfloat16 and y_all.dtype == torch.bfloat16) else torch.float32
        code = _mmd_code(dt, strict_fp32)
        za = z_all.detach().to(dt).contiguous()
        ya = y_all.detach().to(dt).contiguous()
        n_total, D = za.shape
        if tuple(ya.shape) != (n_total, D):
            raise ValueError(f"prior samples {tuple(ya.shape)} must match latents {(n_total, D)}")
        n_rows = z_rows.shape[0]
        # needs_input_grad stays True under torch.no_grad(); the wrappers pass the grad mode they were called in, so that
        # evaluation steps run the sum-only plan (no gradient slabs, no dz)
        want_grad = bool(grad_enabled) and bool(ctx.needs_input_grad[0])
        dev = za.device
        lib = _lib.lib()
        ws_bytes = lib.tvae_mmd_workspace_bytes_for(n_total, n_rows, D, int(want_grad), code)
        if ws_bytes < 0:
            raise RuntimeError(f"MMD regulariser: unsupported problem size (n_total={n_total}, n_rows={n_rows}, D={D})")
        ws = _workspace(ws_bytes, dev)
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dz = torch.empty((n_rows, D), dtype=dt, device=dev) if want_grad else None
        _lib.check(lib.tvae_mmd_fwd_bwd(_ptr(za), _ptr(ya), n_total, int(row_offset), n_rows, D, code,
                                        _ptr(sums), _ptr(loss), float(loss_scale), _ptr(dz), float(grad_scale),
                                        _ptr(ws), ws.numel(), _stream(za)), "tvae_mmd_fwd_bwd")
        if want_grad:
            ctx.save_for_backward(dz)
        ctx.z_dtype = z_rows.dtype
        ctx.mark_non_differentiable(sums)
        return loss, sums

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g_loss, _g_sums):
        (dz,) = ctx.saved_tensors
        g = (dz.float() * g_loss.float()).to(ctx.z_dtype)
        return g, None, None, None, None, None, None, None


class _BottleneckStep(torch.autograd.Function):
    """encode -> MMD -> decode as ONE autograd node (the three classes above, back to back).

    Same kernels, same numbers; what it saves is host time: a Python autograd.Function costs ~40 us of
    bookkeeping per apply and per backward, which is most of an eager step at the reference's small batches
    (BASELINE cfg1/cfg2).  Used by EncoderDecoderVAE.forward for the stock encoder/decoder pair when nothing
    else (autocast, hooks, a sharded regulariser, a given latent) asks for the separate nodes."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, h, We, be, Wd, bd, gamma, beta, prior, n_tokens: int, eps: float, strict_fp32: bool = False,
                grad_enabled: bool = True):
        ctx.set_materialize_grads(False)
        dt = h.dtype
        hc = h if h.stride(2) == 1 else h.contiguous()
        Wc, bc = We.contiguous(), be.contiguous()
        B, S, d = hc.shape
        L = Wc.shape[0]
        n_eff = min(int(n_tokens), S)
        N = B * n_eff
        dev = h.device
        lib = _lib.lib()
        stream = _stream(hc)
        code = _DTYPES[dt]
        has_lin, has_ln = Wd is not None, gamma is not None
        want_grad = bool(grad_enabled) and any(ctx.needs_input_grad[:3])
        ctx.small = False
        if (SMALL_STEP and 0 < N <= SMALL_STEP_MAX_ROWS and dt == torch.float32 and has_lin and has_ln and prior.dtype == torch.float32
                and not strict_fp32 and lib.tvae_small_step_supported(B, n_eff, d, L) and hc.stride(0) % 4 == 0
                and hc.stride(1) % 4 == 0
                and all(t.data_ptr() % 16 == 0 for t in (hc, Wc, bc, Wd, bd, gamma, beta) if t.is_contiguous())):
            # the reference's own batch sizes: the whole forward in two launches (csrc/small_step.hip)
            Wdc, bdc, gc, btc = Wd.contiguous(), bd.contiguous(), gamma.contiguous(), beta.contiguous()
            ya = prior.reshape(N, L).contiguous()
            z = torch.empty((B, n_eff, L), dtype=dt, device=dev)
            recon = torch.empty((N, d), dtype=dt, device=dev)
            pre = torch.empty((N, d), dtype=dt, device=dev)
            mean = torch.empty((N,), dtype=torch.float32, device=dev)
            rstd = torch.empty((N,), dtype=torch.float32, device=dev)
            sums = torch.empty(3, dtype=torch.float64, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            dz_mmd = torch.empty((N, L), dtype=dt, device=dev) if want_grad else None
            loss_scale, grad_scale = mmd_scales(B, n_eff, N, L)
            ws = _workspace(lib.tvae_small_step_fwd_workspace_bytes(B, n_eff, d, L), dev)
            _lib.check(lib.tvae_small_step_fwd(_ptr(hc), hc.stride(0), hc.stride(1), _ptr(Wc), _ptr(bc), _ptr(Wdc), _ptr(bdc),
                                               _ptr(gc), _ptr(btc), float(eps), _ptr(ya), B, n_eff, d, L, code, _ptr(z),
                                               _ptr(recon), _ptr(pre), _ptr(mean), _ptr(rstd), _ptr(dz_mmd), _ptr(sums),
                                               _ptr(loss), float(loss_scale), float(grad_scale), _ptr(ws), ws.numel(), stream),
                       "tvae_small_step_fwd")
            ctx.save_for_backward(hc, Wc, z, Wdc, gc, pre, mean, rstd, dz_mmd)
            ctx.meta = (B, S, d, L, n_eff, dt)
            ctx.small = True
            return z, recon.view(B, n_eff, d), loss
        # --- encode (autoencoders.py:21-22)
        z = torch.empty((B, n_eff, L), dtype=dt, device=dev)
        ws = _workspace(lib.tvae_encode_fwd_workspace_bytes(B, n_eff, d, L), dev)
        _lib.check(lib.tvae_encode_fwd(_ptr(hc), hc.stride(0), hc.stride(1), _ptr(Wc), _ptr(bc), _ptr(z),
                                       B, n_eff, d, L, code, _ptr(ws), ws.numel(), stream), "tvae_encode_fwd")
        # --- regulariser (autoencoders.py:110-112, 117-136); bf16 latents against an fp32 prior promote to fp32
        mdt = torch.bfloat16 if (dt == torch.bfloat16 and prior.dtype == torch.bfloat16) else torch.float32
        za = z.reshape(N, L).to(mdt)
        ya = prior.reshape(N, L).to(mdt).contiguous()
        loss_scale, grad_scale = mmd_scales(B, n_eff, N, L)
        mcode = _mmd_code(mdt, strict_fp32)
        ws_bytes = lib.tvae_mmd_workspace_bytes_for(N, N, L, int(want_grad), mcode)
        if ws_bytes < 0:
            raise RuntimeError(f"MMD regulariser: unsupported problem size (n_total={N}, D={L})")
        ws = _workspace(ws_bytes, dev)
        sums = torch.empty(3, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dz_mmd = torch.empty((N, L), dtype=mdt, device=dev) if want_grad else None
        _lib.check(lib.tvae_mmd_fwd_bwd(_ptr(za), _ptr(ya), N, 0, N, L, mcode, _ptr(sums), _ptr(loss),
                                        float(loss_scale), _ptr(dz_mmd), float(grad_scale), _ptr(ws), ws.numel(),
                                        stream), "tvae_mmd_fwd_bwd")
        # --- decode (autoencoders.py:38-39, 66-67)
        Wdc = Wd.contiguous() if has_lin else None
        bdc = bd.contiguous() if has_lin else None
        gc = gamma.contiguous() if has_ln else None
        btc = beta.contiguous() if has_ln else None
        recon = torch.empty((N, d), dtype=dt, device=dev)
        pre = torch.empty((N, d), dtype=dt, device=dev) if (has_lin and has_ln) else None
        mean = torch.empty((N,), dtype=torch.float32, device=dev) if has_ln else None
        rstd = torch.empty((N,), dtype=torch.float32, device=dev) if has_ln else None
        ws = _workspace(lib.tvae_decode_fwd_workspace_bytes(N, L, d), dev)
        _lib.check(lib.tvae_decode_fwd(_ptr(z), _ptr(Wdc), _ptr(bdc), _ptr(gc), _ptr(btc), float(eps), _ptr(recon),
                                       _ptr(pre), _ptr(mean), _ptr(rstd), N, L, d, code, _ptr(ws), ws.numel(),
                                       stream), "tvae_decode_fwd")
        ctx.save_for_backward(hc, Wc, z, Wdc, gc, pre, mean, rstd, dz_mmd)
        ctx.meta = (B, S, d, L, n_eff, dt)
        return z, recon.view(B, n_eff, d), loss

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g_latent, g_recon, g_reg):
        hc, Wc, z, Wdc, gc, pre, mean, rstd, dz_mmd = ctx.saved_tensors
        B, S, d, L, n_eff, dt = ctx.meta
        N = B * n_eff
        need = ctx.needs_input_grad
        dev = hc.device
        lib = _lib.lib()
        stream = _stream(hc)
        code = _DTYPES[dt]
        has_lin, has_ln = Wdc is not None, gc is not None
        want_z = any(need[:3])
        dz = dWd = dbd = dgamma = dbeta = None
        if (ctx.small and g_recon is not None and g_latent is None and g_recon.dtype == dt
                and (g_reg is None or (dz_mmd is not None and g_reg.dtype == torch.float32 and g_reg.numel() == 1))):
            # the usual training step at the reference's batch sizes: the whole backward in two launches
            dc = g_recon.reshape(N, d).contiguous()
            if dc.data_ptr() % 16 == 0:
                mk = lambda shape, want: torch.empty(shape, dtype=dt, device=dev) if want else None
                dh, dWe, dbe = mk((B, S, d), need[0]), mk((L, d), need[1]), mk((L,), need[2])
                dWd, dbd, dgamma, dbeta = mk((d, L), need[3]), mk((d,), need[4]), mk((d,), need[5]), mk((d,), need[6])
                use_reg = g_reg is not None and dz_mmd is not None
                ws = _workspace(lib.tvae_small_step_bwd_workspace_bytes(B, n_eff, d, L), dev)
                _lib.check(lib.tvae_small_step_bwd(_ptr(dc), _ptr(g_reg) if use_reg else None, _ptr(z), _ptr(pre), _ptr(mean),
                                                   _ptr(rstd), _ptr(gc), _ptr(Wdc), _ptr(Wc), _ptr(hc), hc.stride(0),
                                                   hc.stride(1), _ptr(dz_mmd) if use_reg else None, _ptr(dh), _ptr(dWe),
                                                   _ptr(dbe), _ptr(dWd), _ptr(dbd), _ptr(dgamma), _ptr(dbeta), B, S, n_eff, d, L,
                                                   code, _ptr(ws), ws.numel(), stream), "tvae_small_step_bwd")
                return dh, dWe, dbe, dWd, dbd, dgamma, dbeta, None, None, None, None, None
        if g_recon is not None:
            dc = g_recon.to(dt).reshape(N, d).contiguous()
            dz = torch.empty((N, L), dtype=dt, device=dev) if want_z else None
            dWd = torch.empty((d, L), dtype=dt, device=dev) if (has_lin and need[3]) else None
            dbd = torch.empty((d,), dtype=dt, device=dev) if (has_lin and need[4]) else None
            want_ln = has_ln and (need[5] or need[6])
            dgamma = torch.empty((d,), dtype=dt, device=dev) if want_ln else None
            dbeta = torch.empty((d,), dtype=dt, device=dev) if want_ln else None
            ws = _workspace(lib.tvae_decode_bwd_workspace_bytes(N, L, d), dev)
            _lib.check(lib.tvae_decode_bwd(_ptr(dc), _ptr(z), _ptr(Wdc), _ptr(gc), _ptr(pre), _ptr(mean), _ptr(rstd),
                                           _ptr(dz), _ptr(dWd), _ptr(dbd), _ptr(dgamma), _ptr(dbeta), N, L, d, code,
                                           _ptr(ws), ws.numel(), stream), "tvae_decode_bwd")
        dh = dWe = dbe = None
        if (want_z and dz is not None and dz_mmd is not None and g_reg is not None and g_latent is None
                and dz_mmd.dtype == dt and g_reg.dtype == torch.float32 and g_reg.numel() == 1):
            # the usual training step: decoder's share + w * regulariser's share, added inside the tanh-backward pass
            dh = torch.empty((B, S, d), dtype=dt, device=dev) if need[0] else None
            dWe = torch.empty((L, d), dtype=dt, device=dev) if need[1] else None
            dbe = torch.empty((L,), dtype=dt, device=dev) if need[2] else None
            ws = _workspace(lib.tvae_encode_bwd_workspace_bytes(B, n_eff, d, L), dev)
            _lib.check(lib.tvae_encode_bwd_sum(_ptr(dz), _ptr(dz_mmd), _ptr(g_reg), _ptr(z), _ptr(hc), hc.stride(0),
                                               hc.stride(1), _ptr(Wc), _ptr(dh), _ptr(dWe), _ptr(dbe), B, S, n_eff, d, L,
                                               code, _ptr(ws), ws.numel(), stream), "tvae_encode_bwd_sum")
        elif want_z:
            # d latent = decoder's share + the regulariser's (scaled by its upstream gradient) + a direct one
            if g_reg is not None and dz_mmd is not None:
                gm = g_reg.to(dz_mmd.dtype)
                if dz is None:
                    dz = (dz_mmd * gm).to(dt)
                elif dz_mmd.dtype == dt:
                    dz = torch.addcmul(dz, dz_mmd, gm)
                else:
                    dz = torch.addcmul(dz.to(dz_mmd.dtype), dz_mmd, gm).to(dt)
            if g_latent is not None:
                gl = g_latent.to(dt).reshape(N, L)
                dz = gl.contiguous() if dz is None else dz + gl
            if dz is not None:
                dh = torch.empty((B, S, d), dtype=dt, device=dev) if need[0] else None
                dWe = torch.empty((L, d), dtype=dt, device=dev) if need[1] else None
                dbe = torch.empty((L,), dtype=dt, device=dev) if need[2] else None
                ws = _workspace(lib.tvae_encode_bwd_workspace_bytes(B, n_eff, d, L), dev)
                _lib.check(lib.tvae_encode_bwd(_ptr(dz), _ptr(z), _ptr(hc), hc.stride(0), hc.stride(1), _ptr(Wc),
                                               _ptr(dh), _ptr(dWe), _ptr(dbe), B, S, n_eff, d, L, code,
                                               _ptr(ws), ws.numel(), stream), "tvae_encode_bwd")
        return dh, dWe, dbe, dWd, dbd, dgamma, dbeta, None, None, None, None, None


class _Upsample(torch.autograd.Function):
    """HF funnel `upsample` as used on reconstructed_encoding (reference: model.py:265-271)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, stride: int, target_len: int, separate_cls: bool, truncate_seq: bool):
        _require_device(x, "upsample")
        if x.dim() != 3:
            raise ValueError(f"upsample expects (batch, seq, hidden), got {tuple(x.shape)}")
        dt = x.dtype if x.dtype in _DTYPES else torch.float32
        xc = x.to(dt).contiguous()
        B, n, d = xc.shape
        lib = _lib.lib()
        out_len = lib.tvae_upsample_out_len(n, stride, target_len, int(separate_cls), int(truncate_seq))
        if out_len <= 0:
            raise ValueError(f"upsample: empty output for n={n}, stride={stride}, target_len={target_len}")
        out = torch.empty((B, out_len, d), dtype=dt, device=x.device)
        if B * d > 0:
            _lib.check(lib.tvae_upsample_fwd(_ptr(xc), _ptr(out), B, n, d, stride, target_len, int(separate_cls),
                                             int(truncate_seq), _DTYPES[dt], _stream(xc)), "tvae_upsample_fwd")
        ctx.meta = (B, n, d, stride, target_len, int(separate_cls), int(truncate_seq), dt, x.dtype)
        return out if dt == x.dtype else out.to(x.dtype)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        B, n, d, stride, target_len, sep, trunc, dt, x_dtype = ctx.meta
        gc = g.to(dt).contiguous()
        dx = torch.empty((B, n, d), dtype=dt, device=g.device)
        if B * d > 0:
            _lib.check(_lib.lib().tvae_upsample_bwd(_ptr(gc), _ptr(dx), B, n, d, stride, target_len, sep, trunc,
                                                    _DTYPES[dt], _stream(gc)), "tvae_upsample_bwd")
        return (dx if dt == x_dtype else dx.to(x_dtype)), None, None, None, None


class _RbfKernel(torch.autograd.Function):
    """K[i, j] = exp(-mean_c((x_i - y_j)^2) / dim), materialised (reference: autoencoders.py:117-126)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, y):
        _require_device(x, "_compute_kernel")
        _require_device(y, "_compute_kernel")
        if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
            raise ValueError(f"_compute_kernel expects (Nx, D) and (Ny, D), got {tuple(x.shape)} and {tuple(y.shape)}")
        out_dtype = torch.promote_types(x.dtype, y.dtype)
        xc, yc = x.detach().float().contiguous(), y.detach().float().contiguous()
        nx, dim = xc.shape
        ny = yc.shape[0]
        K = torch.empty((nx, ny), dtype=torch.float32, device=x.device)
        if nx * ny * dim > 0:
            _lib.check(_lib.lib().tvae_rbf_kernel_fwd(_ptr(xc), _ptr(yc), nx, ny, dim, _ptr(K), _stream(xc)), "tvae_rbf_kernel_fwd")
        elif nx * ny > 0:
            K.fill_(float("nan"))            # the reference takes the mean over an empty feature axis: nan
        ctx.save_for_backward(xc, yc, K)
        ctx.dtypes = (x.dtype, y.dtype)
        return K if out_dtype == torch.float32 else K.to(out_dtype)

    @staticmethod
    @_on_tensor_device
    def backward(ctx, gK):
        xc, yc, K = ctx.saved_tensors
        nx, dim = xc.shape
        ny = yc.shape[0]
        need_x, need_y = ctx.needs_input_grad
        g = gK.float().contiguous()
        dx = torch.zeros_like(xc) if need_x else None
        dy = torch.zeros_like(yc) if need_y else None
        if nx * ny * dim > 0:
            _lib.check(_lib.lib().tvae_rbf_kernel_bwd(_ptr(xc), _ptr(yc), _ptr(K), _ptr(g), nx, ny, dim, _ptr(dx), _ptr(dy),
                                                      _stream(xc)), "tvae_rbf_kernel_bwd")
        cast = lambda t, to: None if t is None else (t if t.dtype == to else t.to(to))
        return cast(dx, ctx.dtypes[0]), cast(dy, ctx.dtypes[1])


# ------------------------------------------------------------------------------------------
# functional API used by the modules in autoencoders.py
# ------------------------------------------------------------------------------------------

def latent_encode(h, weight, bias, n_tokens: int):
    return _LatentEncode.apply(h, weight, bias, int(n_tokens))


def latent_decode(z, weight=None, bias=None, ln_weight=None, ln_bias=None, eps: float = 1e-5):
    return _LatentDecode.apply(z, weight, bias, ln_weight, ln_bias, float(eps))


def rbf_kernel(x, y):
    """The (Nx, Ny) kernel matrix of the reference's ``_compute_kernel`` (autoencoders.py:117-126), differentiable."""
    return _RbfKernel.apply(x, y)


def mmd_scales(batch_size: int, n_latents: int, n_total: int, dim: int):
    """(loss_scale, grad_scale) for reg = MMD / batch * n_latents (autoencoders.py:112):
    reg = loss_scale * (S_yy + S_zz - 2 S_zy),  d reg/d z = grad_scale * [...]."""
    loss_scale = n_latents / (batch_size * float(n_total) ** 2)
    grad_scale = 4.0 * n_latents / (batch_size * float(n_total) ** 2 * float(dim) ** 2)
    return loss_scale, grad_scale


def mmd_reg_loss(latent2d, prior, batch_size: int, n_latents: int, strict_fp32: bool = False):
    """Single-device regulariser: MMD(prior, latent) / batch_size * n_latents, fp32 scalar."""
    n_total, dim = latent2d.shape
    ls, gs = mmd_scales(batch_size, n_latents, n_total, dim)
    loss, _ = _MMDRegulariser.apply(latent2d, latent2d, prior, 0, ls, gs, bool(strict_fp32), torch.is_grad_enabled())
    return loss


def mmd_row_block(z_rows, z_all, y_all, row_offset: int, loss_scale: float, grad_scale: float, strict_fp32: bool = False):
    """Row-sharded building block: returns (local loss share, float64[3] partial sums)."""
    return _MMDRegulariser.apply(z_rows, z_all, y_all, int(row_offset), float(loss_scale), float(grad_scale),
                                 bool(strict_fp32), torch.is_grad_enabled())


class HipShardStages:
    """The three stages of a row-sharded MMD evaluation over the C ABI (include/tvae_hip.h: tvae_mmd_shard_layout,
    tvae_mmd_stats_local, tvae_mmd_prep_local, tvae_mmd_tiles).  ``distributed.sharded_mmd_reg_loss`` owns the collectives
    between them; a test may hand it another object with these four methods to rehearse them without a GPU."""

    @staticmethod
    def layout(n_total: int, n_rows: int, dim: int, want_grad: bool, code: int):
        """The 16 layout words, or None when the shape has no sharded form (latent_size > 1024)."""
        import ctypes
        if dim > 1024:
            return None
        out = (ctypes.c_int64 * 16)()
        _lib.check(_lib.lib().tvae_mmd_shard_layout(n_total, n_rows, dim, int(want_grad), code, ctypes.addressof(out)),
                   "tvae_mmd_shard_layout")
        return [int(v) for v in out]

    @staticmethod
    def workspace(nbytes: int, device):
        return _workspace(nbytes, device)

    @staticmethod
    @_on_tensor_device
    def stats(z, y, n_total, row_offset, code, want_grad, ws):
        n_rows, dim = z.shape
        _lib.check(_lib.lib().tvae_mmd_stats_local(_ptr(z), _ptr(y), n_total, row_offset, n_rows, dim, code, int(want_grad),
                                                   _ptr(ws), ws.numel(), _stream(z)), "tvae_mmd_stats_local")

    @staticmethod
    @_on_tensor_device
    def prep(z, y, n_total, row_offset, code, want_grad, ws):
        n_rows, dim = z.shape
        _lib.check(_lib.lib().tvae_mmd_prep_local(_ptr(z), _ptr(y), n_total, row_offset, n_rows, dim, code, int(want_grad),
                                                  _ptr(ws), ws.numel(), _stream(z)), "tvae_mmd_prep_local")

    @staticmethod
    @_on_tensor_device
    def tiles(z, n_total, row_offset, code, sums, loss, loss_scale, dz, grad_scale, phase, ws):
        n_rows, dim = z.shape
        _lib.check(_lib.lib().tvae_mmd_tiles(_ptr(z), n_total, row_offset, n_rows, dim, code, _ptr(sums), _ptr(loss),
                                             float(loss_scale), _ptr(dz), float(grad_scale), int(phase), _ptr(ws), ws.numel(),
                                             _stream(z)), "tvae_mmd_tiles")


def mmd_code(dtype: torch.dtype, strict_fp32: bool) -> int:
    """The C ABI's dtype word of an MMD call."""
    return _mmd_code(dtype, strict_fp32)


def bottleneck_step(h, We, be, Wd, bd, gamma, beta, prior, n_tokens: int, eps: float, strict_fp32: bool = False):
    """(latent, reconstructed_encoding, reg_loss) of the stock bottleneck as one autograd node."""
    return _BottleneckStep.apply(h, We, be, Wd, bd, gamma, beta, prior, int(n_tokens), float(eps), bool(strict_fp32),
                                 torch.is_grad_enabled())


def upsample(x, stride: int, target_len: int, separate_cls: bool = True, truncate_seq: bool = False):
    """Drop-in for transformers.models.funnel.modeling_funnel.upsample (model.py:10, 265-271)."""
    if stride == 1:
        return x
    return _Upsample.apply(x, int(stride), int(target_len), bool(separate_cls), bool(truncate_seq))


def slerp(ratio, t1, t2):
    """Drop-in for transformer_vae.utils.slerp (utils.py:17-34): `ratio` is a float or a (rows,) tensor;
    t1, t2 are (rows, dim).  Forward only — the result carries no autograd graph."""
    return _slerp(t1, t2, ratio)


@_on_tensor_device
def _slerp(t1, t2, ratio):
    _require_device(t1, "slerp")
    if t1.dim() != 2 or t1.shape != t2.shape:
        raise ValueError(f"slerp expects two (rows, dim) tensors of equal shape, got {tuple(t1.shape)} and {tuple(t2.shape)}")
    dt = t1.dtype if t1.dtype in _DTYPES else torch.float32
    a, b = t1.detach().to(dt).contiguous(), t2.detach().to(dt).contiguous()
    rows, dim = a.shape
    out = torch.empty_like(a)
    if torch.is_tensor(ratio) and ratio.numel() > 1:
        if ratio.numel() != rows:
            raise ValueError(f"slerp: {ratio.numel()} ratios for {rows} rows")
        r = ratio.detach().to(device=a.device, dtype=torch.float32).reshape(-1).contiguous()
        rp, rs = _ptr(r), 0.0
    else:
        rp, rs = None, float(ratio)
    if rows > 0:
        _lib.check(_lib.lib().tvae_slerp(_ptr(a), _ptr(b), rp, rs, _ptr(out), rows, dim, _DTYPES[dt], _stream(a)),
                   "tvae_slerp")
    return out if dt == t1.dtype else out.to(t1.dtype)
