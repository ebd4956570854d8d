"""ctypes binding of libtvae_hip.so (C ABI: include/tvae_hip.h).

The library is built in-tree by ``build()`` (``make`` in csrc/, hipcc --offload-arch=gfx950)
and loaded lazily.  There is NO fallback: if the shared object is missing or a call fails
the caller gets a RuntimeError — the product path never routes through a CPU
implementation.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libtvae_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tvae_hip.h")

TVAE_F32, TVAE_BF16, TVAE_F32_STRICT = 0, 1, 2      # TVAE_F32_STRICT: tvae_mmd_* only (include/tvae_hip.h)

_i64, _vp, _int, _f32, _f64 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); mirrors include/tvae_hip.h one to one
SIGNATURES = {
    "tvae_version": (_int, []),
    "tvae_last_error": (ctypes.c_char_p, []),
    "tvae_encode_fwd_workspace_bytes": (_i64, [_i64] * 4),
    "tvae_encode_fwd": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp]),
    "tvae_encode_bwd_workspace_bytes": (_i64, [_i64] * 4),
    "tvae_encode_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int,
                               _vp, _i64, _vp]),
    "tvae_encode_bwd_sum": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                   _int, _vp, _i64, _vp]),
    "tvae_decode_fwd_workspace_bytes": (_i64, [_i64] * 3),
    "tvae_decode_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _i64,
                               _vp]),
    "tvae_decode_bwd_workspace_bytes": (_i64, [_i64] * 3),
    "tvae_decode_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int,
                               _vp, _i64, _vp]),
    "tvae_mmd_workspace_bytes": (_i64, [_i64, _i64, _i64, _int]),
    "tvae_mmd_workspace_bytes_for": (_i64, [_i64, _i64, _i64, _int, _int]),
    "tvae_profile_next_mmd": (_int, [_vp, _vp]),
    "tvae_upsample_out_len": (_i64, [_i64, _i64, _i64, _int, _int]),
    "tvae_upsample_fwd": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _vp]),
    "tvae_upsample_bwd": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _vp]),
    "tvae_slerp": (_int, [_vp, _vp, _vp, _f32, _vp, _i64, _i64, _int, _vp]),
    "tvae_rbf_kernel_fwd": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "tvae_rbf_kernel_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "tvae_mmd_fwd_bwd": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _f64, _vp, _f64, _vp, _i64, _vp]),
    "tvae_mmd_shard_layout": (_int, [_i64, _i64, _i64, _int, _int, _vp]),
    "tvae_mmd_stats_local": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp]),
    "tvae_mmd_prep_local": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp]),
    "tvae_mmd_tiles": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _f64, _vp, _f64, _int, _vp, _i64, _vp]),
    "tvae_mmd_fwd_bwd_static": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _f64, _vp, _f64, _vp]),
    "tvae_small_step_supported": (_int, [_i64] * 4),
    "tvae_small_step_fwd_workspace_bytes": (_i64, [_i64] * 4),
    "tvae_small_step_bwd_workspace_bytes": (_i64, [_i64] * 4),
    # h, stride_b, stride_s, We, be, Wd, bd, gamma, beta, eps, prior, B, n_eff, d, L, dtype, latent, recon, pre, mean, rstd,
    # d_latent_mmd, sums3, loss, loss_scale, grad_scale, workspace, bytes, stream
    "tvae_small_step_fwd": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i64, _i64, _i64, _i64, _int,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _i64, _vp]),
    # d_recon, d_reg, latent, pre, mean, rstd, gamma, Wd, We, h, stride_b, stride_s, d_latent_mmd, dh, dWe, dbe, dWd, dbd,
    # dgamma, dbeta, B, S, n_eff, d, L, dtype, workspace, bytes, stream
    "tvae_small_step_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp]),
}

_lock = threading.Lock()
_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip into libtvae_hip.so for gfx950.  Works without a GPU."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libtvae_hip.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: the HIP extension has not been built "
                        "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C transformer_vae_amd/csrc`). "
                        "There is no CPU fallback for this path.")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().tvae_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else 'unknown error'}")
