"""CPU: the C-ABI library loads, exports every symbol include/tvae_hip.h declares, and its
host-only entry points (version, workspace queries, argument validation) behave.  No kernel is
launched here."""
import ctypes
import os
import re

import pytest

from transformer_vae_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tvae_hip.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tvae_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in tvae_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signatures out of sync with the header"


def test_version_and_workspace_queries(lib):
    assert lib.tvae_version() == 1000
    # cfg4 per-GPU shard and whole problem; monotone in the problem size; negative for unsupported D
    small = lib.tvae_mmd_workspace_bytes(350, 350, 77, 1)
    big = lib.tvae_mmd_workspace_bytes(245760, 245760, 768, 1)
    shard = lib.tvae_mmd_workspace_bytes(245760, 30720, 768, 1)
    assert 0 < small < shard <= big
    # the dtype-aware query: the fp32-input kernels need a few GiB at cfg4; the matrix-core modes (bf16 inputs, and fp32
    # inputs as split fp16 pairs) add one gradient slab per column chunk (<= 32 GiB)
    f32 = lib.tvae_mmd_workspace_bytes_for(245760, 245760, 768, 1, 0)
    b16 = lib.tvae_mmd_workspace_bytes_for(245760, 245760, 768, 1, 1)
    strict = lib.tvae_mmd_workspace_bytes_for(245760, 245760, 768, 1, 2)
    assert 0 < strict < 8 * 2 ** 30 and strict <= big and 0 < b16 <= big < 40 * 2 ** 30 and 0 < f32 <= big
    assert lib.tvae_mmd_workspace_bytes_for(245760, 245760, 768, 1, 3) == -1
    assert lib.tvae_mmd_workspace_bytes_for(350, 350, 77, 1, 7) == -1
    assert lib.tvae_mmd_workspace_bytes(350, 350, 77, 0) <= small
    assert lib.tvae_mmd_workspace_bytes(100, 100, 5000, 1) > 0         # latent_size > 1024: strip-wise route
    assert lib.tvae_mmd_workspace_bytes(0, 0, 77, 1) == -1
    for fn, args in [(lib.tvae_encode_fwd_workspace_bytes, (70, 5, 768, 77)),
                     (lib.tvae_encode_bwd_workspace_bytes, (70, 5, 768, 77)),
                     (lib.tvae_decode_fwd_workspace_bytes, (350, 77, 768)),
                     (lib.tvae_decode_bwd_workspace_bytes, (350, 77, 768))]:
        assert fn(*args) >= 0


def test_argument_validation_returns_status_and_message(lib):
    # null pointers / bad shapes are rejected before any HIP call, so this runs without a GPU
    st = lib.tvae_mmd_fwd_bwd(None, None, 16, 0, 16, 8, _lib.TVAE_F32, None, None, 1.0, None, 1.0, None, 0, None)
    assert st == 1
    assert b"null pointer" in lib.tvae_last_error()
    st = lib.tvae_encode_fwd(None, 0, 0, None, None, None, 1, 1, 1, 1, 7, None, 0, None)
    assert st == 1 and b"dtype" in lib.tvae_last_error()
    st = lib.tvae_decode_fwd(ctypes.c_void_p(16), None, None, None, None, 1e-9, ctypes.c_void_p(16), None, None, None,
                             4, 8, 16, _lib.TVAE_F32, None, 0, None)
    assert st == 1 and b"identity decoder" in lib.tvae_last_error()
    with pytest.raises(RuntimeError, match="identity decoder"):
        _lib.check(st, "tvae_decode_fwd")
