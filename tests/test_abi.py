"""The C-ABI library loads, exports every symbol include/coin_hip.h declares, and the Python binding table
mirrors the header.  No compute calls (no GPU here).  Also: the product has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from coin_amd import _lib

HEADER = open(_lib.HEADER_PATH).read()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    body = re.sub(r"typedef struct.*?}\s*\w+;", "", body, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*(?:const\s+)?(int|size_t|char\s*\*|const char\s*\*)\s*\*?\s*(coin_\w+)\s*\(([^;]*?)\)\s*;", body, flags=re.M | re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        out[m.group(2)] = [] if args == ["void"] else args
    return out


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    decl = declared_functions()
    assert len(decl) >= 19, decl.keys()
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in coin_hip.h but not exported by libcoin_hip.so"


def test_the_product_library_exports_exactly_the_declared_entry_points_and_reads_no_environment():
    """include/coin_hip.h: "keeps no global state".  The lab switches of tools/gemm_lab / tools/roibench.py (debug bits, variant hooks,
    COIN_CONV_* environment overrides) are compiled only with -DCOIN_LAB into tools/lab/: the product object must not export one, nor
    import getenv."""
    import subprocess

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("coin_")}
    assert exported == set(declared_functions()), exported ^ set(declared_functions())
    data = [l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "BD" and not l.split()[-1].startswith("__hip_")]
    assert not data, f"exported data symbols (state): {data}"
    assert "getenv" not in out


def test_a_library_of_another_abi_version_is_refused(lib, monkeypatch):
    from coin_amd._lib import CoinHipError

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(CoinHipError, match="C-ABI version"):
        _lib.lib()


def test_binding_table_matches_header(lib):
    decl = declared_functions()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in decl, f"{name} bound in _lib.py but not declared in coin_hip.h"
        assert len(argtypes) == len(decl[name]), f"{name}: {len(argtypes)} bound args vs {len(decl[name])} declared"
        for ct, d in zip(argtypes, decl[name]):
            is_ptr = "*" in d or "[" in d
            assert is_ptr == (ct in (ctypes.c_void_p,) or hasattr(ct, "contents") or ct is ctypes.POINTER(ctypes.c_float)), (name, d, ct)
            if not is_ptr:
                want = {"int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t}[d.split()[-2] if len(d.split()) > 1 else d]
                assert ct is want, (name, d, ct)
    unbound = set(decl) - set(_lib.SIGNATURES) - {"coin_abi_version", "coin_build_arch", "coin_clear_last_error", "coin_nms_workspace_bytes", "coin_conv_gemm_stats_bytes", "coin_conv_wgrad_workspace_bytes", "coin_conv_gemm_workspace_bytes",
                                                      "coin_window_attn_bwd_workspace_bytes"}
    assert not unbound, unbound


def test_version_arch_and_workspace_query(lib):
    m = re.search(r"#define\s+COIN_ABI_VERSION\s+(\d+)", HEADER)
    assert lib.coin_abi_version() == int(m.group(1)) == _lib.ABI_VERSION
    assert lib.coin_build_arch() == b"gfx950"
    assert lib.coin_nms_workspace_bytes(2, 12000) == 2 * 12000 * 188 * 8


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.SgdTensor) == 48 and ctypes.sizeof(_lib.EmaTensor) == 24
    assert ctypes.sizeof(_lib.WdTensor) == 32 and _lib.WdTensor.cout.offset == 16
    assert ctypes.sizeof(_lib.RoiLevel) == 24 and _lib.RoiLevel.H.offset == 8 and _lib.RoiLevel.spatial_scale.offset == 16


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected before any launch (safe to call without a device)."""
    assert lib.coin_roi_align_fwd(None, 1, 8, 4, 4, 1, None, 1, 2, 2, 1.0, 0, 1, None, 0, None) == -1
    assert lib.coin_gemm_nt(None, 8, None, 8, None, 8, 1, 1, 8, None, 0, 0.0, 0, 0, None) == -1
    assert lib.coin_mil_ce_fwd_bwd(None, 9, None, None, None, 4, 9, 1, 1, None, None, None) == -1
    assert lib.coin_nms_batched(None, None, 1, 20000, 0.5, 10, None, None, None, None) == -1
    assert lib.coin_conv_gemm_bf16(None, 64, 0, 0, 0, 0, None, 64, None, 8, None, 0, 256, 8, 64, None, 0, None) == -1
    assert lib.coin_conv_gemm_stats_bytes(401408, 512) == 3136 * 3 * 512 * 4   # sized for 128-row tiles (ABI 3)
    assert lib.coin_weight_dgrad_layout(None, 3, 8, None) == -1 and lib.coin_weight_dgrad_layout(None, 0, 0, None) == 0
    assert lib.coin_roi_align_fwd_levels(None, 4, 1, 8, None, None, 1, 7, 7, 0, 1, None, 1, None) == -1
    assert lib.coin_roi_align_bwd_level(None, 1, 8, 4, 4, None, None, 0, 1, 7, 7, 0.25, 0, 1, None, 1, None) == -1
    assert lib.coin_window_attn_bwd(None, None, None, None, None, None, None, 4, 1, 3, 49, 32, 0.17, None) == -1
    assert lib.coin_window_attn_bwd_workspace_bytes(3456, 3) == 341 * 3 * 64 * 64 * 4
    # round 4: the pooled-residual GEMM needs its residual; the three reductions that replaced float atomics need their workspaces
    x = ctypes.c_void_p(0x1000)   # any non-NULL, 16-byte-aligned address: rejected before it is dereferenced
    assert lib.coin_conv_gemm_bf16_rpool(x, 64, 0, 0, 0, 0, x, 64, x, 256, None, 256, 14, 14, 196, 256, 64, None, 0, None) == -1
    assert lib.coin_conv_gemm_bf16_rpool(x, 64, 0, 0, 0, 0, x, 64, x, 256, x, 256, 14, 14, 195, 256, 64, None, 0, None) == -1   # M % (h w)
    assert lib.coin_bias_act_bwd(x, x, x, 64, 8, 64, x, 1, 0.01, 1, None, None) == -1
    assert lib.coin_cosine_logits_bwd(x, x, 64, x, x, x, 8, 64, 9, 100.0, x, x, 1, None, None) == -1
    assert lib.coin_rpn_losses_fwd_bwd(x, x, x, x, x, 16, 16, 0, x, x, None, None, None, None) == -1


def test_product_has_no_cpu_fallback():
    from coin_amd import kernels as K
    from coin_amd import layers as L
    from coin_amd._lib import CoinHipError

    with pytest.raises(CoinHipError):
        K.roi_align_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(1, 5), (2, 2), 1.0)
    with pytest.raises(CoinHipError):
        K.gemm_nt(torch.zeros(4, 16), torch.zeros(4, 16))
    with pytest.raises(CoinHipError):
        L.mil_cross_entropy(torch.zeros(4, 9), labels=torch.zeros(4, dtype=torch.int64))
    with pytest.raises(CoinHipError):
        L.roi_align(torch.zeros(1, 8, 4, 4), torch.zeros(1, 5), (2, 2), 1.0)
