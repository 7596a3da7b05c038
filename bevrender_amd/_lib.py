"""ctypes binding of libbevrender_hip.so (C ABI declared in include/bevrender_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not on a
ROCm device the ops raise.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import torch  # noqa: F401  -- must be imported BEFORE the library is loaded: torch bundles its own HIP runtime
# (libamdhip64.so.7); loading ours first would bind the process to /opt/rocm's copy and torch would then fail with
# "no ROCm-capable device".  With torch loaded first both share torch's runtime.

_HERE = os.path.dirname(os.path.abspath(__file__))
# BEVRENDER_LIB: another build of the same library (tests use it for the -DBEVR_DEBUG build with in-kernel traps)
LIB_PATH = os.environ.get("BEVRENDER_LIB") or os.path.join(_HERE, "lib", "libbevrender_hip.so")
CSRC = os.path.join(_HERE, "csrc")

PREC_F32, PREC_BF16, PREC_F16, PREC_BF16X3 = 0, 1, 2, 3
ABI_VERSION = 6   # BEVR_ABI_VERSION of include/bevrender_hip.h this binding was written against

# every symbol include/bevrender_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "bevr_abi_version", "bevr_strerror", "bevr_attn_table_dims", "bevr_attn_key_ws_bytes", "bevr_attn_key_prep",
    "bevr_attn_fwd", "bevr_attn_bwd_q",
    "bevr_attn_bwd_k", "bevr_attn_fwd_dropout", "bevr_attn_bwd_q_dropout", "bevr_attn_bwd_k_dropout", "bevr_attn_cell_fwd", "bevr_attn_cell_bwd_q", "bevr_attn_cell_bwd_k", "bevr_attn_tap_ws_bytes", "bevr_attn_tap_prep", "bevr_attn_tap_fwd", "bevr_attn_tap_bwd_q", "bevr_attn_tap_bwd_k", "bevr_attn_gather_fwd", "bevr_attn_slab_ws_bytes", "bevr_attn_slab_prep", "bevr_attn_slab_bwd_q", "bevr_sample_fwd", "bevr_sample_bwd", "bevr_sample_fwd_bf16", "bevr_sample_bwd_bf16", "bevr_project_bev_grid", "bevr_project_bev_grid_masked", "bevr_corr_fwd",
    "bevr_corr_bwd", "bevr_recall_rank", "bevr_dwconv_fwd", "bevr_dwconv_bwd_w", "bevr_dwconv_res_gelu", "bevr_affine_warp_fwd", "bevr_affine_warp_bwd",
    "bevr_offset_head_fwd", "bevr_offset_head_bwd", "bevr_key_positions_fwd", "bevr_key_positions_bwd", "bevr_kv_project", "bevr_layernorm_fwd", "bevr_layernorm_bwd", "bevr_merge_views_fwd", "bevr_merge_views_bwd", "bevr_merge_tap_fwd", "bevr_merge_tap_bwd", "bevr_attn_bwd_prep", "bevr_pack_kv", "bevr_unpack_dkv",
]


class AttnDesc(C.Structure):
    """struct bevr_attn_desc (include/bevrender_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_prob", "q_div", "heads", "groups", "S", "Sp", "N", "Np", "Ht", "Wt", "Hp", "Wp",
        "y_off", "x_off", "precision", "reserved")]


class BevrError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into bevrender_amd/lib/libbevrender_hip.so."""
    cmd = ["make", "-C", CSRC, "-j", "4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise BevrError("building libbevrender_hip.so failed (see output above)")
    return LIB_PATH


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises BevrError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise BevrError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C bevrender_amd/csrc`.")
        L = C.CDLL(LIB_PATH)
        vp, fp, ip = C.c_void_p, C.c_void_p, C.c_int
        dp = C.POINTER(AttnDesc)
        L.bevr_abi_version.restype = C.c_int
        L.bevr_strerror.restype = C.c_char_p
        L.bevr_strerror.argtypes = [C.c_int]
        L.bevr_attn_table_dims.argtypes = [dp]
        L.bevr_attn_key_ws_bytes.argtypes = [dp]
        L.bevr_attn_key_prep.argtypes = [dp, fp, fp, vp, vp]
        L.bevr_attn_fwd.argtypes = [dp, vp, vp, vp, vp, fp, fp, fp, vp]
        L.bevr_attn_bwd_q.argtypes = [dp] + [vp] * 4 + [vp, fp] + [vp, fp, fp, fp, fp, fp, vp]
        L.bevr_attn_bwd_k.argtypes = [dp] + [vp] * 4 + [fp] * 3 + [vp, vp, fp, fp, fp] + [fp] * 4 + [vp]
        u32 = C.c_uint32
        L.bevr_attn_fwd_dropout.argtypes = [dp, vp, vp, vp, vp, fp, fp, fp, u32, u32, vp]
        L.bevr_attn_bwd_q_dropout.argtypes = [dp] + [vp] * 4 + [vp, fp] + [vp, fp, fp, fp, fp, fp, u32, u32, vp]
        L.bevr_attn_bwd_k_dropout.argtypes = [dp] + [vp] * 4 + [fp] * 3 + [vp, vp, fp, fp, fp] + [fp] * 4 + [u32, u32, vp]
        L.bevr_attn_cell_fwd.argtypes = [dp, vp, vp, vp, vp, fp, fp, fp, fp, fp, vp]
        L.bevr_attn_cell_bwd_q.argtypes = [dp] + [vp] * 4 + [vp, fp] + [vp, fp, fp, fp, fp, fp, vp]
        L.bevr_attn_cell_bwd_k.argtypes = [dp] + [vp] * 4 + [vp, fp] + [vp, vp, fp, fp, fp] + [fp] * 4 + [vp]
        L.bevr_attn_tap_ws_bytes.argtypes = [dp]
        L.bevr_attn_tap_prep.argtypes = [dp, fp, fp, fp, fp, vp, vp]
        L.bevr_attn_tap_fwd.argtypes = [dp, vp, vp, fp, fp, fp, vp, vp]
        L.bevr_attn_tap_bwd_q.argtypes = [dp, vp, vp, vp, fp, fp, fp, vp]
        L.bevr_attn_tap_bwd_k.argtypes = [dp, vp, vp, vp, fp, fp, fp, fp, fp, vp]
        L.bevr_attn_gather_fwd.argtypes = [dp, vp, vp, vp, vp, vp, fp, fp, fp, vp, vp]
        L.bevr_attn_slab_ws_bytes.argtypes = [dp]
        L.bevr_attn_slab_prep.argtypes = [dp, fp, fp, vp, vp, vp]
        L.bevr_attn_slab_bwd_q.argtypes = [dp, vp, vp, vp, vp, fp, vp, fp, fp, fp, fp, fp, vp]
        L.bevr_sample_fwd.argtypes = [fp, fp, fp] + [ip] * 5 + [vp]
        L.bevr_sample_bwd.argtypes = [fp] * 5 + [ip] * 5 + [vp]
        L.bevr_sample_fwd_bf16.argtypes = [vp, fp, fp] + [ip] * 5 + [vp]
        L.bevr_sample_bwd_bf16.argtypes = [vp] + [fp] * 4 + [ip] * 5 + [vp]
        L.bevr_project_bev_grid.argtypes = [fp] * 4 + [ip] * 4 + [vp]
        L.bevr_project_bev_grid_masked.argtypes = [fp] * 4 + [ip] * 4 + [vp] + [ip] * 3 + [vp]
        L.bevr_corr_fwd.argtypes = [fp] * 5 + [ip] * 4 + [vp]
        L.bevr_corr_bwd.argtypes = [fp] * 8 + [ip] * 4 + [vp]
        L.bevr_recall_rank.argtypes = [fp, vp, ip, vp]
        L.bevr_dwconv_fwd.argtypes = [fp, fp, fp, fp] + [ip] * 7 + [vp]
        L.bevr_dwconv_bwd_w.argtypes = [fp, fp, fp, fp] + [ip] * 6 + [vp]
        L.bevr_dwconv_res_gelu.argtypes = [fp] * 5 + [ip] * 6 + [vp]
        L.bevr_affine_warp_fwd.argtypes = [fp, fp, fp] + [ip] * 4 + [vp]
        L.bevr_affine_warp_bwd.argtypes = [fp, fp, fp] + [ip] * 4 + [vp]
        L.bevr_offset_head_fwd.argtypes = [fp] * 7 + [C.c_longlong] + [ip] * 4 + [C.c_float, vp]
        L.bevr_offset_head_bwd.argtypes = [fp] * 13 + [C.c_longlong] + [ip] * 4 + [C.c_float, vp]
        L.bevr_layernorm_fwd.argtypes = [fp] * 6 + [C.c_longlong, ip, C.c_float, vp]
        L.bevr_layernorm_bwd.argtypes = [fp] * 8 + [C.c_longlong, ip, vp]
        L.bevr_kv_project.argtypes = [vp, ip, fp, C.c_longlong, vp, fp] + [ip] * 9 + [vp] * 6 + [ip, vp]
        L.bevr_key_positions_fwd.argtypes = [fp, fp, vp, fp] + [ip] * 8 + [C.c_float, C.c_float, vp]
        L.bevr_key_positions_bwd.argtypes = [fp, fp, vp, fp, fp] + [ip] * 8 + [C.c_float, C.c_float, vp]
        L.bevr_pack_kv.argtypes = [fp, fp, C.c_longlong, C.c_longlong] + [ip] * 6 + [vp] * 5
        L.bevr_merge_views_fwd.argtypes = [fp] * 5 + [ip] * 6 + [vp]
        L.bevr_merge_views_bwd.argtypes = [fp] * 9 + [ip] * 6 + [vp]
        L.bevr_merge_tap_fwd.argtypes = [fp] * 7 + [ip] * 6 + [vp]
        L.bevr_merge_tap_bwd.argtypes = [fp] * 13 + [ip] * 6 + [vp]
        L.bevr_attn_bwd_prep.argtypes = [fp] * 5 + [vp, vp, fp, fp] + [ip] * 3 + [vp]
        L.bevr_unpack_dkv.argtypes = [fp] * 4 + [C.c_longlong, C.c_longlong] + [ip] * 5 + [vp]
        for name in SYMBOLS:
            fn = getattr(L, name)
            if name in ("bevr_attn_key_ws_bytes", "bevr_attn_tap_ws_bytes", "bevr_attn_slab_ws_bytes"):
                fn.restype = C.c_size_t
            elif name not in ("bevr_strerror",):
                fn.restype = C.c_int
        if L.bevr_abi_version() != ABI_VERSION:
            raise BevrError(f"libbevrender_hip.so ABI version {L.bevr_abi_version()} != {ABI_VERSION}: rebuild "
                            "(make -C bevrender_amd/csrc)")
        _lib = L
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().bevr_strerror(code).decode()
        raise BevrError(f"{what} failed: {msg} (code {code})")
