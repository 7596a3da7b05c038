"""The C-ABI library loads and exports every symbol include/bevrender_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from bevrender_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bevrender_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bevr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    L.bevr_abi_version.restype = ctypes.c_int
    assert L.bevr_abi_version() == _lib.ABI_VERSION == 6


def test_argument_contract_is_checked_without_a_gpu():
    """Bad descriptors / NULL pointers are rejected before anything is launched."""
    L = _lib.lib()
    d = _lib.AttnDesc()
    assert L.bevr_attn_table_dims(ctypes.byref(d)) == -2            # S < 2
    d.S, d.Wt = 200, 1999
    assert L.bevr_attn_table_dims(ctypes.byref(d)) == 0
    assert (d.Sp, d.Ht, d.Hp, d.y_off, d.x_off, d.Wp) == (224, 399, 851, 226, 1003, 4006)
    d.n_prob, d.q_div, d.heads, d.groups, d.N, d.Np, d.precision = 24, 6, 2, 1, 100000, 100032, 1
    assert L.bevr_attn_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None) == -1
    d.Np = 100000                                                    # not a multiple of 64
    assert L.bevr_attn_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None) == -2
    assert L.bevr_sample_fwd(None, None, None, 1, 4, 4, 8, 4, None) == -1
    d.Np = 100032
    assert L.bevr_attn_key_ws_bytes(ctypes.byref(d)) == 24 * 100032 * 16 + 24 * (100032 // 32) * 16 * (1 + 8)
    assert L.bevr_attn_key_prep(ctypes.byref(d), None, None, None, None) == -1
    assert L.bevr_dwconv_fwd(None, None, None, None, 1, 4, 4, 8, 3, 1, 0, None) == -1
    one = ctypes.c_void_p(16)                                        # never dereferenced: the shape check comes first
    assert L.bevr_dwconv_fwd(one, one, None, one, 1, 4, 4, 8, 4, 1, 0, None) == -2     # even kernel size
    assert L.bevr_dwconv_bwd_w(one, one, one, None, 1, 4, 4, 8, 7, 0, None) == -2       # k > 5
    assert b"contract" in L.bevr_strerror(-2)


def test_ops_refuse_cpu_tensors():
    import torch
    from bevrender_amd import ops
    with pytest.raises(_lib.BevrError):
        ops.sample_features(torch.zeros(1, 4, 4, 4), torch.zeros(1, 3, 2), 1)
