"""Error behaviour of the C ABI (include/bevrender_hip.h: "a negative BEVR_E_* code when the arguments violate the stated
contract (nothing is launched)").  Every attention entry-point family is called with a bad descriptor, a NULL pointer, a
misaligned pointer and an unsupported precision; the checks come before any launch, so this runs without a GPU (fake
non-NULL pointers are never dereferenced)."""
import ctypes as C

import pytest

from bevrender_amd import _lib, ops

E_NULL, E_SHAPE, E_PRECISION, E_ALIGN = -1, -2, -3, -4
P = C.c_void_p(0x10000)          # a 16-byte aligned, never dereferenced "device pointer"
ODD = C.c_void_p(0x10004)        # misaligned
NUL = None


def desc(**over):
    g = ops.AttnGeom(n_prob=2, q_div=1, heads=2, groups=1, S=12, N=100, Wt=71, precision=_lib.PREC_BF16)
    d = g.desc()
    for k, v in over.items():
        setattr(d, k, v)
    return d


@pytest.fixture(scope="module")
def L():
    return _lib.lib()


def calls(L, d, p=P, q=P):
    """One call per attention entry point; `p` stands for the first operand pointer, `q` for the first output."""
    r = C.byref(d)
    return {
        "key_prep": lambda: L.bevr_attn_key_prep(r, p, P, q, None),
        "fwd": lambda: L.bevr_attn_fwd(r, p, P, P, P, P, q, P, None),
        "bwd_q": lambda: L.bevr_attn_bwd_q(r, p, P, P, P, P, P, P, P, P, P, q, P, None),
        "bwd_k": lambda: L.bevr_attn_bwd_k(r, p, P, P, P, P, P, P, P, P, P, P, P, q, P, P, P, None),
        "fwd_dropout": lambda: L.bevr_attn_fwd_dropout(r, p, P, P, P, P, q, P, 100, 1, None),
        "bwd_q_dropout": lambda: L.bevr_attn_bwd_q_dropout(r, p, P, P, P, P, P, P, P, P, P, q, P, 100, 1, None),
        "bwd_k_dropout": lambda: L.bevr_attn_bwd_k_dropout(r, p, P, P, P, P, P, P, P, P, P, P, P, q, P, P, P, 100, 1, None),
        "cell_fwd": lambda: L.bevr_attn_cell_fwd(r, p, P, P, P, P, None, None, q, P, None),
        "cell_bwd_q": lambda: L.bevr_attn_cell_bwd_q(r, p, P, P, P, P, P, P, P, P, P, q, P, None),
        "cell_bwd_k": lambda: L.bevr_attn_cell_bwd_k(r, p, P, P, P, P, P, P, P, P, P, P, q, P, P, P, None),
        "tap_prep": lambda: L.bevr_attn_tap_prep(r, p, P, P, P, q, None),
        "tap_fwd": lambda: L.bevr_attn_tap_fwd(r, p, P, P, P, q, P, None),
        "tap_bwd_q": lambda: L.bevr_attn_tap_bwd_q(r, p, P, P, P, q, P, None),
        "tap_bwd_k": lambda: L.bevr_attn_tap_bwd_k(r, p, P, P, P, q, P, P, P, None),
        "gather_fwd": lambda: L.bevr_attn_gather_fwd(r, p, P, P, P, P, P, q, P, P, None),
    }


NAMES = list(calls(None, desc()).keys()) if False else [
    "key_prep", "fwd", "bwd_q", "bwd_k", "fwd_dropout", "bwd_q_dropout", "bwd_k_dropout", "cell_fwd", "cell_bwd_q",
    "cell_bwd_k", "tap_prep", "tap_fwd", "tap_bwd_q", "tap_bwd_k", "gather_fwd"]


@pytest.mark.parametrize("name", NAMES)
def test_bad_descriptor_is_a_shape_error(L, name):
    for bad in (dict(Sp=16), dict(Np=101), dict(Ht=20), dict(heads=3, groups=2), dict(n_prob=3, q_div=2), dict(Hp=7)):
        assert calls(L, desc(**bad))[name]() == E_SHAPE, bad
    assert calls(L, desc(precision=9))[name]() == E_PRECISION


@pytest.mark.parametrize("name", NAMES)
def test_null_and_misaligned_pointers(L, name):
    assert calls(L, desc(), p=NUL)[name]() == E_NULL
    assert calls(L, desc(), q=NUL)[name]() == E_NULL
    if name not in ("key_prep", "tap_prep"):          # their first operands are plain float arrays (no alignment contract)
        assert calls(L, desc(), p=ODD)[name]() == E_ALIGN


def test_contract_limits_of_the_specialised_entry_points(L):
    # the tap entry points: 16-bit operand modes and one channel group only
    for name in ("tap_fwd", "tap_bwd_q", "tap_bwd_k"):
        assert calls(L, desc(precision=_lib.PREC_F32))[name]() == E_PRECISION
        assert calls(L, desc(heads=2, groups=2))[name]() == E_SHAPE
    assert calls(L, desc(heads=2, groups=2))["tap_prep"]() == E_SHAPE
    # the cell kernels run one wave per 32-row block of a BEV column: Sp <= 512
    big = ops.AttnGeom(n_prob=1, q_div=1, heads=1, groups=1, S=530, N=64, Wt=2 * 530 - 1, precision=_lib.PREC_BF16).desc()
    assert calls(L, big)["cell_fwd"]() == E_SHAPE
    # the gather forward: bf16 operands; a window column holds a key's taps for every BEV row of a column: S <= 224
    assert calls(L, desc(precision=_lib.PREC_F32))["gather_fwd"]() == E_PRECISION
    tall = ops.AttnGeom(n_prob=1, q_div=1, heads=1, groups=1, S=230, N=64, Wt=2 * 230 - 1, precision=_lib.PREC_BF16).desc()
    assert calls(L, tall)["gather_fwd"]() == E_SHAPE
    # a dropout threshold is a 16-bit number
    r = C.byref(desc())
    assert L.bevr_attn_fwd_dropout(r, P, P, P, P, P, P, P, 65536, 1, None) == E_SHAPE
    assert L.bevr_strerror(E_SHAPE) and L.bevr_strerror(E_ALIGN)


def test_workspace_sizes_of_a_bad_descriptor_are_zero(L):
    assert L.bevr_attn_key_ws_bytes(C.byref(desc(Sp=16))) == 0
    assert L.bevr_attn_tap_ws_bytes(C.byref(desc(Sp=16))) == 0
    assert L.bevr_attn_tap_ws_bytes(C.byref(desc())) > 0
