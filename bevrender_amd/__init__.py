"""bevrender_amd: MI355X-native (gfx950) BEV-lift + correlation hot path behind the reference's module API.

Sub-packages mirror the reference's import paths (`model.*`, `loss.*`); see INTEGRATION.md for how a
reference checkout imports them.  The arithmetic lives in csrc/ (HIP) behind include/bevrender_hip.h.
"""
import os

from . import _lib  # noqa: F401

__all__ = ["default_precision", "_lib"]


def default_precision() -> int:
    """BEVRENDER_PRECISION=bf16x3|f32|bf16|f16.  Default bf16x3 (round 4): f32 storage and per-pair
    arithmetic, matrix products as three split-bf16 MFMAs -- the reference's fp32 results to ~1e-5 (the tests hold it
    to the exact mode's limits) at twice the exact mode's speed.  f32: exact-f32 MFMA; bf16: the throughput mode (the
    benchmark's); f16: fp16 operands."""
    v = os.environ.get("BEVRENDER_PRECISION", "bf16x3").lower()
    if v in ("bf16", "bfloat16"):
        return _lib.PREC_BF16
    if v in ("f16", "fp16", "float16", "half"):
        return _lib.PREC_F16
    if v in ("f32", "fp32", "float32"):
        return _lib.PREC_F32
    if v in ("bf16x3", "f32s", "split"):
        return _lib.PREC_BF16X3
    raise ValueError(f"BEVRENDER_PRECISION={v!r}: expected f32, bf16x3, bf16 or f16")


def resolve_precision(p) -> int:
    if p is None:
        return default_precision()
    if isinstance(p, str):
        return {"f32": _lib.PREC_F32, "fp32": _lib.PREC_F32, "float32": _lib.PREC_F32,
                "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16, "bf16x3": _lib.PREC_BF16X3,
                "f16": _lib.PREC_F16, "fp16": _lib.PREC_F16, "float16": _lib.PREC_F16, "half": _lib.PREC_F16}[p.lower()]
    return int(p)
