"""Host side of the HIP kernels: geometry, packed layouts and autograd wiring.

Everything that is pure layout (NCHW <-> the kernels' packed orders, scaling by head_dim^-0.5 log2 e,
table transposition/padding) is ordinary differentiable torch code, so autograd undoes it for free;
the arithmetic of the hot path (feature sampling, QK^T + RPE bias + softmax + PV, correlation) runs in
libbevrender_hip.so through the C ABI in include/bevrender_hip.h.  There is no CPU implementation
here: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, replace as dc_replace
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib

LOG2E = 1.4426950408889634
HEAD_DIM = 32  # kernels' fixed head width (reference dims/heads is always 32); smaller heads are zero padded


def perm32(r: int) -> int:
    """bits 2 and 3 swapped: the order an MFMA accumulator tile is consumed in (bevr_common.h)."""
    return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)


_PERM_CACHE = {}


def perm_index(n: int, device) -> torch.Tensor:
    """index tensor idx with idx[p] = 32*(p//32) + perm32(p%32); n must be a multiple of 32."""
    key = (n, str(device))
    if key not in _PERM_CACHE:
        base = torch.arange(n, device=device)
        r = base % 32
        pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
        _PERM_CACHE[key] = (base - r + pr)
    return _PERM_CACHE[key]


@dataclass(frozen=True)
class AttnGeom:
    """Mirror of struct bevr_attn_desc with the derived paddings."""
    n_prob: int
    q_div: int
    heads: int
    groups: int
    S: int
    N: int
    Wt: int
    precision: int

    @property
    def Sp(self): return 32 * ((self.S + 31) // 32)
    @property
    def Mp(self): return self.S * self.Sp
    @property
    def Np(self): return 64 * ((self.N + 63) // 64)
    @property
    def Ht(self): return 2 * self.S - 1
    @property
    def y_off(self): return self.Sp + 2
    @property
    def Hp(self): return self.Ht + 2 * self.Sp + 4
    @property
    def x_off(self): return self.Wt // 2 + 4
    @property
    def Wp(self): return self.Wt + 2 * (self.Wt // 2) + 9
    @property
    def rx(self): return (self.Wt - 1) / (2.0 * (self.S - 1))

    def desc(self) -> _lib.AttnDesc:
        return _lib.AttnDesc(self.n_prob, self.q_div, self.heads, self.groups, self.S, self.Sp, self.N, self.Np,
                             self.Ht, self.Wt, self.Hp, self.Wp, self.y_off, self.x_off, self.precision, 0)


# --------------------------------------------------------------------------------------------------
# packed layouts (differentiable torch code; also exercised on CPU by tests/test_packing_cpu.py)
# --------------------------------------------------------------------------------------------------
def pack_query(query: torch.Tensor, heads: int) -> torch.Tensor:
    """(B, C, S, S) NCHW raw (layer-normed) query -> (B, h, Mp, 32), packed index j*Sp + i, scaled by
    c^-0.5 * log2(e).  The reference uses the raw query as Q (model/SCA_deform_attn.py:304-306)."""
    B, Cc, S, _ = query.shape
    c = Cc // heads
    if c > HEAD_DIM:
        raise ValueError(f"head_dim {c} > {HEAD_DIM} is not supported by the gfx950 kernels")
    Sp = 32 * ((S + 31) // 32)
    q = query.reshape(B, heads, c, S, S) * (c ** -0.5 * LOG2E)
    q = q.permute(0, 1, 4, 3, 2)                      # (B, h, j, i, c)
    q = F.pad(q, (0, HEAD_DIM - c, 0, Sp - S))
    return q.reshape(B, heads, S * Sp, HEAD_DIM)


def pack_keys(x: torch.Tensor, heads: int) -> torch.Tensor:
    """(B', N, C) projected keys or values -> (B', h, Np, 32) row layout (zero padded)."""
    Bp, N, Cc = x.shape
    c = Cc // heads
    Np = 64 * ((N + 63) // 64)
    k = x.reshape(Bp, N, heads, c).permute(0, 2, 1, 3)
    return F.pad(k, (0, HEAD_DIM - c, 0, Np - N))


def unpack_out(O: torch.Tensor, S: int, c: int) -> torch.Tensor:
    """(B', h, Mp, 32) -> (B', S*S, h*c) with the row index i*S + j (the reference's flattening)."""
    Bp, h, Mp, _ = O.shape
    Sp = Mp // S
    o = O.reshape(Bp, h, S, Sp, HEAD_DIM)[:, :, :, :S, :c]          # (B', h, j, i, c)
    return o.permute(0, 3, 2, 1, 4).reshape(Bp, S * S, h * c)


def unpack_out_views(O: torch.Tensor, S: int, c: int, views: int) -> torch.Tensor:
    """(B * views, h, Mp, 32) -> (B, S*S, views * h * c): the views of a sample side by side in the channel axis, view-major
    -- the layout `proj_out` of SCA contracts (reference model/SCA_deform_attn.py:415-420: the per-view outputs
    concatenated along channels, then a 1x1 convolution V C -> C) -- in ONE copy out of the packed layout (unpack_out
    followed by the reference's reshape / permute of the views was two: 2 x 491 MB per call at the benchmark)."""
    BV, h, Mp, _ = O.shape
    Sp = Mp // S
    o = O.reshape(BV // views, views, h, S, Sp, HEAD_DIM)[:, :, :, :, :S, :c]        # (B, v, h, j, i, c)
    return o.permute(0, 4, 3, 1, 2, 5).reshape(BV // views, S * S, views * h * c)


class _LinearRows(torch.autograd.Function):
    """linear_rows below: F.linear forward; the weight gradient with its long contraction split (see linear_rows)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        K, N = x.shape[-1], dy.shape[-1]
        d2, x2 = dy.reshape(-1, N), x.reshape(-1, K)
        R = d2.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (d2 @ weight).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            nfull = R // LINEAR_ROWS_CHUNK
            n0 = nfull * LINEAR_ROWS_CHUNK
            dw = torch.bmm(d2[:n0].reshape(nfull, LINEAR_ROWS_CHUNK, N).transpose(1, 2),
                           x2[:n0].reshape(nfull, LINEAR_ROWS_CHUNK, K)).sum(0)
            if n0 < R:
                dw = dw + d2[n0:].t() @ x2[n0:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = d2.sum(0)
        return dx, dw, db


LINEAR_ROWS_CHUNK = 4096     # rows per partial product of the weight gradient
LINEAR_ROWS_MIN = 65536      # below this many rows the stock single GEMM is as good


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.linear for the path's tall-and-thin products (rows = B S S = 320 000 BEV cells, 64 ... 384 channels: the 1x1
    convolutions of the layer MLPs, model/model_utils.py:51-59, and proj_out, model/SCA_deform_attn.py:415-420).  Same
    forward; in the backward the weight gradient dW = dY^T X -- a (N x rows) @ (rows x K) product whose output is a few
    tiles while its contraction is 320 000 long -- is computed as rows / 4096 partial products (one batched GEMM) and
    summed: the stock single GEMM runs it on 8 workgroups (0.7 ms per call at 64 x 256; 0.15 ms batched).  float32, same
    results up to the order of the sum."""
    if not x.is_cuda or x.dtype != torch.float32 or x.numel() // x.shape[-1] < LINEAR_ROWS_MIN or not x.is_contiguous():
        return F.linear(x, weight, bias)
    return _LinearRows.apply(x, weight, bias)


class _MergeViews(torch.autograd.Function):
    """merge_views below, through bevr_merge_views_fwd / _bwd (csrc/merge.hip)."""

    @staticmethod
    def forward(ctx, O_r, L_r, O_c, L_c, S, c, views):
        _require_gpu(O_r, L_r, O_c, L_c)
        L = _lib.lib()
        BV, h, Mp, hd = O_r.shape
        assert hd == HEAD_DIM and Mp % S == 0 and BV % views == 0
        two = O_c is not None
        O_r = O_r.float().contiguous()
        if two:
            O_c, L_r, L_c = O_c.float().contiguous(), L_r.float().contiguous(), L_c.float().contiguous()
            assert O_c.shape == O_r.shape and L_r.shape == L_c.shape == O_r.shape[:3]
        out = torch.empty(BV // views, S * S, views * h * c, device=O_r.device, dtype=torch.float32)
        nbytes = float((O_r.numel() * (2 if two else 1)) * S / (Mp // S) * c / HEAD_DIM * 4 + out.numel() * 4)
        _lib.check(KERNEL_TIMER.run("bevr_merge_views_fwd", 0.0, L.bevr_merge_views_fwd, _ptr(O_r),
                                    _ptr(L_r) if two else None, _ptr(O_c) if two else None, _ptr(L_c) if two else None,
                                    _ptr(out), BV, views, h, S, Mp // S, c, _stream(), nbytes=nbytes),
                   "bevr_merge_views_fwd")
        ctx.save_for_backward(*((O_r, L_r, O_c, L_c) if two else ()))
        ctx.dims = (BV, views, h, S, Mp // S, c, two)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        BV, views, h, S, Sp, c, two = ctx.dims
        dout = dout.float().contiguous()
        dO_r = torch.empty(BV, h, S * Sp, HEAD_DIM, device=dout.device, dtype=torch.float32)
        if two:
            O_r, L_r, O_c, L_c = ctx.saved_tensors
            dO_c, dL_r, dL_c = torch.empty_like(dO_r), torch.empty_like(L_r), torch.empty_like(L_c)
        else:
            O_r = L_r = O_c = L_c = dO_c = dL_r = dL_c = None
        nbytes = float(dout.numel() * 4 * (5 if two else 2))
        _lib.check(KERNEL_TIMER.run("bevr_merge_views_bwd", 0.0, L.bevr_merge_views_bwd, _ptr(dout), _ptr(O_r) if two else None,
                                    _ptr(L_r) if two else None, _ptr(O_c) if two else None, _ptr(L_c) if two else None,
                                    _ptr(dO_r), _ptr(dL_r) if two else None, _ptr(dO_c) if two else None,
                                    _ptr(dL_c) if two else None, BV, views, h, S, Sp, c, _stream(), nbytes=nbytes),
                   "bevr_merge_views_bwd")
        return dO_r, dL_r, dO_c, dL_c, None, None, None


class _MergeTap(torch.autograd.Function):
    """merge_tap below, through bevr_merge_tap_fwd / _bwd (csrc/merge.hip)."""

    @staticmethod
    def forward(ctx, O_r, L_r, Rn, L_c, Vp, bv, S, c, views):
        _require_gpu(O_r, L_r, Rn, L_c, Vp, bv)
        L = _lib.lib()
        BV, h, Mp, hd = O_r.shape
        assert hd == HEAD_DIM and Mp % S == 0 and BV % views == 0 and Rn.shape == (BV, h, Mp, TAP_N)
        assert Vp.shape == (BV, h, TAP_N, HEAD_DIM) and bv.shape == (h, HEAD_DIM)
        O_r, L_r, Rn, L_c, Vp, bv = (t.float().contiguous() for t in (O_r, L_r, Rn, L_c, Vp, bv))
        out = torch.empty(BV // views, S * S, views * h * c, device=O_r.device, dtype=torch.float32)
        live = S / (Mp // S)
        nbytes = float(O_r.numel() * live * (c / HEAD_DIM + TAP_N / HEAD_DIM) * 4 + out.numel() * 4)
        _lib.check(KERNEL_TIMER.run("bevr_merge_views_fwd", 0.0, L.bevr_merge_tap_fwd, _ptr(O_r), _ptr(L_r), _ptr(Rn), _ptr(L_c),
                                    _ptr(Vp), _ptr(bv), _ptr(out), BV, views, h, S, Mp // S, c, _stream(), nbytes=nbytes),
                   "bevr_merge_tap_fwd")
        ctx.save_for_backward(O_r, L_r, Rn, L_c, Vp, bv)
        ctx.dims = (BV, views, h, S, Mp // S, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        BV, views, h, S, Sp, c = ctx.dims
        O_r, L_r, Rn, L_c, Vp, bv = ctx.saved_tensors
        dout = dout.float().contiguous()
        dO_r, dL_r, dRn, dL_c = torch.empty_like(O_r), torch.empty_like(L_r), torch.empty_like(Rn), torch.empty_like(L_c)
        dVp, dbv = torch.zeros_like(Vp), torch.zeros_like(bv)
        nbytes = float(dout.numel() * 4 * (3 + 2 * TAP_N / HEAD_DIM))
        _lib.check(KERNEL_TIMER.run("bevr_merge_views_bwd", 0.0, L.bevr_merge_tap_bwd, _ptr(dout), _ptr(O_r), _ptr(L_r), _ptr(Rn),
                                    _ptr(L_c), _ptr(Vp), _ptr(bv), _ptr(dO_r), _ptr(dL_r), _ptr(dRn), _ptr(dL_c), _ptr(dVp),
                                    _ptr(dbv), BV, views, h, S, Sp, c, _stream(), nbytes=nbytes), "bevr_merge_tap_bwd")
        return dO_r, dL_r, dRn, dL_c, dVp, dbv, None, None, None


def merge_tap(O_r, L_r, Rn, L_c, Vp, bv, S: int, c: int, views: int) -> torch.Tensor:
    """merge_views with the tap segment's half formed on the way: O_c = Rn Vp + bv is never materialised (Rn (B', h, Mp, 12)
    the tap kernels' normalised weight sums, Vp (B', h, 12, 32) the 12 pixels' value rows, bv (h, 32) the value bias: the
    thin product, the add and, in the backward, two thin batched GEMMs and a row sum join the one pass each way)."""
    return _MergeTap.apply(O_r, L_r, Rn, L_c, Vp, bv, S, c, views)


def merge_views(O_r: torch.Tensor, S: int, c: int, views: int, L_r=None, O_c=None, L_c=None) -> torch.Tensor:
    """The attention kernels' packed output (B * views, h, Mp, 32) -> (B, S*S, views * h * c): unpack_out_views (views = 1:
    unpack_out) in one kernel; with a second segment (O_c, L_c and the first's L_r: the halves of one softmax over two key
    segments, each normalised by its own log2-sum-exp) also their merge
        O = 2^(L_r - L) O_r + 2^(L_c - L) O_c,   L = log2(2^L_r + 2^L_c)
    in the same pass -- forward and backward (the gradients of all four inputs)."""
    if c % 4 != 0:      # head widths the 16-byte kernel does not take: the same arithmetic as stock device ops
        if O_c is not None:
            L_t = torch.logaddexp2(L_r, L_c)
            O_r = torch.exp2(L_r - L_t)[..., None] * O_r + torch.exp2(L_c - L_t)[..., None] * O_c
        return unpack_out_views(O_r, S, c, views)
    return _MergeViews.apply(O_r, L_r, O_c, L_c, S, c, views)


def key_coords(pos: torch.Tensor, S: int, Wt: int, Np: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pos (P, N, 2) in (y, x), [-1, 1] units -> table coordinates a (rows), b (cols), padded to Np.
    ty = i + a, tx = j*rx + b reproduces grid_sample(align_corners=True) of (q_grid - pos)/2
    (model/SCA_deform_attn.py:365-389): ((q - p)/2 + 1)/2 * (size - 1)."""
    a = (1.0 - pos[..., 0]) * ((S - 1) / 2.0)
    b = (1.0 - pos[..., 1]) * ((Wt - 1) / 4.0)
    N = pos.shape[1]
    return F.pad(a, (0, Np - N)).contiguous(), F.pad(b, (0, Np - N)).contiguous()


def pack_table(rpe_table: torch.Tensor, g: AttnGeom) -> torch.Tensor:
    """(h, Ht, Wt) -> transposed, zero padded, times log2(e): (h, Wp, Hp + 1)."""
    t = rpe_table * LOG2E
    t = F.pad(t, (g.x_off, g.Wp - g.Wt - g.x_off, g.y_off, g.Hp + 1 - g.Ht - g.y_off))
    return t.transpose(1, 2).contiguous()


# --------------------------------------------------------------------------------------------------
# key ordering (host, init time)
# --------------------------------------------------------------------------------------------------
def kd_key_order(pos_ref, S: int, Wt: int, leaf: int = 32):
    """Order keys so that every run of `leaf` consecutive keys is spatially compact in rpe-table space.

    pos_ref (N, 2) STATIC key positions (y, x) in [-1, 1] units (camera projections of the pillar grid for
    SCA, the regular grid for TSA; learned offsets move a key by a few table cells only).  Returns a
    permutation (numpy int64, length N) from a k-d tree: split the longer table-space extent at a multiple
    of `leaf` until a node holds <= leaf keys.  Softmax attention is invariant to the order of its keys,
    so this changes no result; it bounds the table window a 32-key half of a step needs (csrc/attn_tile.h: the
    kernels window the two halves of a 64-key step separately).
    """
    import numpy as np
    leaf = int(os.environ.get("BEVR_KD_LEAF", leaf))
    pr = np.asarray(pos_ref, dtype=np.float64)
    a = (1.0 - pr[:, 0]) * ((S - 1) / 2.0)
    b = (1.0 - pr[:, 1]) * ((Wt - 1) / 4.0)
    out = []

    def rec(ids):
        n = len(ids)
        if n <= leaf:
            out.append(ids)
            return
        ka, kb = a[ids], b[ids]
        key = ka if (ka.max() - ka.min()) >= (kb.max() - kb.min()) else kb
        order = np.argsort(key, kind="stable")
        n_leaf = -(-n // leaf)
        left = (n_leaf // 2) * leaf
        rec(ids[order[:left]])
        rec(ids[order[left:]])

    rec(np.arange(len(pr)))
    return np.concatenate(out)


def split_key_order(ref_yx, S: int, Wt: int, min_cell_keys: int = 1024):
    """Static key order of an SCA call with the projector's pinned keys split off: (order (V, N) int64 tensor, split).

    ref_yx (V, N, 2) STATIC reference positions (y, x) of each view's keys.  Keys whose reference is exactly (-1, -1)
    are the pillar points the camera does not see (the projector pins them to pixel (0, 0), reference
    model/bev_cmr_proj.py:76): they differ only by their learned offsets and crowd ~70 cells of the rpe table.  Every
    view sends the same number of them -- the smallest pinned count over the views, rounded down to a multiple of 64
    -- to the END of its order, keys [split, N): the cell segment (sorted by table cell per call, cell_order, and
    attended through the cell kernels).  Keys [0, split): everything else, in the k-d order of kd_key_order (the region
    kernels).  Fewer than min_cell_keys pinned keys per view, or BEVR_CELL=0: split = N (no cell segment).
    Softmax attention is invariant to the order of its keys: neither the order nor the split changes a result."""
    import numpy as np
    yx = np.asarray(ref_yx, dtype=np.float64)
    V, N, _ = yx.shape
    pinned = (yx == -1.0).all(-1)
    n_b = (int(pinned.sum(1).min()) // 64) * 64
    if n_b < min_cell_keys or os.environ.get("BEVR_CELL", "1") == "0":
        n_b = 0
    orders = []
    for v in range(V):
        ip = np.nonzero(pinned[v])[0]
        seg_b = ip[len(ip) - n_b:] if n_b else ip[:0]
        keep = np.ones(N, dtype=bool)
        keep[seg_b] = False
        seg_a = np.nonzero(keep)[0]
        seg_a = seg_a[kd_key_order(yx[v][seg_a], S, Wt)]
        orders.append(torch.from_numpy(np.concatenate((seg_a, seg_b))))
    return torch.stack(orders, 0), N - n_b


# --------------------------------------------------------------------------------------------------
# kernel launchers
# --------------------------------------------------------------------------------------------------
def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.BevrError("bevrender_amd ops need ROCm device tensors; there is no CPU fallback")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _KernelTimer:
    """Optional per-kernel device timing with events recorded on the stream the kernels are launched on
    (bench.py's roofline leg).  Off by default: no events, no overhead."""

    def __init__(self):
        self.on = False
        self.records = []
        self.by_tag = {}

    def start(self):
        self.on, self.records = True, []

    def stop(self):
        self.on = False
        torch.cuda.synchronize()
        out = {}
        self.by_tag = {}
        for name, flops, nbytes, e0, e1, tag in self.records:
            ms = e0.elapsed_time(e1)
            for key, dst in ((name, out), (f"{name}[{tag}]", self.by_tag)):
                r = dst.setdefault(key, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "max_ms": 0.0})
                r["ms"] += ms
                r["max_ms"] = max(r["max_ms"], ms)
                r["n"] += 1
                r["flops"] += flops
                r["bytes"] += nbytes
        self.records = []
        return out

    def run(self, name, flops, fn, *args, nbytes=0.0, tag=""):
        """flops / nbytes: the launch's work (matrix flops of an attention launch, 2 * 32 per pair and product it runs;
        HBM bytes of a sampling launch), for the roofline lines of bench.py.  tag: which call (bench.py's per-call split)."""
        if not self.on:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.records.append((name, flops, nbytes, e0, e1, tag))
        return rc


KERNEL_TIMER = _KernelTimer()


def _edtype(precision: int):
    return {_lib.PREC_BF16: torch.bfloat16, _lib.PREC_F16: torch.float16}.get(precision, torch.float32)


def _perm_t(x: torch.Tensor) -> torch.Tensor:
    """(.., L, 32) row layout -> (.., 32, L) with the in-32 permutation over L (Vt / Kt / Qt / dOt)."""
    L = x.shape[-2]
    return x.index_select(-2, perm_index(L, x.device)).transpose(-1, -2).contiguous()


def _split_rows(x: torch.Tensor) -> torch.Tensor:
    """(.., 32) float rows -> the split-bf16 row format of BEVR_PREC_BF16X3 (csrc/bevr_common.h), same shape and dtype
    as a raw container: per 16-element half  hi(e0..7) | hi(e8..15) | lo(e0..7) | lo(e8..15),  hi = bf16(x), lo = bf16(x - hi)."""
    sh = x.shape[:-1]
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    out = torch.stack((hi.reshape(*sh, 2, 16), lo.reshape(*sh, 2, 16)), dim=-2)        # (.., half, plane, 16)
    return out.reshape(*sh, 64).contiguous().view(torch.float32)


def _unsplit_rows(xs: torch.Tensor) -> torch.Tensor:
    """hi + lo of a split row container (.., 32) -> float (.., 32)."""
    sh = xs.shape[:-1]
    b = xs.contiguous().view(torch.bfloat16).reshape(*sh, 2, 2, 16).float()
    return (b[..., 0, :] + b[..., 1, :]).reshape(*sh, 32)


def _split_perm_t(x: torch.Tensor) -> torch.Tensor:
    """(.., L, 32) float rows -> (.., 32, L) transposed split format: per 32-block, with y the block in perm32 order,
    16-byte chunk 2 h + s = hi(y[16 s + 8 h .. + 7]), chunk 4 + 2 h + s = lo(same)."""
    y = _perm_t(x)
    sh, L = y.shape[:-1], y.shape[-1]
    hi = y.to(torch.bfloat16)
    lo = (y - hi.float()).to(torch.bfloat16)
    out = torch.stack((hi.reshape(*sh, L // 32, 2, 2, 8), lo.reshape(*sh, L // 32, 2, 2, 8)), dim=-4)  # (.., blk, plane, s, h, 8)
    return out.transpose(-3, -2).reshape(*sh, 2 * L).contiguous().view(torch.float32)


def _attn_flops(geom, n_matmul, n_keys=None):
    """algorithmic MFMA flops of one attention launch: 2 flop/MAC x head_dim 32 x query-key pairs."""
    n = geom.N if n_keys is None else n_keys
    return 2.0 * HEAD_DIM * geom.n_prob * geom.heads * (geom.S * geom.S) * n * n_matmul


def _call_tag(geom) -> str:
    """which attention call a launch belongs to: TSA's table is square (2S - 1 wide), SCA's 2 S D - 1 wide."""
    return ("tsa" if geom.Wt == 2 * geom.S - 1 else "sca") + f",N={geom.N}"


def cell_order(a: torch.Tensor, b: torch.Tensor, n_tail: int = 0) -> torch.Tensor:
    """The order that makes a key segment CELL-SORTED (csrc/attn_cell.h): per problem, keys sorted by rpe-table cell
    (floor(a), floor(b)), rows ascending, columns boustrophedon (consecutive cells are neighbours).  a, b (P, N) table
    coordinates -> (P, N) long.  Softmax attention is invariant to the order of its keys: the order only decides how
    many 32-key tiles of the segment fit one table chunk.

    n_tail > 0: the n_tail keys that sit in the LEAST POPULATED cells come first (in cell order among themselves), the
    cell-sorted rest after them.  The caller hands those first n_tail keys to the region kernels: in the sparse tail of
    the offset distribution a 32-key tile spans more than one 4 x 4 chunk and takes the cell kernels' slow pass (~200x
    a fast tile); at the benchmark rig moving 128 keys per view removes 4 of 5 slow tiles
    (tools/analysis/slow_tiles.py).  No host sync: populations are run lengths of the sorted cell ids."""
    A = torch.floor(a).to(torch.int64)
    Bc = torch.floor(b).to(torch.int64)
    A = A - A.amin(1, keepdim=True)
    Bc = Bc - Bc.amin(1, keepdim=True)
    nB = Bc.amax(1, keepdim=True) + 1
    snake = torch.where(A % 2 == 0, Bc, nB - 1 - Bc)
    # the sort keys as narrow integers: a radix sort's passes go with the key width (cell ids of a table are < 2^31,
    # populations < 2^31, the tail flag is one byte) -- three sorts per SCA call on the product path
    cid = (A * nB + snake).to(torch.int32)
    order = cid.argsort(1)
    if n_tail <= 0:
        return order
    P, N = cid.shape
    s = cid.gather(1, order)
    idx = torch.arange(N, device=cid.device, dtype=torch.int32).expand(P, N)
    # population of each key's cell, in cell order: the length of its run of equal ids (two binary searches per key)
    pop = torch.searchsorted(s, s, right=True, out_int32=True) - torch.searchsorted(s, s, right=False, out_int32=True)
    # positions (in cell order) of the sparsest cells' keys: the n_tail smallest of (population, position) -- what a stable
    # sort by population would put first, selected instead of sorted
    if N <= 65536:
        tail = (pop.clamp(max=32767) * 65536 + idx).topk(n_tail, dim=1, largest=False, sorted=False).indices
    else:
        tail = pop.argsort(dim=1, stable=True)[:, :n_tail]
    flag = torch.ones(P, N, dtype=torch.uint8, device=cid.device).scatter_(1, tail, 0)
    return order.gather(1, flag.argsort(dim=1, stable=True))       # tail first, the rest after it, both in cell order


def _u32(x):
    return x & 0xFFFFFFFF


def dropout_keep_mask(seed: int, thr16: int, n_ph: int, S: int, N: int, device="cpu") -> torch.Tensor:
    """The attention-dropout keep mask the kernels evaluate (csrc/bevr_common.h:bevr_drop_keep), on the host:
    (n_ph, S*S, N) bool, query index m = i*S + j (the reference's flattening), key index n in the kernels' key order.
    keep iff the hash's top 16 bits >= thr16 = round(p * 65536).  Tests build the oracle's mask with it."""
    Sp = 32 * ((S + 31) // 32)
    i = torch.arange(S, dtype=torch.int64, device=device)
    mq = (i[None, :] * Sp + i[:, None]).reshape(-1)                    # m = i*S + j -> packed j*Sp + i
    ph = torch.arange(n_ph, dtype=torch.int64, device=device)
    n = torch.arange(N, dtype=torch.int64, device=device)
    row = _u32(seed ^ _u32(ph * 0x9E3779B1)[:, None] ^ _u32(mq * 0x85EBCA77)[None, :])       # (n_ph, M)
    x = _u32(row[:, :, None] ^ _u32(n * 0xC2B2AE3D)[None, None, :])
    x = x ^ (x >> 16)
    x = _u32(x * 0x7FEB352D)
    x = x ^ (x >> 15)
    x = _u32(x * 0x846CA68B)
    x = x ^ (x >> 16)
    return (x >> 16) >= thr16


@dataclass
class _Seg:
    """One key segment of an attention call: keys [n0, n0 + geom.N) of the caller's arrays."""
    cell: bool          # cell kernels (attn_cell_*.hip) or region kernels (attn_fwd.hip ...)
    n0: int
    geom: AttnGeom


def gather_supported(precision, S) -> bool:
    """Does the forward over scattered keys run on the gather kernel (csrc/attn_gather_fwd.hip: bias on the matrix cores,
    table taps gathered from an LDS window) instead of the region kernel?  bf16 operands, S <= 224 (a window column holds a
    key's taps for every BEV row of a column); BEVR_GATHER=0 turns it off (A/B timing)."""
    return precision == _lib.PREC_BF16 and S <= 224 and os.environ.get("BEVR_GATHER", "1") != "0"


def slab_supported(precision, S, Wt=None) -> bool:
    """Does the query-side backward over scattered keys run on the slab-stationary kernel (csrc/attn_slab_bwd_q.hip: a
    workgroup owns a slab of rpe-table columns, the keys arrive sorted by their table column) instead of the query-tile
    kernel bevr_attn_bwd_q?  16-bit operand modes, S <= 211 (7 row blocks of 31 queries and all their table rows in one
    LDS column).  With Wt given: only tables at least twice as wide as TSA's (rx >= 2 table columns per BEV column: SCA) --
    on TSA's regular key grid the query-tile kernel's window hardly moves and it is the faster one (0.91 against 0.77 T
    pairs/s on the benchmark, profiles/r05_*); BEVR_SLAB=0 turns the kernel off, BEVR_SLAB=2 forces it for every table
    (A/B timing, tests).  Same results up to the order of the float atomics and one rounding of the tap weights."""
    mode = os.environ.get("BEVR_SLAB", "1")
    if mode == "0" or precision not in (_lib.PREC_BF16, _lib.PREC_F16) or S > 211:
        return False
    return mode == "2" or Wt is None or Wt - 1 >= 4 * (S - 1)


class _AttnCore(torch.autograd.Function):
    """O = softmax(Q K^T + bias(a, b, table)) V in packed layouts (all inputs float32).

    The keys may be split into two segments that share one softmax: keys [0, split) go through the region kernels
    (scattered keys: LDS table windows, per-pair bias gather), keys [split, N) through the cell kernels (keys the
    caller sorted by table cell: bias as an MFMA).  The forward chains the segments through (O, LSE) in place; the
    backward passes of both use the final LSE and delta and accumulate into the same dQ and d(table)."""

    @staticmethod
    def forward(ctx, Qp, kv, key_a, key_b, Tt, geom: AttnGeom, split: int, feat=None, spos=None, Wkv=None, bkv=None,
                drop=None):
        """kv (B', N, 2C) projected rows -- or None with the K | V SOURCE instead: feat (B', Hi, Wi, C) channels-last
        feature map (float or bf16), spos (B', N, 2) sampling positions, Wkv (2C, C), bkv (2C,) the proj_k | proj_v
        weights: the operands are then produced by bevr_kv_project (csrc/kvproj.hip: sample -> project -> packed layouts
        in one pass, 16-bit operand modes)."""
        _require_gpu(Qp, kv, key_a, key_b, Tt, feat, spos, Wkv, bkv)
        L = _lib.lib()
        ed = _edtype(geom.precision)
        # drop = (thr16, seed): attention dropout, region kernels only (the keep mask is a function of (seed, ph, mq, n))
        ctx.drop = drop if drop and drop[0] > 0 else None
        x3 = geom.precision == _lib.PREC_BF16X3
        Qe = _split_rows(Qp.float()) if x3 else Qp.to(ed).contiguous()
        fused = kv is None
        if fused:
            feat, spos = feat.contiguous(), spos.float().contiguous()
            N, C2 = spos.shape[1], 2 * feat.shape[-1]
            W_e = Wkv.detach().to(ed).contiguous()
            vn2 = torch.zeros(1, device=Qp.device, dtype=torch.float32)   # largest squared V row norm (kv_project)
            b_f = None if bkv is None else bkv.detach().float().contiguous()
        else:
            kv = kv.float().contiguous()
            N, C2 = kv.shape[1], kv.shape[-1]
        c = C2 // 2 // geom.heads
        dev = Qp.device
        segs = []
        if split > 0:
            segs.append(_Seg(False, 0, dc_replace(geom, N=split)))
        if split < N:
            if ctx.drop:
                raise _lib.BevrError("attention dropout runs on the region kernels: pass no cell segment")
            segs.append(_Seg(True, split, dc_replace(geom, N=N - split)))
        Ttc = Tt.contiguous()
        pair = torch.stack((Ttc[..., :-1], Ttc[..., 1:]), dim=-1).contiguous()   # (h, Wp, Hp, 2)
        # zeros: the rows past the grid are never written, and a merge with another segment multiplies them by weight 0
        O = torch.zeros(geom.n_prob, geom.heads, geom.Mp, HEAD_DIM, device=dev, dtype=torch.float32)
        # plane 0: log2-sum-exp; plane 1: a bound of log2 of the row's largest softmax weight (rows past the grid: -inf here)
        LSE = torch.full((2, geom.n_prob, geom.heads, geom.Mp), float("-inf"), device=dev, dtype=torch.float32)
        need_bwd = any(ctx.needs_input_grad)
        saved = []
        pair_pk = None
        for i, sg in enumerate(segs):
            g = sg.geom
            # K | V rows (B', N, 2 h c) float -> the kernels' per-head layouts in one pass (csrc/pack.hip); the
            # transposed K is only read by the backward.  A segment is a row range of the caller's array (row stride C2,
            # problem stride N rows): no copy.
            Ke = torch.empty(g.n_prob, g.heads, g.Np, HEAD_DIM, device=dev, dtype=ed)
            Ve = torch.empty_like(Ke)
            Vt = torch.empty(g.n_prob, g.heads, HEAD_DIM, g.Np, device=dev, dtype=ed)
            Kt = torch.empty_like(Vt) if need_bwd else None
            kn2 = None
            if fused:
                nb, Hi, Wi, Cc = feat.shape
                # the largest squared K row norm per (problem, head), out of the projection kernel: the static softmax
                # reference of the gather forward needs it (a pass over K otherwise)
                if not sg.cell and not ctx.drop and gather_supported(g.precision, g.S) and os.environ.get("BEVR_KNORM", "1") != "0":
                    kn2 = torch.zeros(g.n_prob, g.heads, device=dev, dtype=torch.float32)
                _lib.check(KERNEL_TIMER.run(
                    "bevr_kv_project", 0.0, L.bevr_kv_project, _ptr(feat), int(feat.dtype == torch.bfloat16),
                    C.c_void_p(spos.data_ptr() + sg.n0 * 8), N, _ptr(W_e), _ptr(b_f), nb, Hi, Wi, Cc, g.N, g.Np, g.heads, c,
                    g.precision, _ptr(Ke), _ptr(Ve), _ptr(Kt), _ptr(Vt), _ptr(vn2) if need_bwd else None, _ptr(kn2),
                    g.groups, _stream(),
                    nbytes=float(feat.numel() * feat.element_size() + 8 * nb * g.N + (4 if need_bwd else 3) * Ke.numel() * 2)),
                    "bevr_kv_project")
            else:
                kp = kv.data_ptr() + sg.n0 * C2 * 4
                _lib.check(L.bevr_pack_kv(C.c_void_p(kp), C.c_void_p(kp + 2 * C2), C2, N, g.n_prob, g.N, g.Np, g.heads, c,
                                          g.precision, _ptr(Ke), _ptr(Ve), _ptr(Kt), _ptr(Vt), _stream()), "bevr_pack_kv")
            ka = F.pad(key_a[:, sg.n0:sg.n0 + g.N], (0, g.Np - g.N)).contiguous()
            kb = F.pad(key_b[:, sg.n0:sg.n0 + g.N], (0, g.Np - g.N)).contiguous()
            d = g.desc()
            # per-key table coordinates + per-tile tap boxes, shared by the forward and the backward passes
            key_ws = torch.empty(L.bevr_attn_key_ws_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
            _lib.check(L.bevr_attn_key_prep(C.byref(d), _ptr(ka), _ptr(kb), _ptr(key_ws), _stream()), "bevr_attn_key_prep")
            if sg.cell:
                o_in = _ptr(O) if i > 0 else None
                l_in = _ptr(LSE) if i > 0 else None
                _lib.check(KERNEL_TIMER.run("bevr_attn_cell_fwd", _attn_flops(g, 2), L.bevr_attn_cell_fwd, C.byref(d),
                                            _ptr(Qe), _ptr(Ke), _ptr(Vt), _ptr(key_ws), _ptr(pair), o_in, l_in, _ptr(O),
                                            _ptr(LSE), _stream(), tag=_call_tag(g)), "bevr_attn_cell_fwd")
            elif ctx.drop:
                _lib.check(L.bevr_attn_fwd_dropout(C.byref(d), _ptr(Qe), _ptr(Ke), _ptr(Vt), _ptr(key_ws), _ptr(pair), _ptr(O),
                                                   _ptr(LSE), ctx.drop[0], ctx.drop[1], _stream()), "bevr_attn_fwd_dropout")
            elif gather_supported(g.precision, g.S):
                if pair_pk is None:
                    pair_pk = pair.to(torch.bfloat16)       # (h, Wp, Hp, 2) 16-bit: one dword per (column, row) entry
                    # Tt is the table in log2 units already; two-stage maximum (a reduction to `heads` outputs in one stage
                    # runs on `heads` workgroups: 0.6 ms for the 27 MB table)
                    tmax = torch.maximum(Ttc.amax(-1).amax(-1), -Ttc.amin(-1).amin(-1))
                    qn = torch.linalg.vector_norm(Qe, dim=-1, dtype=torch.float32)          # (B, h, Mp)
                # static softmax reference: |Q_q . K_n| <= ||Q_q|| max_n ||K_n||, |bias| <= max |T2| (a convex combination;
                # 1 % for the 16-bit rounding of operands and weights), minus the headroom
                if kn2 is not None:      # of the unrounded rows: the rounding is inside the 1 % below
                    kmx = kn2.sqrt()
                else:
                    kmx = torch.linalg.vector_norm(Ke[:, :, :g.N], dim=-1, dtype=torch.float32).amax(-1)    # (B', h)
                ub = 1.01 * (qn.repeat_interleave(g.q_div, 0) * kmx[..., None] + tmax[None, :, None]) + 0.01
                mref = (ub - TAP_HEADROOM).contiguous()
                gflags = torch.zeros(g.n_prob * g.heads * g.S, device=dev, dtype=torch.int32)
                _lib.check(KERNEL_TIMER.run("bevr_attn_gather_fwd", _attn_flops(g, 2), L.bevr_attn_gather_fwd, C.byref(d),
                                            _ptr(Qe), _ptr(Ke), _ptr(Ve), _ptr(key_ws), _ptr(pair_pk), _ptr(mref), _ptr(O),
                                            _ptr(LSE), _ptr(gflags), _stream(), tag=_call_tag(g)), "bevr_attn_gather_fwd")
            else:
                _lib.check(KERNEL_TIMER.run("bevr_attn_fwd", _attn_flops(g, 2), L.bevr_attn_fwd, C.byref(d), _ptr(Qe),
                                            _ptr(Ke), _ptr(Vt), _ptr(key_ws), _ptr(pair), _ptr(O), _ptr(LSE),
                                            _stream(), tag=_call_tag(g)), "bevr_attn_fwd")
            saved += [Ke, Ve, Kt, ka, kb, key_ws]
        ctx.geom = geom
        # split mode: the packed V is not readable as numbers; the norm bound of the backward's scales from the rows
        ctx.vmax = kv[..., C2 // 2:].reshape(kv.shape[0], N, geom.heads, c).norm(dim=-1).max() if x3 and need_bwd else None
        ctx.segs = segs
        ctx.kv_shape = (geom.n_prob, N, C2)
        ctx.fused = fused
        if fused:
            ctx.n_seg_saved = len(saved)
            saved += [feat, spos, Wkv, vn2] + ([bkv] if bkv is not None else [])
            ctx.has_bias = bkv is not None
        ctx.save_for_backward(Qe, pair, O, LSE, *saved)
        ctx.set_materialize_grads(False)
        # second output: the rows' log2-sum-exp (plane 0).  Differentiable: a caller that merges this softmax with
        # another key segment (attention_core(tap_source=...)) sends a cotangent back, which enters delta below
        return O, LSE[0].clone()

    @staticmethod
    def backward(ctx, dO, dLSE=None):
        geom: AttnGeom = ctx.geom
        Qe, pair, O, LSE, *saved = ctx.saved_tensors
        if ctx.fused:
            src = saved[ctx.n_seg_saved:]
            saved = saved[:ctx.n_seg_saved]
        L = _lib.lib()
        ed = _edtype(geom.precision)
        if dO is None:
            dO = torch.zeros_like(O)
        dO = dO.contiguous()
        f16 = geom.precision == _lib.PREC_F16
        if f16:
            # fp16 has 5 exponent bits and the reference trains without a loss scaler: a cotangent of a mean-type loss
            # (~1e-7 per element) would fall into fp16's subnormals.  Every gradient is linear in dO, so the cotangent is
            # brought to ~2^10 with a power of two (exact) here and the gradients are scaled back below.  No host sync.
            sdo = torch.exp2(torch.floor(10.0 - torch.log2(dO.abs().max().clamp_min(1e-30))))
        x3 = geom.precision == _lib.PREC_BF16X3
        dev = dO.device
        prep = geom.precision in (_lib.PREC_BF16, _lib.PREC_F16) and dO.dtype == torch.float32 and O.dtype == torch.float32
        # delta = rowsum(dO o O) from the SAME (rounded) dO the kernels contract with V for dP: dS = P (dP - delta) then
        # cancels as it must where P -> 1 (with the f32 dO here and the bf16 one there, |dS| kept a floor of
        # 2^-9 |dO||V| -- pure noise in dQ, dK, d(pos), d(table) of a row dominated by one key).
        # With outputs (O, LSE2): d logit = ln2 P (dO . V - dO . O) + P dLSE2 = ln2 P (dP - (delta - dLSE2 / ln2))
        if prep:
            # 16-bit modes: the rounded cotangent (rows and transposed), delta and the two maxima behind `bound` in ONE pass
            # over dO and O (csrc/attn_bwd_prep.hip) -- seven stock passes before
            dOe = torch.empty(O.shape, device=dev, dtype=ed)
            dOt_p = torch.empty(geom.n_prob, geom.heads, HEAD_DIM, geom.Mp, device=dev, dtype=ed)
            delta = torch.empty(geom.n_prob, geom.heads, geom.Mp, device=dev, dtype=torch.float32)
            pstats = torch.zeros(2, device=dev, dtype=torch.float32)
            dl_in = dLSE.float().contiguous() if dLSE is not None else None
            lse0 = LSE[0].contiguous() if dLSE is not None else None
            _lib.check(L.bevr_attn_bwd_prep(_ptr(dO), _ptr(O), _ptr(sdo) if f16 else None, _ptr(dl_in), _ptr(lse0), _ptr(dOe),
                                            _ptr(dOt_p), _ptr(delta), _ptr(pstats), geom.n_prob * geom.heads, geom.Mp,
                                            geom.precision, _stream()), "bevr_attn_bwd_prep")
            dOr = None
        else:
            if f16:
                dO = dO * sdo
            dOe = _split_rows(dO.float()) if x3 else dO.to(ed).contiguous()
            dOr = dO.float() if x3 else dOe.float()
            delta = (dOr * O).sum(-1)
            if dLSE is not None:
                dl = torch.where(torch.isfinite(LSE[0]), dLSE.float(), torch.zeros_like(delta)) * LOG2E
                delta = delta - (dl * sdo if f16 else dl)
            delta = delta.contiguous()
        # zeros: bwd_q walks the grid in 31-row tiles and never visits the padded rows S..Sp-1 of a column; the cell
        # kernels add their segment's share
        dQ = torch.zeros(geom.n_prob, geom.heads, geom.Mp, HEAD_DIM, device=dev, dtype=torch.float32)
        dT = torch.zeros(geom.heads, geom.Wp, geom.Hp + 1, device=dev, dtype=torch.float32)
        Qt = _split_perm_t(_unsplit_rows(Qe)) if x3 else _perm_t(Qe)
        dOt = dOt_p if prep else (_split_perm_t(dOr) if x3 else _perm_t(dOe))
        N, C2 = ctx.kv_shape[1], ctx.kv_shape[-1]
        dkv = torch.empty(ctx.kv_shape, device=dev, dtype=torch.float32)
        # Scales of the backward kernels (include/bevrender_hip.h, grad_scale[8]); powers of two, on the device, no sync.
        # bound >= |dP - delta| of every pair: |dP| = |dO_q . V_n| <= ||dO_q|| ||V_n||; Pmax = the largest softmax weight of
        # the launch (forward, LSE plane 1; +0.05 in log2 for the rounding of the recomputed logits) -- with 10^5 keys
        # per row Pmax is far below 1.
        #   [0], [1]  s, 1/s with s Pmax bound <= 2^30: unit of bwd_q's fixed-point table-gradient cells (it applies s to
        #             dO and delta as it loads them -- exact -- and ln2 / s when it stores);
        #   fp16 only: [2] kp with Pmax 2^kp <= 2^14 (softmax weights as fp16 operands), [3] c2 with
        #             Pmax bound 2^kp c2 <= 2^14 (logit gradients as fp16 operands), [4], [5] the inverses; s = 2^16 2^kp c2.
        if ctx.fused:      # from the packing kernel (squared, of the unrounded rows: + 1 % for the rounding to E)
            vmax = src[3][0].sqrt() * 1.01
        elif x3:
            vmax = ctx.vmax
        else:
            vmax = torch.stack([saved[6 * i + 1].float().norm(dim=-1).max() for i in range(len(ctx.segs))]).max()
        if prep:
            bound = (pstats[0].sqrt() * vmax + pstats[1]).clamp_min(1e-30)
        else:
            bound = (dOr.norm(dim=-1).max() * vmax + delta.abs().max()).clamp_min(1e-30)
        pmax_log2 = (LSE[1].max() + 0.05).clamp(-60.0, 0.0)
        zero, one = torch.zeros((), device=dev), torch.ones((), device=dev)
        if f16:
            kp = torch.floor(14.0 - pmax_log2)
            e16 = torch.floor(14.0 - torch.log2(bound) - pmax_log2).clamp(-100.0, 100.0)
            gscale = torch.stack((torch.exp2(e16 + 16.0), torch.exp2(-e16 - 16.0), kp, torch.exp2(e16 - kp),
                                  torch.exp2(-e16), torch.exp2(-kp), zero, zero)).float().contiguous()
        else:
            e = torch.floor(30.0 - torch.log2(bound) - pmax_log2).clamp(-100.0, 100.0)
            gscale = torch.stack((torch.exp2(e), torch.exp2(-e), zero, one, one, one, zero, zero)).float().contiguous()
        das, dbs = [], []
        for i, sg in enumerate(ctx.segs):
            g = sg.geom
            Ke, Ve, Kt, ka, kb, key_ws = saved[6 * i:6 * i + 6]
            d = g.desc()
            if sg.cell:
                _lib.check(KERNEL_TIMER.run("bevr_attn_cell_bwd_q", _attn_flops(g, 3), L.bevr_attn_cell_bwd_q, C.byref(d),
                                            _ptr(Qe), _ptr(Ke), _ptr(Kt), _ptr(Ve), _ptr(key_ws), _ptr(pair), _ptr(dOe),
                                            _ptr(LSE), _ptr(delta), _ptr(gscale), _ptr(dQ), _ptr(dT), _stream(),
                                            tag=_call_tag(g)), "bevr_attn_cell_bwd_q")
            elif ctx.drop:
                _lib.check(L.bevr_attn_bwd_q_dropout(C.byref(d), _ptr(Qe), _ptr(Ke), _ptr(Kt), _ptr(Ve), _ptr(key_ws), _ptr(pair),
                                                     _ptr(dOe), _ptr(LSE), _ptr(delta), _ptr(gscale), _ptr(dQ), _ptr(dT),
                                                     ctx.drop[0], ctx.drop[1], _stream()), "bevr_attn_bwd_q_dropout")
            elif slab_supported(g.precision, g.S, g.Wt):
                # keys sorted by table column b per problem-group; K and V rows gathered into that order (softmax and its
                # gradients do not depend on the order of the keys; dK / dV come from the key-side kernel in the caller's)
                order = kb[:, :g.N].argsort(1)
                hpg = g.heads // g.groups
                idx = order.view(g.n_prob, g.groups, 1, g.N).expand(-1, -1, hpg, -1).reshape(g.n_prob, g.heads, g.N, 1)
                idx = idx.expand(-1, -1, -1, HEAD_DIM)
                Ks, Vs = Ke[:, :, :g.N].gather(2, idx), Ve[:, :, :g.N].gather(2, idx)
                sws = torch.empty(L.bevr_attn_slab_ws_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
                _lib.check(L.bevr_attn_slab_prep(C.byref(d), _ptr(ka), _ptr(kb), _ptr(order.to(torch.int32).contiguous()),
                                                 _ptr(sws), _stream()), "bevr_attn_slab_prep")
                _lib.check(KERNEL_TIMER.run("bevr_attn_slab_bwd_q", _attn_flops(g, 3), L.bevr_attn_slab_bwd_q, C.byref(d),
                                            _ptr(Qe), _ptr(Ks), _ptr(Vs), _ptr(sws), _ptr(pair), _ptr(dOe), _ptr(LSE),
                                            _ptr(delta), _ptr(gscale), _ptr(dQ), _ptr(dT), _stream(),
                                            tag=_call_tag(g)), "bevr_attn_slab_bwd_q")
                del Ks, Vs, sws
            else:
                _lib.check(KERNEL_TIMER.run("bevr_attn_bwd_q", _attn_flops(g, 3), L.bevr_attn_bwd_q, C.byref(d), _ptr(Qe),
                                            _ptr(Ke), _ptr(Kt), _ptr(Ve), _ptr(key_ws), _ptr(pair), _ptr(dOe),
                                            _ptr(LSE), _ptr(delta), _ptr(gscale), _ptr(dQ), _ptr(dT), _stream(),
                                            tag=_call_tag(g)), "bevr_attn_bwd_q")
            del Kt
            dK = torch.empty(g.n_prob, g.heads, g.Np, HEAD_DIM, device=dev, dtype=torch.float32)
            dV = torch.empty_like(dK)
            da = torch.zeros_like(ka)
            db = torch.zeros_like(kb)
            if sg.cell:
                _lib.check(KERNEL_TIMER.run("bevr_attn_cell_bwd_k", _attn_flops(g, 4), L.bevr_attn_cell_bwd_k, C.byref(d),
                                            _ptr(Qe), _ptr(Qt), _ptr(Ke), _ptr(Ve), _ptr(key_ws), _ptr(pair), _ptr(dOe),
                                            _ptr(dOt), _ptr(LSE), _ptr(delta), _ptr(gscale), _ptr(dK), _ptr(dV),
                                            _ptr(da), _ptr(db), _stream(), tag=_call_tag(g)), "bevr_attn_cell_bwd_k")
            elif ctx.drop:
                _lib.check(L.bevr_attn_bwd_k_dropout(C.byref(d), _ptr(Qe), _ptr(Qt), _ptr(Ke), _ptr(Ve), _ptr(ka), _ptr(kb),
                                                     _ptr(pair), _ptr(dOe), _ptr(dOt), _ptr(LSE), _ptr(delta), _ptr(gscale),
                                                     _ptr(dK), _ptr(dV), _ptr(da), _ptr(db), ctx.drop[0], ctx.drop[1],
                                                     _stream()), "bevr_attn_bwd_k_dropout")
            else:
                _lib.check(KERNEL_TIMER.run("bevr_attn_bwd_k", _attn_flops(g, 4), L.bevr_attn_bwd_k, C.byref(d), _ptr(Qe),
                                            _ptr(Qt), _ptr(Ke), _ptr(Ve), _ptr(ka), _ptr(kb), _ptr(pair), _ptr(dOe),
                                            _ptr(dOt), _ptr(LSE), _ptr(delta), _ptr(gscale), _ptr(dK), _ptr(dV),
                                            _ptr(da), _ptr(db), _stream(), tag=_call_tag(g)), "bevr_attn_bwd_k")
            # gradients of the row layout back to K | V rows (the adjoint of the packing), into the segment's rows
            kp = dkv.data_ptr() + sg.n0 * C2 * 4
            _lib.check(L.bevr_unpack_dkv(_ptr(dK), _ptr(dV), C.c_void_p(kp), C.c_void_p(kp + 2 * C2), C2, N, g.n_prob,
                                         g.N, g.Np, g.heads, C2 // 2 // g.heads, _stream()), "bevr_unpack_dkv")
            das.append(da[:, :g.N])
            dbs.append(db[:, :g.N])
        if geom.q_div > 1:  # the views of one sample share the query: sum their query gradients
            dQ = dQ.reshape(geom.n_prob // geom.q_div, geom.q_div, geom.heads, geom.Mp, HEAD_DIM).sum(1)
        da = das[0] if len(das) == 1 else torch.cat(das, 1)
        db = dbs[0] if len(dbs) == 1 else torch.cat(dbs, 1)
        if f16:   # undo the cotangent's power-of-two scale
            inv = 1.0 / sdo
            dQ, dkv, da, db, dT = dQ * inv, dkv * inv, da * inv, db * inv, dT * inv
        if not ctx.fused:
            return dQ, dkv, da, db, dT, None, None, None, None, None, None, None
        # adjoint of the fused K | V source in its unfused form: the projection's GEMMs on the float samples (recomputed:
        # the forward never wrote them) and the sampler's scatter
        feat, spos, Wkv = src[0], src[1], src[2]
        G = geom.groups
        if G > 1:
            # group gi's channels sampled at group gi's positions: the map as B' G images of C / G channels (a copy of the
            # feature map: small next to the rows), the rows back to (B', N, C)
            nb, Hi, Wi, Cc = feat.shape
            fg = feat.reshape(nb, Hi, Wi, G, Cc // G).permute(0, 3, 1, 2, 4).reshape(nb * G, Hi, Wi, Cc // G).contiguous()
            xs = _Sample.sample(fg, spos).reshape(nb, G, -1, Cc // G).permute(0, 2, 1, 3).reshape(nb, -1, Cc)
            d2 = dkv.reshape(-1, dkv.shape[-1])
            dW = torch.bmm(dkv.transpose(1, 2), xs).sum(0) if ctx.needs_input_grad[9] else None
            dbias = d2.sum(0) if ctx.has_bias and ctx.needs_input_grad[10] else None
            dxs = (d2 @ Wkv.float()).reshape(nb, -1, G, Cc // G).permute(0, 2, 1, 3).reshape(nb * G, -1, Cc // G).contiguous()
            del xs
            dfg, dspos = _Sample.scatter(fg, spos, dxs, ctx.needs_input_grad[7])
            dfeat = None if dfg is None else dfg.reshape(nb, G, Hi, Wi, Cc // G).permute(0, 2, 3, 1, 4).reshape(feat.shape)
            return dQ, None, da, db, dT, None, None, dfeat, dspos, dW, dbias, None
        xs = _Sample.sample(feat, spos)                                              # (B', N, C) float
        d2 = dkv.reshape(-1, dkv.shape[-1])
        # the weight gradient as one GEMM per problem, summed: rocBLAS runs the single (2C x B'N) @ (B'N x C) product with
        # its 1.7 M-long contraction at 2.2 ms per SCA call, the batched form at 0.57 (tools/prof_kv_adjoint.py)
        dW = torch.bmm(dkv.transpose(1, 2), xs).sum(0) if ctx.needs_input_grad[9] else None
        dbias = d2.sum(0) if ctx.has_bias and ctx.needs_input_grad[10] else None
        dxs = (d2 @ Wkv.float()).reshape(xs.shape)
        del xs
        dfeat, dspos = _Sample.scatter(feat, spos, dxs, ctx.needs_input_grad[7])
        return dQ, None, da, db, dT, None, None, dfeat, dspos, dW, dbias, None


def attention_core(query: torch.Tensor, kproj: Optional[torch.Tensor], vproj: Optional[torch.Tensor], pos: torch.Tensor,
                   rpe_table: torch.Tensor, *, heads: int, groups: int, views: int, precision: int,
                   kv: Optional[torch.Tensor] = None, cell_split: Optional[int] = None, kv_source=None,
                   tap_source=None, attn_drop=None, concat_views: bool = False) -> torch.Tensor:
    """Fused attention of the BEV query against sampled keys.

    query (B, C, S, S) layer-normed BEV query (used raw as Q); kproj, vproj (B*views, N, C) projected
    sampled features -- or `kv` (B*views, N, 2C) = K | V side by side, as one GEMM emits them (kproj = vproj = None);
    pos (B*views*groups, N, 2) key positions (y, x); rpe_table (h, 2S-1, Wt).
    cell_split: keys [cell_split, N) are CELL-SORTED (cell_order: the caller ordered them by rpe-table cell) and go
    through the cell kernels (bias as an MFMA), keys [0, cell_split) through the region kernels; None = N (no cell
    segment).  Any split gives the same result; it only decides the speed.
    kv_source = (feat, Wkv, bkv) instead of kproj / vproj / kv: feat (B*views, Hi, Wi, C) the channels-last feature map
    (float or bf16) the keys are sampled from AT `pos`, Wkv (2C, C) / bkv (2C,) the proj_k | proj_v weights: sampling,
    projection and operand packing run as one kernel (csrc/kvproj.hip; 16-bit operand modes; with channel groups pos is
    (B*views*groups, N, 2) and group gi's channels are sampled at its own positions -- see kv_source_supported).
    tap_source = True (with kv_source and cell_split): the keys [cell_split, N) all sample inside the top-left 4 x 3
    pixels of `feat` (the caller's contract: the projector-pinned keys, tap_supported) and go through the TAP kernels
    (csrc/attn_tap.h): their K and V are never formed -- the logits come from G = Q Kpix^T (12 pixels), the output from
    O = Rn Vpix, both thin GEMMs here -- and the segment is merged with the region kernels' through (O, LSE).
    attn_drop = (p, seed): dropout on the softmax weights (the reference's attn_drop, :402-409): a weight is kept with
    probability 1 - p (p rounded to 1/65536) and scaled by 1 / (1 - p); the mask is the function dropout_keep_mask of
    (seed, problem-head, query, key) that the forward and backward kernels share.  Every key then runs on the region
    kernels (cell_split / tap_source are ignored).
    Returns (B*views, S*S, C): per view softmax(QK^T c^-0.5 + bias) V, rows in i*S + j order -- or, with
    concat_views=True, (B, S*S, views*C): the views side by side in the channel axis (unpack_out_views: what SCA's
    proj_out contracts, written in one pass).
    Replaces model/SCA_deform_attn.py:304-413 / model/TSA_deform_attn.py:220-333.
    """
    B, Cc, S, _ = query.shape
    drop = None
    if attn_drop is not None and attn_drop[0] > 0.0:
        thr = int(round(float(attn_drop[0]) * 65536.0))
        if not 0 < thr < 65536:
            raise ValueError("attention dropout probability must lie in (0, 1)")
        drop = (thr, int(attn_drop[1]) & 0xFFFFFFFF)
        cell_split, tap_source = None, None
    if kv_source is not None:
        if kv is not None or kproj is not None or vproj is not None:
            raise ValueError("pass kv_source alone")
        feat, Wkv, bkv = kv_source
        if not kv_source_supported(Cc, heads, groups, precision):
            raise ValueError("kv_source needs a 16-bit operand mode, C % 16 == 0 and (C / groups) % 4 == 0")
        Bp, N = feat.shape[0], pos.shape[1]
        if pos.shape[0] != Bp * groups or feat.shape[-1] != Cc or tuple(Wkv.shape) != (2 * Cc, Cc):
            raise ValueError("kv_source shapes: feat (B*views, Hi, Wi, C), pos (B*views*groups, N, 2), Wkv (2C, C)")
    else:
        if kv is None:
            kv = torch.cat((kproj, vproj), -1)
        elif kproj is not None or vproj is not None:
            raise ValueError("pass either kproj and vproj, or kv")
        Bp, N, C2 = kv.shape
        if C2 != 2 * Cc:
            raise ValueError(f"K | V rows must have 2 x {Cc} channels, got {C2}")
    c = Cc // heads
    split = N if cell_split is None else int(cell_split)
    if not 0 <= split <= N:
        raise ValueError("cell_split must lie in [0, N]")
    Wt = rpe_table.shape[-1]
    if rpe_table.shape[-2] != 2 * S - 1:
        raise ValueError("rpe_table height must be 2S-1")
    tap = bool(tap_source) and split < N
    if tap and (kv_source is None or not tap_supported(precision, groups) or 16 * ((S + 15) // 16) > 448):
        raise ValueError("tap_source needs kv_source, groups == 1, a 16-bit operand mode and S <= 448")
    geom = AttnGeom(n_prob=Bp, q_div=views, heads=heads, groups=groups, S=S, N=split if tap else N, Wt=Wt, precision=precision)
    f32_layout = precision in (_lib.PREC_F32, _lib.PREC_BF16X3)
    if not tap and split < N and (geom.Sp > 480 or (f32_layout and geom.Sp > 224) or (N - split) > 8 * 100 * 1024):
        # the cell kernels run one wave per 32-row block of a BEV column + a producer wave, 16 at most; with f32-sized
        # operands the 512-thread instantiation (the one without spills) ends at 7 row blocks + the producer; and the slow
        # pass lists a segment's tiles in LDS (4 bytes per 32 keys next to the staging buffers): a segment beyond
        # ~800 000 keys would not fit.  Such calls keep every key on the region kernels (any split is a valid result)
        split = N
    Qp = pack_query(query.float(), heads)
    a, b = key_coords(pos.float(), S, Wt, N)
    Tt = pack_table(rpe_table.float(), geom)
    if not tap:
        if kv_source is not None:
            O, _ = _AttnCore.apply(Qp, None, a, b, Tt, geom, split, feat, pos.float(), Wkv, bkv, drop)
        else:
            O, _ = _AttnCore.apply(Qp, kv.float(), a, b, Tt, geom, split, None, None, None, None, drop)
        return _unpacked(O, S, c, views, concat_views)

    # ---- keys [0, split): region kernels; keys [split, N): tap kernels; one softmax, merged through (O, LSE) ----
    V = views
    Hi, Wi = feat.shape[1], feat.shape[2]
    O_r = LSE_r = None
    if split > 0:
        O_r, LSE_r = _AttnCore.apply(Qp, None, a[:, :split], b[:, :split], Tt, geom, split, feat,
                                     pos[:, :split].float().contiguous(), Wkv, bkv)
    # the 12 pixels' K | V rows, without the bias: (B', 12, 2C); rows the image does not have are zero (zero padding)
    fpix = feat[:, :TAP_R, :TAP_C, :].float()
    fpix = F.pad(fpix, (0, 0, 0, TAP_C - fpix.shape[2], 0, TAP_R - fpix.shape[1])).reshape(Bp, TAP_N, Cc)
    kvp = F.linear(fpix, Wkv.float())
    pad_c = HEAD_DIM - c
    # G[q][t] = Q_q . Kpix_t per (sample, view, head): one GEMM per (sample, head) against the views' 12 pixels side by side
    Kp = F.pad(kvp[..., :Cc].reshape(B, V, TAP_N, heads, c), (0, pad_c)).permute(0, 3, 4, 1, 2).reshape(B, heads, HEAD_DIM, V * TAP_N)
    G = torch.matmul(Qp, Kp).reshape(B, heads, geom.Mp, V, TAP_N).permute(0, 3, 1, 2, 4).reshape(Bp, heads, geom.Mp, TAP_N)
    Gb = torch.matmul(Qp[..., :c], bkv[:Cc].float().reshape(heads, c, 1)).squeeze(-1)                # (B, h, Mp)
    tgeom = dc_replace(geom, N=N - split)
    key_y = (pos[:, split:, 0].float() + 1.0) * (0.5 * (Hi - 1))
    key_x = (pos[:, split:, 1].float() + 1.0) * (0.5 * (Wi - 1))
    Rn, LSE_c = _TapAttn.apply(G, a[:, split:], b[:, split:], key_y, key_x, Tt, tgeom)
    LSE_c = (LSE_c.reshape(B, V, heads, geom.Mp) + Gb[:, None]).reshape(Bp, heads, geom.Mp)
    Vp = F.pad(kvp[..., Cc:].reshape(Bp, TAP_N, heads, c), (0, pad_c)).permute(0, 2, 1, 3)          # (B', h, 12, 32)
    bv = F.pad(bkv[Cc:].float().reshape(1, heads, 1, c), (0, pad_c))
    if O_r is not None and c % 4 == 0 and os.environ.get("BEVR_MERGE_TAP", "1") != "0":
        # the two halves of the softmax merged and unpacked in one pass, the tap half's O = Rn Vpix + bv formed on the way
        return merge_tap(O_r, LSE_r, Rn, LSE_c, Vp, bv.reshape(heads, HEAD_DIM), S, c, views if concat_views else 1)
    O_c = torch.matmul(Rn, Vp) + bv
    if O_r is None:
        return _unpacked(O_c, S, c, views, concat_views)
    # the two halves of the softmax merged and unpacked in one pass (csrc/merge.hip)
    return _unpacked(O_r, S, c, views, concat_views, LSE_r, O_c, LSE_c)


def _unpacked(O, S, c, views, concat_views, L_r=None, O_c=None, L_c=None):
    """attention_core's return value out of the packed layout: (B, S*S, views*C) with concat_views, else (B', S*S, C)."""
    return merge_views(O, S, c, views if concat_views else 1, L_r, O_c, L_c)


# --------------------------------------------------------------------------------------------------
# tap kernels (csrc/attn_tap*.hip): the projector-pinned keys without K and V
# --------------------------------------------------------------------------------------------------
TAP_R, TAP_C, TAP_N, TAP_SLOTS = 4, 3, 12, 16     # csrc/attn_tap.h: feature rows 0..3 x columns 0..2, 16 operand slots
TAP_HEADROOM = 64.0                               # binades between the static softmax reference and the logits' upper bound
TAP_HEADROOM_F16 = 8.0                            # fp16 operands: the weights 2^(S - mref) are rounded to fp16 (largest value
                                                  # 2^15.9; include/bevrender_hip.h: headroom <= 8).  Weights more than ~32
                                                  # binades under the bound flush to zero; a row that loses ALL of them is
                                                  # flagged and recomputed with an online maximum (the EXACT pass)


def tap_headroom(precision: int) -> float:
    return TAP_HEADROOM_F16 if precision == _lib.PREC_F16 else TAP_HEADROOM
LN2 = 0.6931471805599453


def _neg_big(ed):
    """logit of a masked key / offset of a padding row in the 16-bit operand dtype (finite in it)."""
    return -1.0e30 if ed == torch.bfloat16 else -30000.0


def _set_offset(G16: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """slots 12, 13 of the packed operand <- hi, lo 16-bit parts of the row offset c; returns hi + lo (float): what the
    kernels add to the row's logits."""
    hi = c.to(G16.dtype)
    lo = (c - hi.float()).to(G16.dtype)
    G16[..., 12] = hi
    G16[..., 13] = lo
    return hi.float() + lo.float()


class _TapAttn(torch.autograd.Function):
    """Softmax over a key segment whose keys sample inside the top-left 4 x 3 feature pixels, in terms of the TAP
    WEIGHTS instead of K and V (csrc/attn_tap.h):  logits S[n][q] = sum_t w_t(n) G[q][t] + bias[n][q]  ->
        Rn[q][t] = sum_n softmax_n(S)[n][q] w_t(n)      (P, h, Mp, 12)
        LSE[q]   = log2 sum_n 2^S[n][q]                 (P, h, Mp)
    Both outputs are differentiable (the caller turns Rn into O = Rn Vpix + bv and merges LSE with the other key segment
    of the same softmax in plain torch code).  G (P, h, Mp, 12) float: log2-domain logit per tap; key_a, key_b (P, N)
    table coordinates, key_y, key_x (P, N) sampling positions in feature pixels; Tt the packed table (pack_table)."""

    @staticmethod
    def forward(ctx, G, key_a, key_b, key_y, key_x, Tt, geom: AttnGeom):
        _require_gpu(G, key_a, key_b, key_y, key_x, Tt)
        L = _lib.lib()
        ed = _edtype(geom.precision)
        dev = G.device
        P, h, Mp = geom.n_prob, geom.heads, geom.Mp
        d = geom.desc()
        pad = geom.Np - geom.N
        ka, kb, ky, kx = (F.pad(t.float(), (0, pad)).contiguous() for t in (key_a, key_b, key_y, key_x))
        ws = torch.empty(L.bevr_attn_tap_ws_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
        _lib.check(L.bevr_attn_tap_prep(C.byref(d), _ptr(ka), _ptr(kb), _ptr(ky), _ptr(kx), _ptr(ws), _stream()),
                   "bevr_attn_tap_prep")
        Ttc = Tt.contiguous()
        pair = torch.stack((Ttc[..., :-1], Ttc[..., 1:]), dim=-1).contiguous()
        G16 = torch.zeros(P, h, Mp, TAP_SLOTS, device=dev, dtype=ed)
        G16[..., :TAP_N] = G
        G16[..., 14] = _neg_big(ed)
        # static softmax reference: an upper bound of the row's logits (the tap weights and the 4 bias taps are convex
        # weights up to their 16-bit rounding) minus the headroom -- no weight can overflow, nothing is tracked in the loop
        tmax = Ttc.amax(-1).amax(-1).clamp_min(0.0)       # two stages: see _AttnCore.forward
        ub = 1.01 * (G16[..., :TAP_N].float().amax(-1).clamp_min(0.0) + tmax[None, :, None]) + 0.01
        mref = (-_set_offset(G16, tap_headroom(geom.precision) - ub)).contiguous()
        # zeros: the kernels work in 16-row blocks and never touch the rows past the last block of a column
        R = torch.zeros(P, h, Mp, TAP_SLOTS, device=dev, dtype=torch.float32)
        flags = torch.zeros(P * h, geom.S, device=dev, dtype=torch.int32)
        _lib.check(KERNEL_TIMER.run("bevr_attn_tap_fwd", _attn_flops(geom, 2), L.bevr_attn_tap_fwd, C.byref(d), _ptr(G16),
                                    _ptr(ws), _ptr(pair), _ptr(mref), _ptr(R), _ptr(flags), _stream(), tag=_call_tag(geom)), "bevr_attn_tap_fwd")
        # rows past the grid: Rn = 0, LSE = 0 (finite: the caller's merge with the other key segment stays finite there)
        valid = (torch.arange(Mp, device=dev) % geom.Sp) < geom.S
        l = torch.where(valid, R[..., 15], torch.ones_like(mref))
        LSE = torch.where(valid, mref + torch.log2(l), torch.zeros_like(mref))
        Rn = torch.where(valid[:, None], R[..., :TAP_N] / l[..., None], torch.zeros_like(R[..., :TAP_N]))
        ctx.geom = geom
        ctx.save_for_backward(G16, Rn, LSE, ws, pair, Ttc)
        ctx.set_materialize_grads(False)
        return Rn, LSE

    @staticmethod
    def backward(ctx, dRn, dLSE):
        geom: AttnGeom = ctx.geom
        G16, Rn, LSE, ws, pair, Ttc = ctx.saved_tensors
        L = _lib.lib()
        ed = G16.dtype
        dev = G16.device
        P, h, Mp = geom.n_prob, geom.heads, geom.Mp
        d = geom.desc()
        valid = (torch.arange(Mp, device=dev) % geom.Sp) < geom.S
        H16 = torch.zeros(P, h, Mp, TAP_SLOTS, device=dev, dtype=ed)
        sdo = None
        if ed == torch.float16:
            # fp16 has 5 exponent bits and the reference trains without a loss scaler: the cotangents of a mean-type loss
            # (~1e-7 per element) would fall into fp16's subnormals as the H operand.  Every output is linear in
            # (dRn, dLSE): both are brought to ~2^8 with a power of two (exact) and the gradients scaled back.  No sync.
            big = torch.zeros((), device=dev)
            if dRn is not None:
                big = torch.maximum(big, dRn.abs().max())
            if dLSE is not None:
                big = torch.maximum(big, dLSE.abs().max())
            sdo = torch.exp2(torch.floor(8.0 - torch.log2(big.clamp_min(1e-30))))
            dRn = None if dRn is None else dRn * sdo
            dLSE = None if dLSE is None else dLSE * sdo
        if dRn is not None:
            H16[..., :TAP_N] = dRn * LN2
        # delta from the values the kernel contracts (the rounded H): dS = P (dP - delta) then cancels where P -> 1
        delta = (Rn * H16[..., :TAP_N].float()).sum(-1)
        if dLSE is not None:
            delta = delta - dLSE
        _set_offset(H16, -delta)
        Gq = G16.clone()
        _set_offset(Gq, torch.where(valid, -LSE, torch.full_like(LSE, _neg_big(ed))))
        dG = torch.zeros(P, h, Mp, TAP_SLOTS, device=dev, dtype=torch.float32)     # rows past the last 16-row block: never written
        dT = torch.zeros_like(Ttc)
        _lib.check(KERNEL_TIMER.run("bevr_attn_tap_bwd_q", _attn_flops(geom, 3), L.bevr_attn_tap_bwd_q, C.byref(d), _ptr(Gq),
                                    _ptr(H16), _ptr(ws), _ptr(pair), _ptr(dG), _ptr(dT), _stream(), tag=_call_tag(geom)), "bevr_attn_tap_bwd_q")
        dk = [torch.zeros(P, geom.Np, device=dev, dtype=torch.float32) for _ in range(4)]
        _lib.check(KERNEL_TIMER.run("bevr_attn_tap_bwd_k", _attn_flops(geom, 4), L.bevr_attn_tap_bwd_k, C.byref(d), _ptr(Gq),
                                    _ptr(H16), _ptr(ws), _ptr(Ttc), *[_ptr(t) for t in dk], _stream(), tag=_call_tag(geom)), "bevr_attn_tap_bwd_k")
        da, db, dy, dx = (t[:, :geom.N] for t in dk)
        dGo = dG[..., :TAP_N] * valid[:, None]
        if sdo is not None:
            inv = 1.0 / sdo
            dGo, da, db, dy, dx, dT = dGo * inv, da * inv, db * inv, dy * inv, dx * inv, dT * inv
        return dGo, da, db, dy, dx, dT, None


def tap_supported(precision: int, groups: int) -> bool:
    """The 16-bit operand modes (fp16 since round 5: headroom 8 and a power-of-two scale on the cotangents, see
    _TapAttn); BEVR_TAP=0 keeps the pinned keys on the cell kernels."""
    return precision in (_lib.PREC_BF16, _lib.PREC_F16) and groups == 1 and os.environ.get("BEVR_TAP", "1") != "0"


def kv_source_supported(C: int, heads: int, groups: int, precision: int) -> bool:
    """Can the K | V operands come from the fused sample -> project -> pack kernel (csrc/kvproj.hip)?  BEVR_FUSED_KV=0
    turns it off (the unfused chain sample -> rocBLAS -> bevr_pack_kv is then used; same results to the operands'
    rounding)."""
    return (groups >= 1 and C % groups == 0 and (C // groups) % 4 == 0 and precision in (_lib.PREC_BF16, _lib.PREC_F16)
            and C % 16 == 0 and C <= 256 and C % heads == 0 and C // heads <= 32
            and os.environ.get("BEVR_FUSED_KV", "1") != "0")


class _Sample(torch.autograd.Function):
    """grid_sample(bilinear, align_corners=True, zeros) on a channels-last map: (nb,Hi,Wi,C),(nb,N,2)->(nb,N,C).
    The map is float or bfloat16 (read as it is: the bf16 configurations keep the backbone features bf16 in HBM); the
    samples and every gradient are float."""

    @staticmethod
    def forward(ctx, feat, pos):
        _require_gpu(feat, pos)
        feat, pos = feat.contiguous(), pos.contiguous()
        out = _Sample.sample(feat, pos)
        ctx.save_for_backward(feat, pos)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, pos = ctx.saved_tensors
        return _Sample.scatter(feat, pos, dout, ctx.needs_input_grad[0])

    @staticmethod
    def sample(feat, pos):
        nb, Hi, Wi, Cc = feat.shape
        N = pos.shape[1]
        out = torch.empty(nb, N, Cc, device=feat.device, dtype=torch.float32)
        # algorithmic (compulsory) HBM bytes: the feature map once, a position and an output row per key.  The 4 taps
        # of a key are NOT 4 HBM reads: neighbouring keys share them through L2 (SURVEY 8d counts the map once too)
        bf = feat.dtype == torch.bfloat16
        fn = _lib.lib().bevr_sample_fwd_bf16 if bf else _lib.lib().bevr_sample_fwd
        _lib.check(KERNEL_TIMER.run("bevr_sample_fwd", 0.0, fn, _ptr(feat), _ptr(pos),
                                    _ptr(out), nb, Hi, Wi, Cc, N, _stream(),
                                    nbytes=4.0 * nb * (Hi * Wi * Cc * (0.5 if bf else 1.0) + N * Cc + 2 * N)),
                   "bevr_sample_fwd")
        return out

    @staticmethod
    def scatter(feat, pos, dout, need_dfeat=True):
        nb, Hi, Wi, Cc = feat.shape
        N = pos.shape[1]
        dout = dout.contiguous()
        bf = feat.dtype == torch.bfloat16
        dfeat = torch.zeros(feat.shape, device=feat.device, dtype=torch.float32)
        dpos = torch.empty_like(pos)
        # algorithmic (compulsory) HBM bytes: dout row, position and position gradient per key; the feature map read
        # once (position gradient) and its gradient written once.  What the scatter really costs is the atomic traffic
        # (4 taps x N x C floats leave L2 as memory-side atomics): bench.py reports that as `traffic` from the PMC counters
        fn = _lib.lib().bevr_sample_bwd_bf16 if bf else _lib.lib().bevr_sample_bwd
        _lib.check(KERNEL_TIMER.run("bevr_sample_bwd", 0.0, fn, _ptr(feat), _ptr(pos),
                                    _ptr(dout), _ptr(dfeat), _ptr(dpos), nb, Hi, Wi, Cc, N, _stream(),
                                    nbytes=4.0 * nb * (N * Cc + (1.5 if bf else 2.0) * Hi * Wi * Cc + 4 * N)),
                   "bevr_sample_bwd")
        # autograd wants the input's dtype; the cast is skipped when nobody reads the map's gradient
        return (dfeat.to(feat.dtype) if bf and need_dfeat else dfeat if not bf else None), dpos


def sample_features(feat_nchw: torch.Tensor, pos: torch.Tensor, groups: int) -> torch.Tensor:
    """feat (B, C, Hi, Wi), pos (B*g, N, 2) (y, x) -> sampled (B, N, C).  Group gi's channels are sampled at
    group gi's positions (model/SCA_deform_attn.py:290-301: x.reshape(B*g, C/g, Hi, Wi))."""
    B, Cc, Hi, Wi = feat_nchw.shape
    g = groups
    N = pos.shape[1]
    f = feat_nchw if feat_nchw.dtype == torch.bfloat16 else feat_nchw.float()
    f = f.reshape(B, g, Cc // g, Hi, Wi).permute(0, 1, 3, 4, 2).reshape(B * g, Hi, Wi, Cc // g)
    xs = _Sample.apply(f.contiguous(), pos.float())
    return xs.reshape(B, g, N, Cc // g).permute(0, 2, 1, 3).reshape(B, N, Cc)


def project_bev_grid(points_3d: torch.Tensor, cam_inv: torch.Tensor, Kmat: torch.Tensor, img_w: int,
                     img_h: int, gray_ref: Optional[torch.Tensor] = None) -> torch.Tensor:
    """points_3d (4, P), cam_inv (ncam, 4, 4), Kmat (ncam, 3, 3) -> (ncam, 2, P) normalised (x, y).
    gray_ref (ncam, C, H, W) uint8, optional: the reference images of the grey-pixel mask (remove_ref_in_gray)."""
    _require_gpu(points_3d, cam_inv, Kmat, gray_ref)
    pts = points_3d.float().contiguous()
    ci, km = cam_inv.float().contiguous(), Kmat.float().contiguous()
    ncam, P = ci.shape[0], pts.shape[1]
    out = torch.empty(ncam, 2, P, device=pts.device, dtype=torch.float32)
    if gray_ref is None:
        _lib.check(_lib.lib().bevr_project_bev_grid(_ptr(pts), _ptr(ci), _ptr(km), _ptr(out), ncam, P, img_w, img_h,
                                                    _stream()), "bevr_project_bev_grid")
        return out
    if gray_ref.dtype != torch.uint8 or gray_ref.dim() != 4 or gray_ref.shape[0] != ncam:
        raise ValueError("gray_ref must be a (ncam, C, H, W) uint8 tensor")
    ref = gray_ref.contiguous()
    _lib.check(_lib.lib().bevr_project_bev_grid_masked(_ptr(pts), _ptr(ci), _ptr(km), _ptr(out), ncam, P, img_w, img_h,
                                                       _ptr(ref), ref.shape[1], ref.shape[2], ref.shape[3], _stream()),
               "bevr_project_bev_grid_masked")
    return out


class _Corr(torch.autograd.Function):
    """D = 2 - 2 cam map^T on raw or L2-normalised rows."""

    @staticmethod
    def forward(ctx, cam, mp, normalize: bool):
        _require_gpu(cam, mp)
        cam, mp = cam.float().contiguous(), mp.float().contiguous()
        n, E = cam.shape
        m = mp.shape[0]
        D = torch.empty(n, m, device=cam.device, dtype=torch.float32)
        inc = torch.empty(n, device=cam.device, dtype=torch.float32)
        inm = torch.empty(m, device=cam.device, dtype=torch.float32)
        same = cam.data_ptr() == mp.data_ptr() and n == m
        _lib.check(KERNEL_TIMER.run("bevr_corr_fwd", 0.0, _lib.lib().bevr_corr_fwd, _ptr(cam), _ptr(mp), _ptr(D), _ptr(inc),
                                    _ptr(inm), n, m, E, int(normalize), _stream(),
                                    nbytes=4.0 * E * (n if same else n + m)), "bevr_corr_fwd")
        ctx.normalize = normalize
        ctx.save_for_backward(cam, mp, D, inc, inm)
        return D

    @staticmethod
    def backward(ctx, dD):
        cam, mp, D, inc, inm = ctx.saved_tensors
        n, E = cam.shape
        m = mp.shape[0]
        dcam, dmap = torch.empty_like(cam), torch.empty_like(mp)
        # algorithmic (compulsory) HBM bytes: both operands read once, both gradients written once; one matrix correlated
        # with itself (the retrieval losses): read once, and the two sides' SUM written once (csrc/corr.hip)
        same = cam.data_ptr() == mp.data_ptr() and n == m and n <= 64 and E % 4 == 0
        _lib.check(KERNEL_TIMER.run("bevr_corr_bwd", 0.0, _lib.lib().bevr_corr_bwd, _ptr(cam), _ptr(mp), _ptr(D),
                                    _ptr(dD.contiguous()), _ptr(inc), _ptr(inm), _ptr(dcam), _ptr(dmap), n, m, E,
                                    int(ctx.normalize), _stream(), nbytes=4.0 * E * (2 * n if same else 2 * (n + m))),
                   "bevr_corr_bwd")
        # same: dcam holds d/dcam + d/dmap (one gradient for the one tensor both arguments are)
        return dcam, (None if same else dmap), None


def pairwise_corr(cam: torch.Tensor, mp: torch.Tensor, normalize: bool = False) -> torch.Tensor:
    """2 - 2 cam map^T (train.py:554); normalize=True L2-normalises rows first (retrieval losses)."""
    return _Corr.apply(cam, mp, normalize)


def recall_rank(D: torch.Tensor) -> torch.Tensor:
    """rank[k] = #{i : D[i, k] < D[k, k]}  (train.py:559-563)."""
    _require_gpu(D)
    D = D.float().contiguous()
    n = D.shape[0]
    rank = torch.empty(n, device=D.device, dtype=torch.int32)
    _lib.check(_lib.lib().bevr_recall_rank(_ptr(D), _ptr(rank), n, _stream()), "bevr_recall_rank")
    return rank


# --------------------------------------------------------------------------------------------------
# fused offset heads (csrc/offset_head.hip)
# --------------------------------------------------------------------------------------------------
def offset_head_supported(cg: int, mx: int, dout: int) -> bool:
    """shapes the fused kernel covers: <= 64 input channels per group; SCA form (Mx == Dout <= 8) or TSA form (1, 2)."""
    return cg <= 64 and ((mx == dout and 1 <= mx <= 8) or (mx == 1 and dout == 2))


class _OffsetHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, gamma, beta, W3, groups, eps):
        _require_gpu(x, gamma, beta, W3)
        x = x.float().contiguous()
        B, H, W, Cc = x.shape
        g = groups
        cg = Cc // g
        K = gamma.numel()
        mx, dout = K // cg, W3.shape[0]
        prm = [None if t is None else t.detach().float().contiguous() for t in (w0, b0, gamma, beta, W3)]
        P = B * H * W
        out = torch.empty(g, B, H, W, dout, device=x.device, dtype=torch.float32)
        L = _lib.lib()
        for gi in range(g):
            xp = C.c_void_p(x.data_ptr() + gi * cg * 4)
            _lib.check(L.bevr_offset_head_fwd(xp, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(prm[3]), _ptr(prm[4]),
                                              _ptr(out[gi]), P, cg, Cc, mx, dout, float(eps), _stream()),
                       "bevr_offset_head_fwd")
        ctx.save_for_backward(x, *[t for t in prm if t is not None])
        ctx.meta = (g, cg, mx, dout, float(eps), w0 is not None, b0 is not None)
        return out.permute(1, 0, 2, 3, 4).reshape(B * g, H, W, dout)

    @staticmethod
    def backward(ctx, dout_):
        g, cg, mx, dout, eps, has_w0, has_b0 = ctx.meta
        saved = list(ctx.saved_tensors)
        x = saved.pop(0)
        w0 = saved.pop(0) if has_w0 else None
        b0 = saved.pop(0) if has_b0 else None
        gamma, beta, W3 = saved
        B, H, W, Cc = x.shape
        P = B * H * W
        do = dout_.float().reshape(B, g, H, W, dout).permute(1, 0, 2, 3, 4).contiguous()
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw0 = torch.zeros_like(w0) if has_w0 else None
        db0 = torch.zeros_like(b0) if has_b0 else None
        dga, dbe, dW3 = torch.zeros_like(gamma), torch.zeros_like(beta), torch.zeros_like(W3)
        L = _lib.lib()
        for gi in range(g):
            xp = C.c_void_p(x.data_ptr() + gi * cg * 4)
            dxp = None if dx is None else C.c_void_p(dx.data_ptr() + gi * cg * 4)
            _lib.check(L.bevr_offset_head_bwd(xp, _ptr(w0), _ptr(b0), _ptr(gamma), _ptr(beta), _ptr(W3), _ptr(do[gi]), dxp,
                                              _ptr(dw0), _ptr(db0), _ptr(dga), _ptr(dbe), _ptr(dW3), P, cg, Cc, mx, dout,
                                              eps, _stream()), "bevr_offset_head_bwd")
        return dx, dw0, db0, dga, dbe, dW3, None, None


def offset_head(x: torch.Tensor, w0, b0, gamma: torch.Tensor, beta: torch.Tensor, W3: torch.Tensor, groups: int = 1,
                eps: float = 1e-5) -> torch.Tensor:
    """Fused offset head on a channels-last map.  x (B, H, W, C); the C channels are `groups` groups that share the
    head; w0, b0 (Cg*Mx,) depthwise 1x1 weights with channel multiplier Mx, or None (no expansion: z = x);
    gamma, beta (Cg*Mx,) LayerNorm; W3 (Dout, Cg*Mx) pointwise, no bias.  Returns (B*groups, H, W, Dout), rows in
    the reference's "(b g)" order.  Replaces model/SCA_deform_attn.py:56-77 / model/TSA_deform_attn.py:54-68 (tail)."""
    return _OffsetHead.apply(x, w0, b0, gamma, beta, W3, groups, eps)


class _LayerNorm(torch.autograd.Function):
    """csrc/layernorm.hip: LayerNorm over the last axis of contiguous (.., C) rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _require_gpu(x, gamma, beta)
        x = x.float().contiguous()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        gamma, beta = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        _lib.check(_lib.lib().bevr_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, Cc,
                                                float(eps), _stream()), "bevr_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        Cc = x.shape[-1]
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dg = torch.zeros(Cc, device=x.device, dtype=torch.float32)
        db = torch.zeros_like(dg)
        _lib.check(_lib.lib().bevr_layernorm_bwd(_ptr(x), _ptr(gamma), _ptr(dy), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dg),
                                                _ptr(db), x.numel() // Cc, Cc, _stream()), "bevr_layernorm_bwd")
        return dx, dg, db, None


def layer_norm_supported(C: int) -> bool:
    c4 = C // 4
    return C % 4 == 0 and 0 < c4 <= 64 and (c4 & (c4 - 1)) == 0


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """LayerNorm over the last axis of a channels-last tensor (.., C): y = (x - mean) / sqrt(var + eps) * gamma + beta
    (biased variance, as nn.LayerNorm).  Replaces F.layer_norm in LayerNormProxy (model/model_utils.py:37-49)."""
    return _LayerNorm.apply(x, gamma, beta, eps)


class _KeyPositions(torch.autograd.Function):
    """csrc/keypos.hip: offset-head outputs -> key positions in the attention's key order (one launch), and the adjoint."""

    @staticmethod
    def forward(ctx, off, ref, order, B, G, sca_SD, use_tanh, sy, sx):
        _require_gpu(off, ref, order)
        off, ref = off.float().contiguous(), ref.float().contiguous()
        V, P = off.shape[0], off.shape[1]
        N = ref.shape[1]
        S, D = sca_SD if sca_SD else (0, 0)
        order = None if order is None else order.to(torch.int32).contiguous()
        pos = torch.empty(B, V, G, N, 2, device=off.device, dtype=torch.float32)
        ctx.args = (V, P, G, N, 1 if sca_SD else 0, S, D, 1 if use_tanh else 0, float(sy), float(sx))
        _lib.check(_lib.lib().bevr_key_positions_fwd(_ptr(off), _ptr(ref), _ptr(order), _ptr(pos), *ctx.args, _stream()),
                   "bevr_key_positions_fwd")
        ctx.save_for_backward(off, ref, order)
        return pos

    @staticmethod
    def backward(ctx, dpos):
        off, ref, order = ctx.saved_tensors
        doff = torch.empty_like(off)
        _lib.check(_lib.lib().bevr_key_positions_bwd(_ptr(off), _ptr(ref), _ptr(order), _ptr(dpos.float().contiguous()),
                                                      _ptr(doff), *ctx.args, _stream()), "bevr_key_positions_bwd")
        return doff, None, None, None, None, None, None, None, None


def key_positions(off: torch.Tensor, ref: torch.Tensor, order: Optional[torch.Tensor], batch: int, groups: int,
                  sca_SD=None, use_tanh: bool = True, sy: float = 1.0, sx: float = 1.0) -> torch.Tensor:
    """off (V, B*g, ...) offset-head outputs of V views -- SCA (sca_SD = (S, D)): channels-last (S, S, D) blocks whose
    even / odd BEV rows are the y / x offsets of key row h, key column w D + d; TSA (sca_SD None): (N, 2) blocks --,
    ref (V, N, 2) reference points in (y, x), order (V, N) the static key order or None.  Returns (B, V, g, N, 2):
    tanh(off) * (sy, sx) + ref (use_tanh) or clamp(off + ref, -1, 1), gathered into the key order.
    Replaces model/SCA_deform_attn.py:248-277 and model/TSA_deform_attn.py:170-196."""
    return _KeyPositions.apply(off, ref, order, batch, groups, sca_SD, use_tanh, sy, sx)


# --------------------------------------------------------------------------------------------------
# ego-motion warp of the history BEV (csrc/warp.hip)
# --------------------------------------------------------------------------------------------------
class _AffineWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, theta):
        _require_gpu(img, theta)
        img, theta = img.float().contiguous(), theta.float().contiguous()
        B, Cc, H, W = img.shape
        out = torch.empty_like(img)
        _lib.check(_lib.lib().bevr_affine_warp_fwd(_ptr(img), _ptr(theta), _ptr(out), B, Cc, H, W, _stream()),
                   "bevr_affine_warp_fwd")
        ctx.save_for_backward(theta)
        return out

    @staticmethod
    def backward(ctx, dout):
        (theta,) = ctx.saved_tensors
        dout = dout.float().contiguous()
        B, Cc, H, W = dout.shape
        dimg = torch.zeros_like(dout)
        _lib.check(_lib.lib().bevr_affine_warp_bwd(_ptr(dout), _ptr(theta), _ptr(dimg), B, Cc, H, W, _stream()),
                   "bevr_affine_warp_bwd")
        return dimg, None


def affine_theta(angle_rad: torch.Tensor, translate_xy: torch.Tensor) -> torch.Tensor:
    """(B, 6) inverse affine matrices of torchvision.transforms.functional.affine(angle, translate, scale=1, shear=0)
    about the image centre, in pixel units: [cos, sin, -cos tx - sin ty; -sin, cos, sin tx - cos ty].  Built on the
    device from pose tensors (the reference's per-sample math.degrees / .item() round trips, model/encoder.py:431-453)."""
    c, s = torch.cos(angle_rad), torch.sin(angle_rad)
    tx, ty = translate_xy[:, 0], translate_xy[:, 1]
    return torch.stack((c, s, -c * tx - s * ty, -s, c, s * tx - c * ty), 1).float()


def affine_warp(img: torch.Tensor, angle_rad: torch.Tensor, translate_xy: torch.Tensor) -> torch.Tensor:
    """Batched torchvision F.affine(img[b], degrees(angle[b]), translate[b], 1.0, 0, BILINEAR, fill=0) on (B, C, H, W):
    rotation about the image centre then translation in pixels; bilinear, zero padding, and the second attenuation
    by the resampled ones channel that torchvision applies when a fill value is given."""
    return _AffineWarp.apply(img, affine_theta(angle_rad.to(img.device), translate_xy.to(img.device)))


# --------------------------------------------------------------------------------------------------
# depthwise k x k convolution of the EncoderLayer glue (csrc/dwconv.hip)
# --------------------------------------------------------------------------------------------------
class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, nhwc: bool):
        _require_gpu(x, weight)
        L = _lib.lib()
        x = x.contiguous()
        w = weight.contiguous()
        if nhwc:
            B, H, W, Cc = x.shape
        else:
            B, Cc, H, W = x.shape
        k = w.shape[-1]
        y = torch.empty_like(x)
        _lib.check(L.bevr_dwconv_fwd(_ptr(x), _ptr(w), _ptr(bias.contiguous()) if bias is not None else None, _ptr(y),
                                     B, H, W, Cc, k, int(nhwc), 0, _stream()), "bevr_dwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.meta = (B, H, W, Cc, k, nhwc, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, Cc, k, nhwc, has_bias = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.bevr_dwconv_fwd(_ptr(dy), _ptr(w), None, _ptr(dx), B, H, W, Cc, k, int(nhwc), 1, _stream()),
                       "bevr_dwconv_fwd(flip)")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw = torch.zeros_like(w)
            db = torch.zeros(Cc, device=x.device, dtype=x.dtype) if has_bias else None
            _lib.check(L.bevr_dwconv_bwd_w(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db) if db is not None else None,
                                           B, H, W, Cc, k, int(nhwc), _stream()), "bevr_dwconv_bwd_w")
        return dx, dw, db, None


class _DwResGelu(torch.autograd.Function):
    """dwconv_res_gelu below (csrc/dwconv.hip, bevr_dwconv_res_gelu)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, weight)
        L = _lib.lib()
        x, w = x.contiguous(), weight.contiguous()
        b = bias.contiguous() if bias is not None else None
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        _lib.check(L.bevr_dwconv_res_gelu(_ptr(x), _ptr(w), _ptr(b), None, _ptr(y), B, H, W, Cc, 3, 1, _stream()),
                   "bevr_dwconv_res_gelu(1)")
        ctx.save_for_backward(x, w, b)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, w, b = ctx.saved_tensors
        L = _lib.lib()
        B, H, W, Cc = x.shape
        dout = dout.contiguous()
        # the gradient at the pre-activation x + conv(x) + bias, which is recomputed (one kernel) instead of saved
        g = torch.empty_like(x)
        _lib.check(L.bevr_dwconv_res_gelu(_ptr(x), _ptr(w), _ptr(b), _ptr(dout), _ptr(g), B, H, W, Cc, 3, 2, _stream()),
                   "bevr_dwconv_res_gelu(2)")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.bevr_dwconv_res_gelu(_ptr(g), _ptr(w), None, None, _ptr(dx), B, H, W, Cc, 3, 3, _stream()),
                       "bevr_dwconv_res_gelu(3)")
        if ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2]):
            dw = torch.zeros_like(w)
            db = torch.zeros(Cc, device=x.device, dtype=x.dtype) if b is not None else None
            _lib.check(L.bevr_dwconv_bwd_w(_ptr(x), _ptr(g), _ptr(dw), _ptr(db), B, H, W, Cc, 3, 1, _stream()),
                       "bevr_dwconv_bwd_w")
        return dx, dw, db


def dwconv_res_gelu_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] % 4 == 0 and weight.shape[-1] == 3
            and weight.shape[-2] == 3 and weight.shape[1] == 1 and weight.shape[0] == x.shape[-1]
            and os.environ.get("BEVR_FUSED_MLP", "1") != "0")


def dwconv_res_gelu(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """gelu(x + depthwise3x3(x) + bias) on a channels-last (B, H, W, C) tensor, GELU in its erf form: the middle of the
    layer MLPs (reference model/model_utils.py:51-59: `act(x + dwc(x))`) as ONE kernel in the forward (three stock passes:
    convolution, add, GELU) and three in the backward (pre-activation gradient recomputed from x, input gradient with
    the residual folded in, weight gradient), nothing saved but x.  BEVR_FUSED_MLP=0: the unfused chain."""
    return _DwResGelu.apply(x, weight, bias)


def depthwise_conv(x: torch.Tensor, weight: torch.Tensor, bias, nhwc: bool) -> torch.Tensor:
    """Depthwise k x k (k odd <= 5), stride 1, 'same' padding.  x (B,H,W,C) if nhwc else (B,C,H,W) float32;
    weight (C,1,k,k) as in nn.Conv2d(groups=C); bias (C,) or None."""
    return _DwConv.apply(x, weight, bias, nhwc)

