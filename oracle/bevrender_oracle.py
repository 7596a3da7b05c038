"""CPU oracle for the BEV-lift + correlation hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's
algorithm, written from the reference's source as cited per function (paths are
relative to the reference repo rpl-cmu/bevrender @ 2025-01-14).  It is NOT part
of the product: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it, and only as the checker / the timed CPU
baseline.  The product path (`bevrender_amd/`) never imports it and has no CPU
fallback.

Pinning: every function here is checked against golden vectors generated from
the reference's own modules (tests/golden/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_golden.py).  The retrieval losses (section 5) restate the
published algorithm of the un-vendored, un-pinned dependency
`pytorch_metric_learning`; the reference holds no test or fixture for them:
PARITY UNPINNED for those four functions only (the three retrieval losses and their
shared pairwise distance).  The eval-mode ego-motion warp (section 6) restates
`torchvision.transforms.functional.affine` (absent here): PARITY UNPINNED as well.

All functions are differentiable torch code (no in-place on inputs), so the
gradients the kernels must reproduce come from autograd on this file.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# 0. small shared pieces
# --------------------------------------------------------------------------- #
def layer_norm_proxy(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """LayerNormProxy: NCHW -> NHWC, LayerNorm(C), -> NCHW.  model/model_utils.py:51-59."""
    y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), weight, bias, eps)
    return y.permute(0, 3, 1, 2)


def normalized_grid(H: int, W: int, dtype, device=None) -> Tensor:
    """(H, W, 2) grid in (y, x) order, each axis i/(n-1)*2-1.
    model/SCA_deform_attn.py:167-178, model/TSA_deform_attn.py:98-109."""
    ry = torch.arange(0, H, dtype=dtype, device=device)
    rx = torch.arange(0, W, dtype=dtype, device=device)
    gy, gx = torch.meshgrid(ry, rx, indexing="ij")
    gy = gy / (H - 1.0) * 2.0 - 1.0
    gx = gx / (W - 1.0) * 2.0 - 1.0
    return torch.stack((gy, gx), -1)


def _offset_net(qg: Tensor, p: Dict[str, Tensor], prefix: str, stride: int, pad: int) -> Tensor:
    """depthwise conv -> LayerNormProxy -> GELU -> 1x1 conv (no bias).
    model/SCA_deform_attn.py:56-77 (1x1 depthwise, C -> C*D), model/TSA_deform_attn.py:54-68 (kxk strided)."""
    w0 = p[prefix + ".0.weight"]
    groups = qg.shape[1]
    y = F.conv2d(qg, w0, p[prefix + ".0.bias"], stride=stride, padding=pad, groups=groups)
    y = layer_norm_proxy(y, p[prefix + ".1.norm.weight"], p[prefix + ".1.norm.bias"])
    y = F.gelu(y)
    return F.conv2d(y, p[prefix + ".3.weight"], None)


def attention_core(q: Tensor, k: Tensor, v: Tensor, pos: Tensor, rpe_table: Tensor,
                   Hq: int, Wq: int, n_groups: int, scale: float, rows: Optional[Tensor] = None,
                   keep: Optional[Tensor] = None) -> Tensor:
    """Dense softmax attention with bilinear relative-position bias (materialised, as the reference does).

    q (B*h, c, M) raw query; k, v (B*h, c, N); pos (B*g, N, 2) key positions (y, x) in [-1,1] units;
    rpe_table (h, Ht, Wt).  Returns (B*h, c, M).
    model/SCA_deform_attn.py:331-413, model/TSA_deform_attn.py:245-333.

    rows (long tensor of query indices m = i*Wq + j, optional): attention rows are independent of each other, so
    the same arithmetic restricted to a subset of the queries is the reference's result for those queries; the
    return value is then (B*h, c, len(rows)).  This is how sizes whose (M x N) tensors do not fit in host memory
    (S = 200: 4e9 entries per head) are checked against the materialised formulation.
    """
    Bh, c, M = q.shape
    h = rpe_table.shape[0]
    B = Bh // h
    g = n_groups
    N = k.shape[-1]
    q_grid = normalized_grid(Hq, Wq, q.dtype, q.device).reshape(1, M, 2)
    if rows is not None:
        q = q[:, :, rows]
        q_grid = q_grid[:, rows]
        M = q.shape[-1]
    attn = torch.einsum("bcm,bcn->bmn", q, k) * scale
    disp = (q_grid.unsqueeze(2) - pos.reshape(B * g, 1, N, 2)) * 0.5          # (B*g, M, N, 2) (y,x)
    table = rpe_table[None].expand(B, -1, -1, -1).reshape(B * g, h // g, *rpe_table.shape[-2:])
    bias = F.grid_sample(table, disp[..., (1, 0)], mode="bilinear", align_corners=True)  # (B*g, h/g, M, N)
    attn = attn + bias.reshape(Bh, M, N)
    attn = F.softmax(attn, dim=2)
    if keep is not None:
        # attn_drop (model/SCA_deform_attn.py:402-409: nn.Dropout after the softmax) with an explicit mask: `keep`
        # (B*h, M, N) holds 0 or 1 / (1 - p), what nn.Dropout multiplies by in training mode
        attn = attn * keep
    return torch.einsum("bmn,bcn->bcm", attn, v)


def attention_core_streaming(q: Tensor, k: Tensor, v: Tensor, pos: Tensor, rpe_table: Tensor,
                             Hq: int, Wq: int, n_groups: int, scale: float, tile: int = 4096) -> Tensor:
    """The same function as attention_core, forward only, in the streaming form the kernels use (the algorithmic
    twin of csrc/attn_fwd.hip): keys are visited `tile` at a time with an online softmax, and the bias is the
    explicit 4-tap bilinear lookup at table coordinates ty = i + a_n, tx = j*rx + b_n
    (a_n = (1 - py_n)(Hq-1)/2, b_n = (1 - px_n)(Wt-1)/4, rx = (Wt-1)/(2(Wq-1))): algebraically
    grid_sample(align_corners=True, zeros padding) of (q_grid - pos)/2, model/SCA_deform_attn.py:365-389.
    Never holds more than (B*h, M, tile) values, so it runs where attention_core cannot."""
    Bh, c, M = q.shape
    h, Ht, Wt = rpe_table.shape
    B, g, N = Bh // h, n_groups, k.shape[-1]
    hpg = h // g
    ii = torch.arange(Hq, dtype=q.dtype).repeat_interleave(Wq)              # BEV row of query m = i*Wq + j
    jj = torch.arange(Wq, dtype=q.dtype).repeat(Hq)
    rx = (Wt - 1.0) / (2.0 * (Wq - 1.0))
    ry = (Ht - 1.0) / (2.0 * (Hq - 1.0))                                     # 1 for the reference's Ht = 2 Hq - 1
    m_run = torch.full((Bh, M), -float("inf"), dtype=q.dtype)
    l_run = torch.zeros(Bh, M, dtype=q.dtype)
    o_run = torch.zeros(Bh, c, M, dtype=q.dtype)
    tpad = F.pad(rpe_table, (1, 1, 1, 1))                                    # zero ring: out-of-range taps read 0
    for n0 in range(0, N, tile):
        n1 = min(N, n0 + tile)
        kk, vv = k[:, :, n0:n1], v[:, :, n0:n1]
        s = torch.einsum("bcm,bcn->bmn", q, kk) * scale
        p_ = pos.reshape(B, g, N, 2)[:, :, n0:n1]
        a = (1.0 - p_[..., 0]) * ((Ht - 1.0) / 4.0)                          # (B, g, n)
        b = (1.0 - p_[..., 1]) * ((Wt - 1.0) / 4.0)
        ty = (ii * ry)[None, None, :, None] + a[:, :, None, :]               # (B, g, M, n)
        tx = (jj * rx)[None, None, :, None] + b[:, :, None, :]
        y0, x0 = torch.floor(ty), torch.floor(tx)
        fy, fx = ty - y0, tx - x0
        y0 = y0.long().clamp(-1, Ht - 1) + 1                                  # index into the padded table
        x0 = x0.long().clamp(-1, Wt - 1) + 1
        inside = ((ty > -1) & (ty < Ht) & (tx > -1) & (tx < Wt)).to(q.dtype)
        bias = torch.empty(B, g, hpg, M, n1 - n0, dtype=q.dtype)
        for hh in range(hpg):
            for gi in range(g):
                t = tpad[gi * hpg + hh]
                yy, xx = y0[:, gi], x0[:, gi]
                val = (t[yy, xx] * (1 - fy[:, gi]) * (1 - fx[:, gi]) + t[yy + 1, xx] * fy[:, gi] * (1 - fx[:, gi])
                       + t[yy, xx + 1] * (1 - fy[:, gi]) * fx[:, gi] + t[yy + 1, xx + 1] * fy[:, gi] * fx[:, gi])
                bias[:, gi, hh] = val * inside[:, gi]
        s = s + bias.reshape(Bh, M, n1 - n0)
        m_new = torch.maximum(m_run, s.max(dim=2).values)
        alpha = torch.exp(m_run - m_new)
        pexp = torch.exp(s - m_new[..., None])
        l_run = l_run * alpha + pexp.sum(2)
        o_run = o_run * alpha[:, None, :] + torch.einsum("bmn,bcn->bcm", pexp, vv)
        m_run = m_new
    return o_run / l_run[:, None, :]


# --------------------------------------------------------------------------- #
# 1. TSA  (model/TSA_deform_attn.py:128-337; wrapper model/TSA.py:46-55 is a pass-through)
# --------------------------------------------------------------------------- #
def tsa_key_positions(p: Dict[str, Tensor], query: Tensor, n_groups: int, kernel_size: int, stride: int,
                      scale_offset_range: bool) -> Tensor:
    """offset net + tanh range + regular reference grid -> (B*g, Hk, Wk, 2) in (y, x).
    model/TSA_deform_attn.py:158-196."""
    B, C, H, W = query.shape
    g = n_groups
    pad = kernel_size // 2 if kernel_size != stride else 0
    qg = query.reshape(B * g, C // g, H, W)
    off = _offset_net(qg, p, "conv_offset", stride, pad)                      # (B*g, 2, Hk, Wk)
    Hk, Wk = off.shape[-2:]
    if scale_offset_range:
        rng = torch.tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)], dtype=off.dtype).reshape(1, 2, 1, 1)
        off = off.tanh() * rng * 0.5
    off = off.permute(0, 2, 3, 1)
    ref = normalized_grid(Hk, Wk, off.dtype)[None]
    pos = off + ref
    if not scale_offset_range:
        pos = pos.clamp(-1.0, 1.0)
    return pos


def tsa_forward(p: Dict[str, Tensor], query: Tensor, prev_bev: Optional[Tensor], *, n_heads: int,
                n_groups: int = 1, kernel_size: int = 3, stride: int = 1,
                scale_offset_range: bool = True, rows: Optional[Tensor] = None) -> Tensor:
    """TSADeformableAttention.forward(x=prev_bev, query).  `p` uses the reference's state_dict names.
    rows (long tensor of query indices m = i*W + j, optional): evaluate the module's output at those BEV positions only
    (attention rows and the 1x1 proj_out are independent per position) and return (B, C, len(rows)): how the module is
    checked at sizes whose (M x N) tensors do not fit in host memory."""
    x = query.clone() if prev_bev is None else prev_bev                      # :142-143
    B, C, H, W = x.shape
    h, g = n_heads, n_groups
    c = C // h
    pos = tsa_key_positions(p, query, g, kernel_size, stride, scale_offset_range)
    Hk, Wk = pos.shape[1:3]
    N = Hk * Wk
    xs = F.grid_sample(x.reshape(B * g, C // g, H, W), pos[..., (1, 0)], mode="bilinear",
                       align_corners=True).reshape(B, C, 1, N)                # :210-217
    q = query.reshape(B * h, c, H * W)                                        # :220 (raw query, proj_q unused)
    k = F.conv2d(xs, p["proj_k.weight"], p["proj_k.bias"]).reshape(B * h, c, N)
    v = F.conv2d(xs, p["proj_v.weight"], p["proj_v.bias"]).reshape(B * h, c, N)
    out = attention_core(q, k, v, pos.reshape(B * g, N, 2), p["rpe_table"], H, W, g, c ** -0.5, rows=rows)
    if rows is not None:
        return F.conv1d(out.reshape(B, C, len(rows)), p["proj_out.weight"].flatten(2), p["proj_out.bias"])
    out = out.reshape(B, C, H, W)
    return F.conv2d(out, p["proj_out.weight"], p["proj_out.bias"])            # :336


# --------------------------------------------------------------------------- #
# 2. SCA  (model/SCA_deform_attn.py:180-421; wrapper model/SCA.py:60-110)
# --------------------------------------------------------------------------- #
def sca_key_positions(p: Dict[str, Tensor], query: Tensor, ref_view: Tensor, view_idx: int, n_groups: int,
                      depth_dim: int, scale_offset_range: bool) -> Tensor:
    """offset head m{view} -> even/odd-row split -> tanh range -> + camera reference.
    ref_view (B*g, S/2, S*D, 2) already in (y, x).  Returns (B*g, Hk, Wk, 2) (y, x).
    model/SCA_deform_attn.py:219-277.  Every view uses the m0 head form (D output channels): the
    reference's m1/m2 heads emit 2*D channels and raise in the rearrange at :248-255."""
    B, C, S, _ = query.shape
    g, D = n_groups, depth_dim
    qg = query.reshape(B * g, C // g, S, S)
    off = _offset_net(qg, p, f"conv_offset_m{view_idx}", 1, 0)               # (B*g, D, S, S)
    # "(b g) d (h n) w -> (b g) n h (w d)", n=2: even BEV rows -> y-offset, odd rows -> x-offset
    off = off.reshape(B * g, D, S // 2, 2, S).permute(0, 3, 2, 4, 1).reshape(B * g, 2, S // 2, S * D)
    Hk, Wk = off.shape[-2:]
    if scale_offset_range:
        rng = torch.tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)], dtype=off.dtype).reshape(1, 2, 1, 1)
        off = off.tanh() * rng * 5.0
    off = off.permute(0, 2, 3, 1)
    pos = off + ref_view
    if not scale_offset_range:
        pos = pos.clamp(-1.0, 1.0)
    return pos


def sca_forward(p: Dict[str, Tensor], x: Tensor, query: Tensor, reference_points: Tensor, *, n_heads: int,
                n_groups: int = 1, depth_dim: int = 5, scale_offset_range: bool = True,
                rows: Optional[Tensor] = None) -> Tensor:
    """SCADeformableAttention.forward.  x (B, V, C, Hi, Wi); reference_points (B, V, S/2, S*D, 2) in (x, y).
    rows: as tsa_forward -- the output at the selected BEV positions only, (B, C, len(rows))."""
    B, V, C, Hi, Wi = x.shape
    S = query.shape[-1]
    h, g = n_heads, n_groups
    c = C // h
    ref = reference_points[..., (1, 0)]                                       # :204 -> (y, x)
    ref = ref.repeat_interleave(g, dim=0)                                     # "b v h w n -> (b g) v h w n"
    outs = []
    for v_idx in range(V):
        pos = sca_key_positions(p, query, ref[:, v_idx], v_idx, g, depth_dim, scale_offset_range)
        Hk, Wk = pos.shape[1:3]
        N = Hk * Wk
        xs = F.grid_sample(x[:, v_idx].reshape(B * g, C // g, Hi, Wi), pos[..., (1, 0)], mode="bilinear",
                           align_corners=True).reshape(B, C, 1, N)            # :290-301
        q = query.reshape(B * h, c, S * S)                                    # :304-306
        k = F.conv2d(xs, p["proj_k.weight"], p["proj_k.bias"]).reshape(B * h, c, N)
        v = F.conv2d(xs, p["proj_v.weight"], p["proj_v.bias"]).reshape(B * h, c, N)
        o = attention_core(q, k, v, pos.reshape(B * g, N, 2), p["rpe_table"], S, S, g, c ** -0.5, rows=rows)
        outs.append(o.reshape(B, C, S, S) if rows is None else o.reshape(B, C, len(rows)))
    out = torch.cat(outs, dim=1)                                              # "b v c h w -> b (v c) h w"
    if rows is not None:
        return F.conv1d(out, p["proj_out.weight"].flatten(2), p["proj_out.bias"])
    return F.conv2d(out, p["proj_out.weight"], p["proj_out.bias"])            # :415-420


# --------------------------------------------------------------------------- #
# 3. BEV pillar grid -> camera pixels  (model/SCA.py:112-162, model/bev_cmr_proj.py:13-124)
# --------------------------------------------------------------------------- #
def sample_3d_points(bound: Dict[str, float], S: int, D: int, z_shift: float) -> Tensor:
    """Homogeneous pillar-centre grid (4, S/2, S, D), float32.  Closed form of the float aranges at
    model/SCA.py:130-148: X bins X/S*(2i+1), i< S/2; Y bins -Y+Y/S*(2j+1), j<S; Z bins -Z+Z/D*(2d+1)+z_shift."""
    X, Y, Z = float(bound["X"]), float(bound["Y"]), float(bound["Z"])
    xs, ys, zs = X / S, Y / S, Z / D
    # the reference builds these with torch.arange(start, end, step) in fp32: start + i*step
    gx = torch.arange(0 + xs, X + xs, xs * 2)
    gy = torch.arange(-Y + ys, Y + ys, ys * 2)
    gz = torch.arange(-Z + zs + z_shift, Z + zs + z_shift, zs * 2)
    assert gx.numel() == S // 2 and gy.numel() == S and gz.numel() == D, \
        "float arange produced an extra bin for this (bound, S) pair; see SURVEY 3.5"
    PX = gx[:, None, None].expand(S // 2, S, D)
    PY = gy[None, :, None].expand(S // 2, S, D)
    PZ = gz[None, None, :].expand(S // 2, S, D)
    return torch.stack((PX, PY, PZ, torch.ones_like(PX)), 0).contiguous()


def bev_grid_to_camera(points_3d: Tensor, imu_to_rgb: Sequence[np.ndarray], K: Sequence[np.ndarray],
                       img_width: int, img_height: int, ori_img_width: int, ori_img_height: int,
                       gray_ref: Optional[Sequence[Tensor]] = None) -> List[Tensor]:
    """Per camera (2, h, w, z) normalised (x, y).  K is NOT mutated here (the reference scales the caller's
    arrays in place, bev_cmr_proj.py:41-46; the scaled copy is what is used below).
    gray_ref: per camera a (C, H, W) uint8 reference image -> the remove_ref_in_gray mask of :114-122."""
    sx, sy = img_width / ori_img_width, img_height / ori_img_height
    _, h, w, z = points_3d.shape
    pts = points_3d.reshape(4, -1)
    out = []
    for T, Kc in zip(imu_to_rgb, K):
        Kc = np.array(Kc, dtype=np.float64, copy=True)
        Kc[0, 0] *= sx
        Kc[0, 2] *= sx
        Kc[1, 1] *= sy
        Kc[1, 2] *= sy
        Tm = torch.tensor(np.asarray(T)).float()
        Km = torch.tensor(Kc).float()
        cam = Tm.inverse() @ pts                                               # :72
        uv = Km[:, :3] @ cam[:3]                                               # :73
        uv = uv.div(uv[-1])[:2]                                                # :74
        iu = uv.to(torch.int32)                                                # :106 truncation toward zero
        mask = iu[1].ge(0) & iu[1].lt(img_height - 1) & iu[0].ge(0) & iu[0].lt(img_width - 1)
        if gray_ref is not None:                                               # :114-122
            ref_img = gray_ref[len(out)]
            ip = iu.masked_fill(~mask, 0).long()
            values = ref_img[:, ip[1], ip[0]]
            mask = mask & ((values == 128).sum(0) != 3)
        uv = uv.masked_fill(~mask, 0)                                          # :76
        u = uv[0] / (img_width - 1)
        v_ = uv[1] / (img_height - 1)
        uv = torch.stack((u, v_), 0) * 2 - 1                                   # :95-97
        out.append(uv.reshape(2, h, w, z))
    return out


def sca_reference_points(points_2d: List[Tensor], batch: int) -> Tensor:
    """stack views, "v n h w d -> b v h (w d) n".  model/SCA.py:78-85."""
    r = torch.stack(points_2d, 0)                                              # (V, 2, h, w, d)
    V, _, h, w, d = r.shape
    r = r.permute(0, 2, 3, 4, 1).reshape(V, h, w * d, 2)
    return r[None].expand(batch, -1, -1, -1, -1)


# --------------------------------------------------------------------------- #
# 4. EncoderLayer glue (caller of the hot path; model/encoder.py:339-411, train mode, DropPath rate 0)
# --------------------------------------------------------------------------- #
def transformer_mlp_with_conv(p: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """model/model_utils.py:6-35."""
    y = F.conv2d(x, p[prefix + ".linear1.0.weight"], p[prefix + ".linear1.0.bias"])
    dwc_w = p[prefix + ".dwc.weight"]
    y = y + F.conv2d(y, dwc_w, p[prefix + ".dwc.bias"], padding=1, groups=dwc_w.shape[0])
    y = F.gelu(y)
    return F.conv2d(y, p[prefix + ".linear2.0.weight"], p[prefix + ".linear2.0.bias"])


def _sub(p: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def encoder_layer_forward(p: Dict[str, Tensor], bev_query: Tensor, img_feat: Tensor, prev_bev: Optional[Tensor],
                          reference_points: Tensor, *, n_heads: int, n_groups: int, depth_dim: int, n_views: int,
                          kernel_size: int, stride: int, scale_offset_range: bool = True) -> Tensor:
    """EncoderLayer.forward in training mode (history warp skipped, encoder.py:366)."""
    B = bev_query.shape[0]
    ln = lambda t: layer_norm_proxy(t, p["layer_norm.norm.weight"], p["layer_norm.norm.bias"])
    x = bev_query
    w = p["tsa_local_percept_unit.weight"]
    x = x + F.conv2d(x, w, p["tsa_local_percept_unit.bias"], padding=1, groups=w.shape[0])      # :370
    x0 = x
    x = tsa_forward(_sub(p, "temporal_self_attn.temporal_deform_attn."), ln(x), prev_bev, n_heads=n_heads,
                    n_groups=n_groups, kernel_size=kernel_size, stride=stride,
                    scale_offset_range=scale_offset_range) + x0                                   # :372-379
    x0 = x
    x = transformer_mlp_with_conv(p, "tsa_mlp", ln(x)) + x0                                       # :385-387
    w = p["sca_local_percept_unit.weight"]
    x = x + F.conv2d(x, w, p["sca_local_percept_unit.bias"], padding=1, groups=w.shape[0])      # :390
    x0 = x
    feat = img_feat.reshape(B, n_views, *img_feat.shape[-3:])                                     # SCA.py:88-94
    x = sca_forward(_sub(p, "spatial_cross_attn.spatial_deform_attn."), feat, ln(x), reference_points,
                    n_heads=n_heads, n_groups=n_groups, depth_dim=depth_dim,
                    scale_offset_range=scale_offset_range) + x0                                   # :392-400
    x0 = x
    return transformer_mlp_with_conv(p, "sca_mlp", ln(x)) + x0                                    # :406-408


# --------------------------------------------------------------------------- #
# 5. ground <-> aerial correlation
# --------------------------------------------------------------------------- #
def get_recall(cam: np.ndarray, mp: np.ndarray):
    """Trainer.get_recall, train.py:551-572: D = 2 - 2 cam map^T; rank of the diagonal within its COLUMN;
    recall@{1,5,10} in percent."""
    dist = 2.0 - 2.0 * np.matmul(cam, mp.T)
    gt = np.diag(dist)
    rank = (dist < gt[None, :]).sum(axis=0)                                    # per column k
    return tuple(float((rank < i).mean() * 100.0) for i in (1, 5, 10))


def _pairwise_lp_normalized(emb: Tensor) -> Tensor:
    """pytorch_metric_learning LpDistance(normalize_embeddings=True, p=2, power=1) -- PARITY UNPINNED."""
    e = F.normalize(emb, p=2, dim=1)
    d2 = (2.0 - 2.0 * e @ e.t()).clamp_min(0)
    # sqrt with a safe gradient at 0 (the library uses torch.cdist)
    return torch.where(d2 > 0, torch.sqrt(d2.clamp_min(1e-16)), torch.zeros_like(d2))


def contrastive_loss(cam: Tensor, mp: Tensor, pos_margin: float = 0.0, neg_margin: float = 1.0) -> Tensor:
    """loss/contrastive_loss.py:10-19 -> pytorch_metric_learning.losses.ContrastiveLoss defaults
    (un-vendored, version un-pinned: PARITY UNPINNED).  labels = [0..B-1, 0..B-1];
    pos term relu(d - pos_margin), neg term relu(neg_margin - d), each averaged over its non-zero entries
    (AvgNonZeroReducer), summed."""
    B = cam.shape[0]
    emb = torch.cat((cam, mp), 0)
    lab = torch.cat((torch.arange(B), torch.arange(B)))
    d = _pairwise_lp_normalized(emb)
    same = lab[:, None] == lab[None, :]
    eye = torch.eye(2 * B, dtype=torch.bool)
    pos = F.relu(d - pos_margin)[same & ~eye]
    neg = F.relu(neg_margin - d)[~same]

    def avg_nonzero(t):
        nz = t > 0
        return t[nz].mean() if nz.any() else t.sum() * 0

    return avg_nonzero(pos) + avg_nonzero(neg)


def lifted_structure_loss(cam: Tensor, mp: Tensor, neg_margin: float = 1.0, pos_margin: float = 0.0) -> Tensor:
    """loss/lift_loss.py:13-22 -> pytorch_metric_learning.losses.LiftedStructureLoss(neg_margin=1, pos_margin=0)
    (PARITY UNPINNED).  For every positive pair (i, j), i<j... the library iterates ordered pairs:
    relu( logsumexp_{negatives of i or j}(neg_margin - d) + (d_ij - pos_margin) )^2 / 2, mean over pairs."""
    B = cam.shape[0]
    emb = torch.cat((cam, mp), 0)
    lab = torch.cat((torch.arange(B), torch.arange(B)))
    d = _pairwise_lp_normalized(emb)
    same = lab[:, None] == lab[None, :]
    eye = torch.eye(2 * B, dtype=torch.bool)
    losses = []
    for i in range(2 * B):
        for j in range(2 * B):
            if i == j or not same[i, j]:
                continue
            negs = torch.cat(((neg_margin - d[i])[~same[i]], (neg_margin - d[j])[~same[j]]))
            losses.append(F.relu(torch.logsumexp(negs, 0) + d[i, j] - pos_margin) ** 2 / 2.0)
    return torch.stack(losses).mean()


def triplet_margin_loss(cam: Tensor, mp: Tensor, miner_margin: float = 0.2, loss_margin: float = 0.05,
                        reducer_high: float = 0.3) -> Tensor:
    """loss/triplet_loss_metric.py:8-28 -> pytorch_metric_learning (un-vendored, un-pinned: PARITY UNPINNED):
      miner  TripletMarginMiner(margin=0.2, "semihard") on LpDistance(normalize_embeddings=True): every triplet
             (a, p, n) with label[a] == label[p], a != p, label[n] != label[a] and 0 < d_an - d_ap <= margin (no grad);
      loss   TripletMarginLoss(margin=0.05, distance=CosineSimilarity()): relu(cos_an - cos_ap + margin) per mined
             triplet (an inverted distance: larger = closer);
      reduce ThresholdReducer(high=0.3): mean over the triplet losses < high (zeros included), 0 if none passes;
      plus   LpRegularizer() (p=2) on the raw embeddings with the default weight 1: mean ||e||_2 here.  UNVERIFIED which
             reducer the library applies to that term when the loss is built with `reducer=`: it may copy the caller's
             reducer onto every sub-loss, in which case ThresholdReducer(high=0.3) would also gate the norms (and drop
             every embedding whose norm exceeds 0.3); this restatement assumes a plain mean.  The package is absent here
             and the reference holds no test or fixture for this loss, so nothing decides it.
    Written as the triple loop the library's index lists amount to."""
    B = cam.shape[0]
    emb = torch.cat((cam, mp), 0)
    lab = torch.cat((torch.arange(B), torch.arange(B)))
    with torch.no_grad():
        d = _pairwise_lp_normalized(emb)
    e = F.normalize(emb, p=2, dim=1)
    cos = e @ e.t()
    losses = []
    for a in range(2 * B):
        for p_ in range(2 * B):
            if p_ == a or lab[p_] != lab[a]:
                continue
            for n in range(2 * B):
                if lab[n] == lab[a]:
                    continue
                m = (d[a, n] - d[a, p_]).item()
                if 0 < m <= miner_margin:
                    losses.append(F.relu(cos[a, n] - cos[a, p_] + loss_margin))
    reg = emb.norm(p=2, dim=1).mean()
    if not losses:
        return reg + emb.sum() * 0
    losses = torch.stack(losses)
    keep = losses < reducer_high
    if int(keep.sum()) < 1:
        return reg + emb.sum() * 0
    return losses[keep].mean() + reg


def pairwise_corr(cam: Tensor, mp: Tensor) -> Tensor:
    """The explicit ground<->aerial correlation of train.py:554: 2 - 2 cam @ map^T."""
    return 2.0 - 2.0 * cam @ mp.t()


# --------------------------------------------------------------------------- #
# 6. ego-motion warp of the history BEV  (model/encoder.py:413-466) -- PARITY UNPINNED
# --------------------------------------------------------------------------- #
def tv_affine(img: Tensor, angle_deg: float, translate, fill: float = 0.0) -> Tensor:
    """torchvision.transforms.functional.affine(img (C,H,W), angle, translate, scale=1.0, shear=0,
    interpolation=BILINEAR, fill=fill) restated from torchvision's published tensor path (the package is absent from
    this image, version un-pinned by the reference: PARITY UNPINNED):
      _get_inverse_affine_matrix(center=(0,0), angle, translate, 1, (0,0)): with rot = radians(angle),
          M = [cos, sin, -cos tx - sin ty;  -sin, cos, sin tx - cos ty]
      _gen_affine_grid: pixel-centre grid X in linspace(-W/2 + 1/2, W/2 - 1/2, W) (Y alike), grid = [X Y 1] M^T / (W/2, H/2)
      _apply_grid_transform: with a fill value a ones channel is appended, grid_sample(bilinear, zeros,
          align_corners=False) resamples both, and (bilinear) out = img * mask + (1 - mask) * fill."""
    C, H, W = img.shape
    rot = math.radians(float(angle_deg))
    tx, ty = float(translate[0]), float(translate[1])
    c, s = math.cos(rot), math.sin(rot)
    m = [c, s, c * (-tx) + s * (-ty), -s, c, -s * (-tx) + c * (-ty)]
    theta = torch.tensor(m, dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W, dtype=img.dtype))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H, dtype=img.dtype).unsqueeze(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=img.dtype)
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2)
    x = torch.cat((img[None], torch.ones(1, 1, H, W, dtype=img.dtype)), 1)
    y = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = y[:, -1:]
    out = y[:, :-1] * mask + (1.0 - mask) * fill
    return out[0]


def project_history_bev_feat(bev: Tensor, vehicle_pose: Tensor) -> Tensor:
    """EncoderLayer.project_history_bev_feat, model/encoder.py:413-466: per sample, rotate by +prev_yaw and translate
    by (prev - cur) pixel offsets, then rotate by -cur_yaw: two chained bilinear resamplings with fill 0.
    vehicle_pose (B, 2, 3) = [pixel_x, pixel_y, yaw_rad] of the previous and the current frame."""
    outs = []
    for i in range(bev.shape[0]):
        prev_rot, curr_rot = vehicle_pose[i, :, 2]
        delta_x, delta_y, _ = vehicle_pose[i, 0] - vehicle_pose[i, 1]
        p = tv_affine(bev[i], math.degrees(prev_rot), (delta_x, delta_y))
        p = tv_affine(p, math.degrees(-curr_rot), (0.0, 0.0))
        outs.append(p)
    return torch.stack(outs, 0)
