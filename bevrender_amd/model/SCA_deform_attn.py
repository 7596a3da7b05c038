"""Spatial cross-attention (the camera -> BEV "lift") on the gfx950 kernels.

Counterpart of the reference's model/SCA_deform_attn.py (:14-165 constructor, :180-421 forward): same
class name, constructor arguments, forward signature and parameter names.  Differences, all forced by
the reference itself:
  * every view v < n_views gets an offset head `conv_offset_m{v}` of the m0 form (D output channels);
    the reference's m1/m2 heads emit 2*D channels and raise in its own rearrange, so n_views > 1 has no
    reference semantics.  Heads m1/m2 that a n_views < 3 model does not use are still created with the
    reference's shapes so its state_dict loads unchanged.
  * the batch size is read from the tensors (data-parallel shards), not from the constructor.
All views are batched into one sampling launch and one attention launch.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, resolve_precision
from .model_utils import LayerNormProxy, attention_dropout, trunc_normal_

# keys per view and call that move from the cell segment's sparse tail to the region kernels (ops.cell_order)
CELL_TAIL = int(os.environ.get("BEVR_CELL_TAIL", "128"))


class SCADeformableAttention(nn.Module):
    def __init__(self, bev_feat_shape, bev_depth_dim, dim_embed, n_heads, n_groups, stride, kernel_size,
                 scale_offset_range, batch_size, n_views=3, attn_drop_rate=0.0, proj_drop_rate=0.0,
                 data_type=torch.float32, logger=None, precision=None):
        super().__init__()
        self.n_channel_per_head = dim_embed // n_heads
        self.scale = self.n_channel_per_head ** -0.5
        self.n_heads = n_heads
        self.embed_dim = self.n_channel_per_head * n_heads
        self.n_groups = n_groups
        self.n_channel_per_group = self.embed_dim // n_groups
        self.n_heads_per_group = n_heads // n_groups
        self.query_height = self.query_width = bev_feat_shape
        self.bev_depth_dim = bev_depth_dim
        self.batch_size = batch_size
        self.scale_offset_range = scale_offset_range
        self.kernel_size, self.stride, self.n_views = kernel_size, stride, n_views
        self.data_type, self.logger = data_type, logger
        self.offset_range_factor = 5.0
        self.precision = resolve_precision(precision)
        # reference :155-156: nn.Dropout on the softmax weights and on the projected output (training mode only)
        self.attn_drop_rate, self.proj_drop_rate = float(attn_drop_rate), float(proj_drop_rate)
        cg, D = self.n_channel_per_group, bev_depth_dim

        def head(out_ch):
            return nn.Sequential(nn.Conv2d(cg, cg * D, 1, 1, 0, groups=cg), LayerNormProxy(cg * D), nn.GELU(),
                                 nn.Conv2d(cg * D, out_ch, 1, 1, 0, bias=False))

        for v in range(max(n_views, 3)):
            used = v < n_views
            setattr(self, f"conv_offset_m{v}", head(D if (used or v == 0) else 2 * D))
        C = self.embed_dim
        self.proj_q = nn.Conv2d(C, C, 1)      # unused by the reference forward (:304-306)
        self.proj_k = nn.Conv2d(C, C, 1)
        self.proj_v = nn.Conv2d(C, C, 1)
        self.proj_out = nn.Conv2d(C * n_views, C, 1)
        self.proj_views = nn.Conv2d(cg * n_views, cg, 1)   # unused, state_dict parity
        self.rpe_table = nn.Parameter(torch.zeros(n_heads, 2 * bev_feat_shape - 1, 2 * bev_feat_shape * D - 1))
        trunc_normal_(self.rpe_table, std=0.01)

    @staticmethod
    def _offset_head(head, q_nhwc, groups):
        """The reference's offset head (1x1 depthwise C -> C*D, LayerNorm, GELU, 1x1 -> D; :56-77) on the channels-last
        query (B, S, S, C) whose `groups` channel groups share the head.  Returns (B*g, S, S, D_out).
        One fused HIP kernel per call (ops.offset_head, csrc/offset_head.hip: the C*D expansion of a pixel never leaves
        the wave's registers); shapes the kernel does not cover (more than 64 channels per group, D > 8) run the same
        arithmetic as stock channels-last ops."""
        dw, norm, act, pw = head[0], head[1], head[2], head[3]
        mult = dw.out_channels // dw.in_channels
        if q_nhwc.is_cuda and ops.offset_head_supported(dw.in_channels, mult, pw.out_channels):
            return ops.offset_head(q_nhwc, dw.weight.flatten(), dw.bias, norm.norm.weight, norm.norm.bias,
                                   pw.weight.flatten(1), groups, norm.norm.eps)
        B, S1, S2, Cc = q_nhwc.shape
        qh = q_nhwc.reshape(B, S1, S2, groups, Cc // groups).permute(0, 3, 1, 2, 4).reshape(B * groups, S1, S2, -1, 1)
        # depthwise 1x1 with channel multiplier: out[c * mult + m] = q[c] * w[c * mult + m] + b[c * mult + m], one pass
        y = torch.addcmul(dw.bias.view(-1, mult), qh, dw.weight.view(-1, mult)).flatten(-2)
        y = F.layer_norm(y, norm.norm.normalized_shape, norm.norm.weight, norm.norm.bias, norm.norm.eps)
        return F.linear(act(y), pw.weight.flatten(1), pw.bias)

    def key_positions(self, query, reference_points, key_order=None):
        """(B, V, g, N, 2) key positions (y, x): offset head of each view, even BEV rows -> y-offset of key
        row h, odd rows -> x-offset, key column w*D + d (reference :219-277), in the static key order `key_order`
        (V, N) if given."""
        B, C, S, _ = query.shape
        g, D, V = self.n_groups, self.bev_depth_dim, self.n_views
        Hk, Wk = S // 2, S * D
        q_nhwc = query.permute(0, 2, 3, 1)            # the LayerNormProxy output underneath: contiguous, no copy
        offs = [self._offset_head(getattr(self, f"conv_offset_m{v}"), q_nhwc, g) for v in range(V)]   # (B*g, S, S, D) each
        fac = self.offset_range_factor if self.scale_offset_range else 1.0
        rng_y, rng_x = fac / (Hk - 1.0), fac / (Wk - 1.0)
        if query.is_cuda and (B == 1 or reference_points.stride(0) == 0):
            # one HIP pass over all views: row split, tanh * range, + reference, gathered into the static key order
            # (the references are the projector's, the same for every sample: SpatialCrossAttn broadcasts them as a view)
            ref = reference_points[0, :, :, :, (1, 0)].reshape(V, Hk * Wk, 2)
            return ops.key_positions(torch.stack(offs, 0), ref, key_order, B, g, sca_SD=(S, D),
                                     use_tanh=bool(self.scale_offset_range), sy=rng_y, sx=rng_x)
        ref = reference_points[..., (1, 0)]                                  # (B, V, Hk, Wk, 2) -> (y, x)
        outs = []
        for v in range(V):
            # "(b g) d (h n) w -> (b g) n h (w d)", n = 2
            off = offs[v].reshape(B * g, Hk, 2, S, D).permute(0, 2, 1, 3, 4).reshape(B * g, 2, Hk, Wk)
            if self.scale_offset_range:
                rng = off.new_tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)]).reshape(1, 2, 1, 1)
                off = off.tanh() * rng * self.offset_range_factor
            pos = off.permute(0, 2, 3, 1).reshape(B, g, Hk, Wk, 2) + ref[:, v, None]
            if not self.scale_offset_range:
                pos = pos.clamp(-1.0, 1.0)
            outs.append(pos.reshape(B, g, Hk * Wk, 2))
        pos = torch.stack(outs, 1)
        if key_order is not None:
            N = Hk * Wk
            pos = pos.gather(3, key_order[None, :, None, :, None].expand(B, V, g, N, 2))
        return pos

    def _pinned_keys_tap(self, S, Hi, Wi):
        """Do the keys the projector pins to pixel (0, 0) all sample inside the top-left 4 x 3 feature pixels (the tap
        kernels' contract, csrc/attn_tap.h)?  Their position is the learned offset alone, tanh(.) * 5 / (Hk - 1) in y and
        5 / (Wk - 1) in x (reference :261-277): at most +-2.5 (Hi - 1) / (Hk - 1) x +-2.5 (Wi - 1) / (Wk - 1) pixels.
        Without the tanh range the positions are only clamped to the image: no bound, no tap kernels."""
        if not (self.scale_offset_range and ops.tap_supported(self.precision, self.n_groups)):
            return False
        Hk, Wk = S // 2, S * self.bev_depth_dim
        if Hk < 2 or Wk < 2:
            return False
        ymax = 0.5 * self.offset_range_factor / (Hk - 1.0) * (Hi - 1)
        xmax = 0.5 * self.offset_range_factor / (Wk - 1.0) * (Wi - 1)
        return (ymax < ops.TAP_R - 1 - 1e-3 or Hi <= ops.TAP_R) and (xmax < ops.TAP_C - 1 - 1e-3 or Wi <= ops.TAP_C)

    def forward(self, x, query, reference_points, wandb_log_dict, return_wandb_log=True, key_order=None,
                cell_split=None, split_is_pinned=False):
        """x (B, V, C, Hi, Wi); query (B, C, S, S); reference_points (B, V, S/2, S*D, 2) in (x, y).
        key_order (V, N) long, optional: a per-view permutation of the keys (SpatialCrossAttn passes the static
        k-d order of the camera projections); it changes no result, only the memory locality of the bias.
        cell_split, optional: the keys [cell_split, N) of every view (in that order) form a second key segment: they are
        sorted by rpe-table cell here, per call (their learned offsets decide the cell), and attended through the cell
        kernels (ops.attention_core), which handle ANY key set: the split changes no result.
        split_is_pinned=True is the caller's PROMISE that every key of [cell_split, N) has its reference exactly at
        (-1, -1) -- the pillar points the projector pins to pixel (0, 0); SpatialCrossAttn builds its split that way and
        passes True.  Only then, in the bf16 operand mode and with the offsets inside the tanh range (_pinned_keys_tap),
        does the segment run on the TAP kernels, which never form K and V and are exact only for keys that sample inside
        the top-left 4 x 3 feature pixels (csrc/attn_tap.h; the DEBUG build traps on a key outside them).  Without the
        promise a caller's own reference_points / split stay on the cell kernels."""
        B, V, C, Hi, Wi = x.shape
        S = query.shape[-1]
        if V != self.n_views:
            raise ValueError(f"expected {self.n_views} views, got {V}")
        g = self.n_groups
        pos = self.key_positions(query, reference_points.to(query.dtype), key_order)   # (B, V, g, N, 2), key order
        N = pos.shape[3]
        pos = pos.reshape(B * V * g, N, 2)
        drop = attention_dropout(self)      # (p, seed) in training mode with attn_drop_rate > 0, else None
        if drop is not None:
            cell_split = None               # the keep mask lives in the region kernels: every key goes there
        if cell_split is not None and cell_split < N and g == 1:
            # groups > 1: a key is one row of K built from all groups' samples, so the groups cannot be ordered
            # independently; the split is simply not used then
            n_tail = min(CELL_TAIL, max(0, N - cell_split - 1024))
            with torch.no_grad():
                a, b = ops.key_coords(pos[:, cell_split:], S, self.rpe_table.shape[-1], N - cell_split)
                dyn = ops.cell_order(a, b, n_tail)
            pos = torch.cat((pos[:, :cell_split], pos[:, cell_split:].gather(1, dyn[..., None].expand(-1, -1, 2))), 1)
            cell_split = cell_split + n_tail      # the keys of the sparsest cells join the region segment (ops.cell_order)
        else:
            cell_split = None
        # proj_k and proj_v as ONE GEMM over the sampled features (same arithmetic per output column; the features are
        # read once instead of twice)
        Wkv = torch.cat((self.proj_k.weight.flatten(1), self.proj_v.weight.flatten(1)), 0)
        bkv = torch.cat((self.proj_k.bias, self.proj_v.bias), 0)
        xf = x.reshape(B * V, C, Hi, Wi)
        if x.is_cuda and ops.kv_source_supported(C, self.n_heads, g, self.precision):
            # sampling, projection and operand packing as one kernel (csrc/kvproj.hip): neither the sampled features
            # nor the projected rows reach HBM.  The channels-last view of the backbone's output is read as it is
            feat = (xf if xf.dtype == torch.bfloat16 else xf.float()).permute(0, 2, 3, 1).contiguous()
            o = ops.attention_core(query, None, None, pos, self.rpe_table, heads=self.n_heads, groups=g, views=V,
                                   precision=self.precision, kv_source=(feat, Wkv, bkv), cell_split=cell_split,
                                   tap_source=bool(split_is_pinned) and cell_split is not None and self._pinned_keys_tap(S, Hi, Wi),
                                   attn_drop=drop, concat_views=True)
        else:
            xs = ops.sample_features(xf, pos, g)                                     # (B*V, N, C)
            kv = F.linear(xs, Wkv, bkv)
            o = ops.attention_core(query, None, None, pos, self.rpe_table, heads=self.n_heads, groups=g, views=V,
                                   precision=self.precision, kv=kv, cell_split=cell_split, attn_drop=drop,
                                   concat_views=True)
        # o: (B, S*S, V*C), the views side by side as proj_out contracts them (reference :415-420), written by the
        # attention's unpacking in one pass: no (B, V, M, C) -> (B, M, V C) permute copy in between
        out = ops.linear_rows(o, self.proj_out.weight.flatten(1), self.proj_out.bias)
        out = F.dropout(out, self.proj_drop_rate, self.training)                      # reference :420 (proj_drop)
        return out.permute(0, 2, 1).reshape(B, C, S, S), wandb_log_dict
