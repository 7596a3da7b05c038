"""Temporal self-attention over sampled keys of the previous BEV, on the gfx950 kernels.

Counterpart of the reference's model/TSA_deform_attn.py (class name, constructor arguments, parameter
names and forward signature identical; :14-96, :128-337).  The offset head and the three 1x1
projections stay stock PyTorch (MIOpen / rocBLAS); the bilinear sampling of prev_bev and the whole
QK^T + RPE-bias + softmax + PV core run in libbevrender_hip.so (ops.sample_features, ops.attention_core).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, resolve_precision
from .model_utils import (LayerNormProxy, attention_dropout, _dw_ok, depthwise_conv2d, depthwise_conv2d_nhwc, normalized_grid,
                          trunc_normal_)


class TSADeformableAttention(nn.Module):
    def __init__(self, bev_feat_shape, dim_embed, n_heads, n_groups, stride, kernel_size, scale_offset_range,
                 batch_size, n_views=3, attn_drop_rate=0.0, proj_drop_rate=0.0, data_type=torch.float32,
                 logger=None, precision=None):
        super().__init__()
        self.n_channel_per_head = dim_embed // n_heads
        self.scale = self.n_channel_per_head ** -0.5
        self.n_heads = n_heads
        self.embed_dim = self.n_channel_per_head * n_heads
        self.n_groups = n_groups
        self.n_channel_per_group = self.embed_dim // n_groups
        self.n_heads_per_group = n_heads // n_groups
        self.bev_h = self.bev_w = bev_feat_shape
        self.batch_size = batch_size          # kept for API parity; the batch is read from the tensors
        self.scale_offset_range = scale_offset_range
        self.kernel_size, self.stride, self.n_views = kernel_size, stride, n_views
        self.data_type, self.logger = data_type, logger
        self.offset_range_factor = 0.5
        self.precision = resolve_precision(precision)
        # reference :90-91: nn.Dropout on the softmax weights and on the projected output (training mode only)
        self.attn_drop_rate, self.proj_drop_rate = float(attn_drop_rate), float(proj_drop_rate)
        pad = kernel_size // 2 if kernel_size != stride else 0
        cg = self.n_channel_per_group
        self.conv_offset = nn.Sequential(
            nn.Conv2d(cg, cg, kernel_size, stride, pad, groups=cg),
            LayerNormProxy(cg),
            nn.GELU(),
            nn.Conv2d(cg, 2, 1, 1, 0, bias=False),
        )
        C = self.embed_dim
        self.proj_q = nn.Conv2d(C, C, 1)       # constructed, never used by the reference forward (:220)
        self.proj_k = nn.Conv2d(C, C, 1)
        self.proj_v = nn.Conv2d(C, C, 1)
        self.proj_out = nn.Conv2d(C, C, 1)
        self.proj_views = nn.Conv2d(cg * n_views, cg, 1)   # unused, kept for state_dict parity
        self.rpe_table = nn.Parameter(torch.zeros(n_heads, 2 * self.bev_h - 1, 2 * self.bev_w - 1))
        trunc_normal_(self.rpe_table, std=0.01)
        self._order_cache = {}

    def _key_order(self, Hk, Wk, device):
        """Static k-d ordering of the regular key grid (compact 64-key steps; see ops.kd_key_order)."""
        key = (Hk, Wk, str(device))
        if key not in self._order_cache:
            ref = normalized_grid(Hk, Wk, torch.float64, "cpu").reshape(-1, 2).numpy()
            order = ops.kd_key_order(ref, self.bev_h, 2 * self.bev_w - 1)
            self._order_cache[key] = torch.from_numpy(order).to(device)
        return self._order_cache[key]

    def key_positions(self, query):
        """offset head -> tanh range -> + regular grid: (B*g, Hk*Wk, 2) in (y, x).  reference :158-196."""
        B, C, H, W = query.shape
        g = self.n_groups
        qg = query.reshape(B * g, C // g, H, W)
        co = self.conv_offset
        if g == 1 and query.is_cuda and _dw_ok(query, co[0]):
            # stride-1 head on the channels-last query (the LayerNormProxy output underneath): no layout copy
            z = depthwise_conv2d_nhwc(query.permute(0, 2, 3, 1), co[0]).permute(0, 3, 1, 2)
        else:
            z = depthwise_conv2d(qg, co[0])                               # (B*g, Cg, Hk, Wk): the strided k x k depthwise
        if z.is_cuda and ops.offset_head_supported(z.shape[1], 1, 2):
            # LayerNorm -> GELU -> 1x1 (Cg -> 2) fused per key-grid pixel (ops.offset_head, csrc/offset_head.hip)
            off = ops.offset_head(z.permute(0, 2, 3, 1), None, None, co[1].norm.weight, co[1].norm.bias,
                                  co[3].weight.flatten(1), 1, co[1].norm.eps).permute(0, 3, 1, 2)
        else:
            y = co[2](co[1](z))                                           # LayerNormProxy output: NHWC underneath
            off = F.linear(y.permute(0, 2, 3, 1), co[3].weight.flatten(1)).permute(0, 3, 1, 2)   # 1x1 conv as a GEMM
        Hk, Wk = off.shape[-2:]
        if off.is_cuda and Hk > 1 and Wk > 1:
            # tanh * range, + regular grid, gathered into the static key order: one HIP pass (ops.key_positions)
            fac = self.offset_range_factor if self.scale_offset_range else 1.0
            grid = normalized_grid(Hk, Wk, torch.float32, off.device).reshape(1, Hk * Wk, 2)
            pos = ops.key_positions(off.permute(0, 2, 3, 1).reshape(1, B * g, Hk * Wk, 2), grid,
                                    self._key_order(Hk, Wk, off.device)[None], B, g, sca_SD=None,
                                    use_tanh=bool(self.scale_offset_range), sy=fac / (Hk - 1.0), sx=fac / (Wk - 1.0))
            return pos.reshape(B * g, Hk * Wk, 2)
        if self.scale_offset_range:
            rng = off.new_tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)]).reshape(1, 2, 1, 1)
            off = off.tanh() * rng * self.offset_range_factor
        pos = off.permute(0, 2, 3, 1) + normalized_grid(Hk, Wk, off.dtype, off.device)[None]
        if not self.scale_offset_range:
            pos = pos.clamp(-1.0, 1.0)
        pos = pos.reshape(B * g, Hk * Wk, 2)
        if Hk > 1 and Wk > 1:
            pos = pos.index_select(1, self._key_order(Hk, Wk, pos.device))   # order-invariant for the softmax
        return pos

    def forward(self, x, query, wandb_log_dict, return_wandb_log=True):
        if x is None:                       # no history: self-attention on the query (:142-143)
            x = query
        B, C, H, W = x.shape
        pos = self.key_positions(query)
        # proj_k and proj_v as ONE GEMM over the sampled features (same arithmetic per output column; the features are
        # read once instead of twice)
        Wkv = torch.cat((self.proj_k.weight.flatten(1), self.proj_v.weight.flatten(1)), 0)
        bkv = torch.cat((self.proj_k.bias, self.proj_v.bias), 0)
        drop = attention_dropout(self)      # (p, seed) in training mode with attn_drop_rate > 0, else None
        if x.is_cuda and ops.kv_source_supported(C, self.n_heads, self.n_groups, self.precision):
            # sampling, projection and operand packing as one kernel (csrc/kvproj.hip)
            feat = (x if x.dtype == torch.bfloat16 else x.float()).permute(0, 2, 3, 1).contiguous()
            o = ops.attention_core(query, None, None, pos, self.rpe_table, heads=self.n_heads, groups=self.n_groups,
                                   views=1, precision=self.precision, kv_source=(feat, Wkv, bkv), attn_drop=drop)
        else:
            xs = ops.sample_features(x, pos, self.n_groups)                              # (B, N, C)
            kv = F.linear(xs, Wkv, bkv)
            o = ops.attention_core(query, None, None, pos, self.rpe_table, heads=self.n_heads, groups=self.n_groups,
                                   views=1, precision=self.precision, kv=kv, attn_drop=drop)   # (B, H*W, C)
        out = ops.linear_rows(o, self.proj_out.weight.flatten(1), self.proj_out.bias)
        out = F.dropout(out, self.proj_drop_rate, self.training)                      # reference :336 (proj_drop)
        out = out.permute(0, 2, 1).reshape(B, C, H, W)
        return out, wandb_log_dict
