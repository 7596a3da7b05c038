"""BEV encoder glue around the two hot-path attention blocks (stock PyTorch ops; MIOpen / rocBLAS).

Counterpart of the reference's model/encoder.py: BEVEncoder (:16-128), BEVEncoderStage (:131-234),
EncoderLayer (:237-466) with identical class names, constructor arguments, forward signatures and
parameter names (including the never-called down_proj / ffn_tsa / ffn_sca so state_dicts load).
Differences: the batch size is read from tensors; the ego-motion warp of the history BEV (eval mode, i.e. every
history frame after the first) is two batched HIP resamplings (ops.affine_warp, csrc/warp.hip) instead of a
per-sample torchvision loop (torchvision is absent here, so that step is restated from torchvision's documented
algorithm, fill-mask attenuation included: PARITY UNPINNED).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .SCA import SpatialCrossAttn
from .TSA import TemporalSelfAttn
from .feedforward import FeedForwardLayer
from .img_backbone import PatchProjection, ResNet18_wo_fpn
from .model_utils import LayerNormProxy, TransformerMLPWithConv, depthwise_conv2d


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


class EncoderLayer(nn.Module):
    def __init__(self, bev_bound, bev2cmr_projector, n_views, bev_feat_shape, bev_depth_dim, z_shift, dim_embed,
                 expansion, stage_idx, n_groups, n_heads, stride, kernel_size, batch_size, scale_offset_range,
                 attn_drop_rate=0.0, proj_drop_rate=0.0, mlp_drop_rate=0.0, drop_path_rate=0.2, ffn_drop_rate=0.1,
                 data_type=torch.float32, logger=None, precision=None):
        super().__init__()
        self.logger, self.stage_idx, self.bev_feat_shape = logger, stage_idx, bev_feat_shape
        C = dim_embed
        self.layer_scale = nn.Identity()
        self.layer_norm = LayerNormProxy(C)            # ONE norm shared by all four uses (reference :275)
        self.tsa_mlp = TransformerMLPWithConv(C, expansion, mlp_drop_rate)
        self.sca_mlp = TransformerMLPWithConv(C, expansion, mlp_drop_rate)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.tsa_local_percept_unit = nn.Conv2d(C, C, 3, 1, 1, groups=C)
        self.sca_local_percept_unit = nn.Conv2d(C, C, 3, 1, 1, groups=C)
        self.down_proj = nn.Sequential(nn.Conv2d(C, 2 * C, 3, 2, 1, bias=False), LayerNormProxy(2 * C))
        self.ffn_tsa = FeedForwardLayer(in_dim=bev_feat_shape, hidden_dim=C, dropout=ffn_drop_rate)
        self.ffn_sca = FeedForwardLayer(in_dim=bev_feat_shape, hidden_dim=C, dropout=ffn_drop_rate)
        common = dict(bev_feat_shape=bev_feat_shape, dim_embed=C, n_heads=n_heads, n_groups=n_groups, stride=stride,
                      kernel_size=kernel_size, batch_size=batch_size, scale_offset_range=scale_offset_range,
                      n_views=n_views, attn_drop_rate=attn_drop_rate, proj_drop_rate=proj_drop_rate,
                      data_type=data_type, logger=logger, precision=precision)
        self.temporal_self_attn = TemporalSelfAttn(**common)
        self.spatial_cross_attn = SpatialCrossAttn(bev_bound=bev_bound, bev2cmr_projector=bev2cmr_projector,
                                                   bev_depth_dim=bev_depth_dim, z_shift=z_shift, **common)

    def forward(self, bev_query, img_tensor, prev_bev, vehicle_pose, vehicle_type_idx, wandb_log_dict,
                return_wandb_log=True):
        x = bev_query
        if prev_bev is not None and not self.training:
            prev_bev = self.project_history_bev_feat(prev_bev, vehicle_pose)
        x = x + depthwise_conv2d(x, self.tsa_local_percept_unit)
        a, wandb_log_dict = self.temporal_self_attn(query=self.layer_norm(x), prev_bev=prev_bev,
                                                    wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log)
        x = self.drop_path(self.layer_scale(a)) + x
        x = self.drop_path(self.layer_scale(self.tsa_mlp(self.layer_norm(x)))) + x
        x = x + depthwise_conv2d(x, self.sca_local_percept_unit)
        a, wandb_log_dict = self.spatial_cross_attn(query=self.layer_norm(x), img_feat=img_tensor,
                                                    vehicle_type_idx=vehicle_type_idx, wandb_log_dict=wandb_log_dict,
                                                    return_wandb_log=return_wandb_log)
        x = self.drop_path(self.layer_scale(a)) + x
        x = self.drop_path(self.layer_scale(self.sca_mlp(self.layer_norm(x)))) + x
        return x, wandb_log_dict

    def project_history_bev_feat(self, bev, vehicle_pose, return_mask=False):
        """Warp the history BEV into the current frame: rotate by +prev_yaw and translate by (prev - cur)
        pixel offsets, then rotate by -cur_yaw (reference :413-466, two chained bilinear resamplings)."""
        from .. import ops
        pose = vehicle_pose.to(bev.device, torch.float32)
        prev_rot, curr_rot = pose[:, 0, 2], pose[:, 1, 2]
        delta = (pose[:, 0] - pose[:, 1])[:, :2]
        out = ops.affine_warp(bev, prev_rot, delta)                       # HIP: csrc/warp.hip, one launch per warp
        out = ops.affine_warp(out, -curr_rot, torch.zeros_like(delta))
        if return_mask:
            return out, out != 0
        return out


class BEVEncoderStage(nn.Module):
    def __init__(self, bev_bound, bev2cmr_projector, batch_size, scale_offset_range, stage_idx=0, n_views=3,
                 expansion=4, dims=(64, 128), bev_feat_shapes=(56, 28), bev_depth_dim=5, z_shift=-1.0, depth=2,
                 n_heads=2, strides=8, n_groups=1, kernel_size=9, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.2, data_type=torch.float32, logger=None, precision=None):
        super().__init__()
        self.logger = logger
        dims, shapes = list(dims), list(bev_feat_shapes)
        self.curr_feat_dim, self.next_feat_dim = dims if len(dims) == 2 else (dims[0], dims[0])
        self.curr_bev_feat_shape, self.next_bev_feat_shape = shapes if len(shapes) == 2 else (shapes[0], shapes[0])
        if self.curr_bev_feat_shape == self.next_bev_feat_shape:
            self.stage_project_conv = nn.Identity()
        elif self.curr_bev_feat_shape > self.next_bev_feat_shape:
            self.stage_project_conv = nn.Conv2d(self.curr_feat_dim, self.next_feat_dim, 3, 2, 1)
        else:
            self.stage_project_conv = nn.ConvTranspose2d(self.curr_feat_dim, self.next_feat_dim, kernel_size=2, stride=2)
        self.encoder_layers = nn.ModuleList([
            EncoderLayer(bev_bound=bev_bound, bev2cmr_projector=bev2cmr_projector, n_views=n_views,
                         bev_feat_shape=self.curr_bev_feat_shape, bev_depth_dim=bev_depth_dim, z_shift=z_shift,
                         dim_embed=self.curr_feat_dim, expansion=expansion, stage_idx=stage_idx, n_groups=n_groups,
                         n_heads=n_heads, stride=strides, kernel_size=kernel_size, batch_size=batch_size,
                         scale_offset_range=scale_offset_range, attn_drop_rate=attn_drop_rate,
                         proj_drop_rate=drop_rate, mlp_drop_rate=drop_rate, drop_path_rate=drop_path_rate,
                         data_type=data_type, logger=logger, precision=precision)
            for _ in range(depth)])

    def forward(self, bev_query, img_tensor, prev_bev, vehicle_pose, vehicle_type_idx, wandb_log_dict,
                return_wandb_log=True):
        for layer in self.encoder_layers:
            bev_query, wandb_log_dict = layer(bev_query=bev_query, img_tensor=img_tensor, prev_bev=prev_bev,
                                              vehicle_pose=vehicle_pose, vehicle_type_idx=vehicle_type_idx,
                                              wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log)
        return self.stage_project_conv(bev_query), wandb_log_dict


class BEVEncoder(nn.Module):
    def __init__(self, bev_bound, bev2cmr_projector, batch_size, scale_offset_range, n_stages=7, n_views=3,
                 expansion=4, dims=(64, 128, 256, 512, 256, 128, 64, 64), bev_feat_shapes=(56, 28, 14, 7, 14, 28, 56, 56),
                 bev_depth_dim=5, z_shift=-1.0, depths=(2,) * 7, n_heads=(2, 4, 8, 16, 8, 4, 2),
                 strides=(8, 4, 2, 1, 2, 4, 8), n_groups=(1, 2, 4, 8, 4, 2, 1), kernel_size=(9, 7, 5, 3, 5, 7, 9),
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2, backbone_arch="ResNet18",
                 data_type=torch.float32, logger=None, precision=None, stage_dtype=None):
        super().__init__()
        self.logger = logger
        # staging dtype of the camera images and of the backbone features they become.  None: whatever
        # the caller passes (the reference's behaviour); torch.bfloat16 / "bf16": the images are staged once, channels-last,
        # in bf16, and the backbone's features reach the sampler in bf16 -- the form the 16-bit attention modes read
        # without a float copy (`bevr_sample_*_bf16`, `bevr_kv_project`)
        if isinstance(stage_dtype, str):
            stage_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": None}[stage_dtype]
        self.stage_dtype = stage_dtype
        if backbone_arch == "ResNet18":
            self.img_backbone = ResNet18_wo_fpn(bev_dim=bev_feat_shapes[0], logger=logger)
        elif backbone_arch == "PatchProjection":
            self.img_backbone = PatchProjection(dims[0], {56: 4, 28: 8, 14: 16}[bev_feat_shapes[0]], logger=logger)
        elif backbone_arch in (None, "Identity"):      # features are handed in directly (benchmark path)
            self.img_backbone = nn.Identity()
        else:
            raise ValueError(f"unknown backbone {backbone_arch!r}")
        self.stages = nn.ModuleList([
            BEVEncoderStage(bev_bound=bev_bound, bev2cmr_projector=bev2cmr_projector, batch_size=batch_size,
                            scale_offset_range=scale_offset_range, stage_idx=i, n_views=n_views, expansion=expansion,
                            dims=dims[i:i + 2], bev_feat_shapes=bev_feat_shapes[i:i + 2], bev_depth_dim=bev_depth_dim,
                            z_shift=z_shift, depth=depths[i], n_heads=n_heads[i], strides=strides[i],
                            n_groups=n_groups[i], kernel_size=kernel_size[i], drop_rate=drop_rate,
                            attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, data_type=data_type,
                            logger=logger, precision=precision)
            for i in range(n_stages)])

    def backbone_features(self, img_tensor):
        """(B, V, 3, H, W) or (B*V, 3, H, W) camera images -> backbone features (B*V, C, Hf, Wf): the reference's
        `(b v) c h w` flatten + `img_backbone` (model/encoder.py:98-110)."""
        if img_tensor.dim() == 5:
            img_tensor = img_tensor.flatten(0, 1)                      # views into the batch
        # channels-last staging: MIOpen's NHWC convolutions, and features that arrive in the layout
        # the sampling kernel reads ((B*V, Hf, Wf, C) rows) -- ops.sample_features then takes them without a copy
        return self._staged_backbone(img_tensor)

    def _staged_backbone(self, x):
        """Backbone on channels-last images.  With `stage_dtype` set the images are STAGED in it (what a loader would hand
        over: half the bytes of the input side) and widened to the backbone's own dtype at its door -- the backbone's
        arithmetic is the config's, not autocast -- and the features leave in `stage_dtype`."""
        if self.stage_dtype is None:
            return self.img_backbone(x.contiguous(memory_format=torch.channels_last))
        x = x.to(self.stage_dtype).contiguous(memory_format=torch.channels_last)
        p = next(self.img_backbone.parameters(), None)
        feat = self.img_backbone(x if p is None else x.to(p.dtype))
        return feat.to(self.stage_dtype)

    def history_features(self, img_tensor):
        """(B, T', V, 3, H, W) -> list of T' feature tensors (B*V, C, Hf, Wf), all frames through the backbone as ONE
        batch.  The reference runs the backbone once per history frame inside its recurrent loop
        (model/bevrender.py:203-219); the history pass runs in eval mode (BatchNorm on running statistics), so the
        frames are independent and one launch set serves them all."""
        B, Tn, V = img_tensor.shape[:3]
        x = img_tensor.permute(1, 0, 2, 3, 4, 5).reshape(Tn * B * V, *img_tensor.shape[3:])   # frame-major
        feat = self._staged_backbone(x)
        return list(feat.reshape(Tn, B * V, *feat.shape[1:]).unbind(0))

    def forward(self, bev_query, img_tensor, prev_bev, vehicle_pose, vehicle_type_idx, wandb_log_dict,
                return_wandb_log=True, img_feat=None):
        """img_feat: backbone features computed ahead (history_features); `img_tensor` is then ignored."""
        feat = self.backbone_features(img_tensor) if img_feat is None else img_feat
        for stage in self.stages:
            bev_query, wandb_log_dict = stage(bev_query=bev_query, img_tensor=feat, prev_bev=prev_bev,
                                              vehicle_pose=vehicle_pose, vehicle_type_idx=vehicle_type_idx,
                                              wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log)
        return bev_query
