"""Spatial cross-attention wrapper: static pillar grid -> camera reference points, then the fused lift.

Counterpart of the reference's model/SCA.py (:8-58 constructor, :60-110 forward, :112-162
sample_3d_points).  The reference re-stacks and re-uploads the static reference points on every call
and syncs the host with `.item()`; here they are projected once per (module, device), cached on the
device and broadcast over the batch as a view.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .SCA_deform_attn import SCADeformableAttention


MIN_CELL_KEYS = int(os.environ.get("BEVR_CELL_MIN", "1024"))   # fewer pinned keys per view than this: not worth a second key segment


def pillar_grid(bev_bound, S: int, D: int, z_shift: float) -> torch.Tensor:
    """Homogeneous pillar-centre grid (4, S/2, S, D) float32 in the IMU frame.

    Bin centres: X: X/S*(2i+1), i < S/2 (forward half only);  Y: -Y + Y/S*(2j+1), j < S;
    Z: -Z + Z/D*(2d+1) + z_shift, d < D.   reference model/SCA.py:120-148.
    """
    X, Y, Z = float(bev_bound["X"]), float(bev_bound["Y"]), float(bev_bound["Z"])
    xs, ys, zs = X / S, Y / S, Z / D

    def bins(start, end, step, n):
        # same call as the reference (fp32 torch.arange over python-float bounds) so the centres are
        # bit-identical; float rounding can append one extra bin for some (bound, S) pairs:
        # the bin COUNT is fixed here.
        v = torch.arange(start, end, step)
        if v.numel() < n:
            raise ValueError("BEV bound / shape produce too few bins")
        return v[:n]

    gx = bins(0 + xs, X + xs, 2 * xs, S // 2)
    gy = bins(-Y + ys, Y + ys, 2 * ys, S)
    gz = bins(-Z + zs + z_shift, Z + zs + z_shift, 2 * zs, D)
    shape = (S // 2, S, D)
    return torch.stack((gx[:, None, None].expand(shape), gy[None, :, None].expand(shape),
                        gz[None, None, :].expand(shape), torch.ones(shape)), 0).contiguous()


class SpatialCrossAttn(nn.Module):
    def __init__(self, bev_bound, bev2cmr_projector, bev_feat_shape, bev_depth_dim, z_shift, dim_embed, n_heads,
                 n_groups, stride, kernel_size, batch_size, scale_offset_range, n_views=3, attn_drop_rate=0.0,
                 proj_drop_rate=0.0, data_type=torch.float32, logger=None, precision=None):
        super().__init__()
        self.bev_bound, self.bev_feat_shape, self.bev_depth_dim = bev_bound, bev_feat_shape, bev_depth_dim
        self.z_shift, self.logger, self.batch_size, self.num_views = z_shift, logger, batch_size, n_views
        self.projector = bev2cmr_projector
        self._ref_cache = {}
        assert n_heads % n_groups == 0, "n_heads must be divisible by n_groups"
        self.spatial_deform_attn = SCADeformableAttention(
            bev_feat_shape=bev_feat_shape, bev_depth_dim=bev_depth_dim, dim_embed=dim_embed, n_heads=n_heads,
            n_groups=n_groups, stride=stride, kernel_size=kernel_size, scale_offset_range=scale_offset_range,
            batch_size=batch_size, n_views=n_views, attn_drop_rate=attn_drop_rate, proj_drop_rate=proj_drop_rate,
            data_type=data_type, logger=logger, precision=precision)

    def sample_3d_points(self) -> torch.Tensor:
        return pillar_grid(self.bev_bound, self.bev_feat_shape, self.bev_depth_dim, self.z_shift)

    @property
    def points_2d_dict(self):
        """{vehicle_code: [ (2, S/2, S, D) per camera ]}, projected on the projector's device (lazy)."""
        return self.projector.bev_grid_to_camera(self.sample_3d_points())

    def reference_points(self, vehicle_code: int, device):
        """((V, S/2, S*D, 2) reference points in (x, y), (V, N) static key order, cell_split), cached per device.

        Key order of a view = [segment A | segment B]:
          B, keys [cell_split, N): pillar points the camera does not see.  The projector pins them all to pixel (0, 0)
             (reference model/bev_cmr_proj.py:76), so they differ only by their learned offsets and crowd ~70 cells of
             the rpe table; SCADeformableAttention sorts them by cell per call and runs them through the cell kernels
             (csrc/attn_cell.h).  Every view contributes the same number of them (the smallest pinned count over the
             views, rounded down to a multiple of 64), so that both segments have one length for all problems.
          A, keys [0, cell_split): the points the camera sees (+ the view's surplus pinned points), in the static k-d
             order of their projections (ops.kd_key_order), through the region kernels.
        Softmax attention is invariant to the order of its keys: the order and the split change no result."""
        key = (int(vehicle_code), str(device))
        if key not in self._ref_cache:
            pts = self.projector.bev_grid_to_camera(self.sample_3d_points(), device=device)[int(vehicle_code)]
            r = torch.stack(pts, 0)                                    # (V, 2, h, w, d)
            V, _, h, w, d = r.shape
            ref = r.permute(0, 2, 3, 4, 1).reshape(V, h, w * d, 2).contiguous()
            S, D = self.bev_feat_shape, self.bev_depth_dim
            yx = ref.reshape(V, -1, 2)[..., (1, 0)].double().cpu().numpy()
            order, split = ops.split_key_order(yx, S, 2 * S * D - 1, MIN_CELL_KEYS)
            self._ref_cache[key] = (ref, order.to(device), split)
        return self._ref_cache[key]

    def forward(self, query, img_feat, vehicle_type_idx, wandb_log_dict, return_wandb_log=True):
        B = query.shape[0]
        code = int(vehicle_type_idx) if not torch.is_tensor(vehicle_type_idx) else self._code(vehicle_type_idx)
        ref, order, split = self.reference_points(code, query.device)
        ref = ref[None].expand(B, -1, -1, -1, -1)
        if img_feat.dim() == 4:
            img_feat = img_feat.reshape(B, self.num_views, *img_feat.shape[1:])
        return self.spatial_deform_attn(x=img_feat, query=query, reference_points=ref,
                                        wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log,
                                        key_order=order, cell_split=split, split_is_pinned=True)

    def _code(self, t: torch.Tensor) -> int:
        # One rig per model in the reference (VEHICLE_TYPE_CODE); avoid its per-call .item() sync when possible.
        codes = self.projector.vehicle_codes()
        if len(codes) == 1:
            return codes[0]
        return int(t.item())
