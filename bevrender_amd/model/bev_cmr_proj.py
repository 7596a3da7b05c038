"""BEV pillar grid -> camera pixels on the GPU.

Counterpart of the reference's model/bev_cmr_proj.py (:12-59 constructor, :61-103 bev_grid_to_camera,
:105-124 in-bound mask): same class name, constructor arguments and return structure
({vehicle_code: [tensor (2, h, w, z) per camera]}, (x, y) normalised to [-1, 1], masked points pinned to
pixel (0, 0)).  The arithmetic runs in bevr_project_bev_grid (csrc/project.hip).  Unlike the reference the
caller's intrinsics are NOT rescaled in place (it mutates the config's arrays, :41-46); a rescaled copy is
kept instead.  The optional grey-pixel mask (`remove_ref_in_gray`, :114-122) reads one reference image per camera
from `bound_check_img_paths` (PIL, as the reference does) once, keeps them on the device and hands them to the
kernel, which drops every point whose truncated pixel is (128, 128, 128).
"""
import numpy as np
import torch

from .. import ops


class BEV2CameraProjector:
    def __init__(self, imu_to_rgb, K, vehicle_type_code, img_width, img_height, ori_img_width, ori_img_height,
                 remove_ref_in_gray=False, bound_check_img_paths=None, device="cuda", logger=None, use_wandb=False):
        self.remove_ref_in_gray = bool(remove_ref_in_gray)
        self.bound_check_img_paths = bound_check_img_paths
        self._gray_ref = None
        if self.remove_ref_in_gray and not bound_check_img_paths:
            raise ValueError("remove_ref_in_gray needs bound_check_img_paths (one image per camera)")
        self.scale_x = img_width / ori_img_width
        self.scale_y = img_height / ori_img_height
        self.img_width, self.img_height = img_width, img_height
        self.vehicle_type_code = vehicle_type_code
        self.device, self.logger, self.use_wandb = device, logger, use_wandb
        self.imu_to_cmr, self.K = {}, {}
        for code, mats in imu_to_rgb.items():
            self.imu_to_cmr[code] = [torch.tensor(np.asarray(m)).float() for m in mats]
        for code, mats in K.items():
            scaled = []
            for m in mats:
                k = np.array(m, dtype=np.float64, copy=True)
                k[0, 0] *= self.scale_x
                k[0, 2] *= self.scale_x
                k[1, 1] *= self.scale_y
                k[1, 2] *= self.scale_y
                scaled.append(torch.tensor(k).float())
            self.K[code] = scaled

    def vehicle_codes(self):
        return [self.vehicle_type_code]

    def gray_reference(self, device):
        """(ncam, C, H, W) uint8 reference images of the grey mask (the reference's F.pil_to_tensor(Image.open(path)))."""
        if self._gray_ref is None:
            from PIL import Image
            imgs = [torch.from_numpy(np.array(Image.open(p_))) for p_ in self.bound_check_img_paths]
            imgs = [(im[..., None] if im.dim() == 2 else im).permute(2, 0, 1) for im in imgs]
            self._gray_ref = torch.stack(imgs, 0).contiguous()
        return self._gray_ref.to(device)

    def bev_grid_to_camera(self, points_3d, device=None):
        device = torch.device(device if device is not None else self.device)
        _, h, w, z = points_3d.shape
        code = self.vehicle_type_code
        cam_inv = torch.stack([m.inverse() for m in self.imu_to_cmr[code]], 0)      # fp32 host inverse, as :72
        kmat = torch.stack([k[:, :3] for k in self.K[code]], 0)
        gray = self.gray_reference(device) if self.remove_ref_in_gray else None
        out = ops.project_bev_grid(points_3d.reshape(4, -1).to(device), cam_inv.to(device), kmat.to(device),
                                   self.img_width, self.img_height, gray)            # (ncam, 2, P)
        return {code: [out[c].reshape(2, h, w, z) for c in range(out.shape[0])]}
