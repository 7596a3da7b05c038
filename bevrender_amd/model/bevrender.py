"""Top-level model with the reference's forward/API surface, so it drops into its train.py:
`BEVRender(config, logger, mode)` and
`forward(img_tensor, vehicle_pose_tensor, vehicle_type_tensor, wandb_log_dict, return_wandb_log=True)
 -> (output (B, 3, 224, 224), wandb_log_dict)`.
Counterpart of the reference's model/bevrender.py:14-221 (same config keys, parameter names and the same
eval()/train() toggling around the no-grad history pass).  The batch size is read from the input, so a
data-parallel shard of any size works; `config["PRECISION"]` ("f32" | "bf16", optional) selects the
kernels' operand type.
"""
import torch
import torch.nn as nn

from .bev_cmr_proj import BEV2CameraProjector
from .decoder_img_render import BEVImageRenderDecoder
from .encoder import BEVEncoder


class BEVRender(nn.Module):
    def __init__(self, config, logger, mode):
        super().__init__()
        self.logger = logger
        self.batch_size = config["BATCH_SIZE"] if mode == "train" else 1
        self.data_type = config["DATA_TYPE"]
        self.init_bev_height = self.init_bev_width = config["DAT_BEV_SHAPE"][0]
        self.init_embed_dim = config["DAT_EMBED_DIMS"][0]
        self.last_bev_height = self.last_bev_width = config["DAT_BEV_SHAPE"][-1]
        self.last_embed_dim = config["DAT_EMBED_DIMS"][-1]
        projector = BEV2CameraProjector(
            vehicle_type_code=config["VEHICLE_TYPE_CODE"], imu_to_rgb=config["IMU_TO_RGB"], K=config["INTRINSIC_K"],
            img_height=config["IMG_HEIGHT"], img_width=config["IMG_WIDTH"], ori_img_height=config["ORI_IMG_HEIGHT"],
            ori_img_width=config["ORI_IMG_WIDTH"], remove_ref_in_gray=config["REMOVE_REF_IN_GRAY"],
            bound_check_img_paths=config["BOUND_CHECK_IMG_PATH"], logger=logger)
        self.encoder = BEVEncoder(
            bev_bound=config["BEV_BOUND"], bev2cmr_projector=projector, batch_size=self.batch_size,
            scale_offset_range=config["DAT_SCALE_OFFSET_RANGE"], n_stages=config["DAT_NUM_STAGES"],
            n_views=config["NUM_VIEWS"], expansion=config["DAT_EXPANSION"], dims=config["DAT_EMBED_DIMS"],
            bev_feat_shapes=config["DAT_BEV_SHAPE"], bev_depth_dim=config["DAT_BEV_DEPTH_DIM"],
            z_shift=config["SAMPLE_Z_SHIFT"], depths=config["DAT_VIT_DEPTHS"], n_heads=config["DAT_NUM_HEADS"],
            strides=config["DAT_STRIDES"], n_groups=config["DAT_NUM_GROUPS"], kernel_size=config["DAT_K_SIZES"],
            drop_rate=config["DAT_DROP_RATE"], attn_drop_rate=config["DAT_ATTN_DROP_RATE"],
            drop_path_rate=config["DAT_DROP_PATH_RATE"], backbone_arch=config["DAT_BACKBONE_TYPE"],
            data_type=config["DATA_TYPE"], logger=logger, precision=config.get("PRECISION"),
            stage_dtype=config.get("STAGE_DTYPE"))
        self.decoder = BEVImageRenderDecoder(bev_spatial_dim=config["DAT_BEV_SHAPE"][-1],
                                             model_dim=config["DAT_EMBED_DIMS"][-1],
                                             hid_dim=config["DECODER_HID_DIM"], logger=logger)
        self.bev_embedding = nn.Embedding(self.init_bev_height * self.init_bev_width, self.init_embed_dim)
        self.init_weights()

    def forward(self, img_tensor, vehicle_pose_tensor, vehicle_type_tensor, wandb_log_dict, return_wandb_log=True):
        B = img_tensor.shape[0]
        q = self.bev_embedding.weight.to(self.data_type).to(img_tensor.device)
        q = q.t().reshape(1, self.init_embed_dim, self.init_bev_height, self.init_bev_width).expand(B, -1, -1, -1)
        vehicle_type_idx = vehicle_type_tensor[0, 0]
        self.eval()                                   # history frames: eval mode, no grad (reference :124-133)
        with torch.no_grad():
            prev_bev, wandb_log_dict = self.get_history_bev(q, img_tensor[:, :-1], vehicle_pose_tensor,
                                                            vehicle_type_idx, wandb_log_dict, False)
        self.train()                                  # unconditional, as the reference does (:134)
        bev = self.encoder(bev_query=q, img_tensor=img_tensor[:, -1], prev_bev=prev_bev,
                           vehicle_pose=vehicle_pose_tensor[:, -1], vehicle_type_idx=vehicle_type_idx,
                           wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log)
        return self.decoder(bev), wandb_log_dict

    def get_history_bev(self, bev_query, img_tensor, vehicle_pose, vehicle_type_idx, wandb_log_dict,
                        return_wandb_log=False):
        prev_bev = None
        assert img_tensor.shape[1] == vehicle_pose.shape[1] - 1
        n_hist = img_tensor.shape[1]
        # the backbones of all history frames as one batch when that is exact: eval mode (BatchNorm on running
        # statistics), which is how forward() always runs this pass; the recurrence over frames stays
        feats = self.encoder.history_features(img_tensor) if (n_hist > 1 and not self.training) else None
        for i in range(n_hist):
            prev_bev = self.encoder(bev_query, img_tensor[:, i], prev_bev, vehicle_pose[:, i:i + 2], vehicle_type_idx,
                                    wandb_log_dict=wandb_log_dict, return_wandb_log=return_wandb_log,
                                    img_feat=None if feats is None else feats[i])
        return prev_bev, wandb_log_dict

    def init_weights(self):
        """kaiming-normal convs, unit norms, xavier linears, U(0,1) embedding (reference :152-172)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.uniform_(m.weight)
