"""Image backbones (left to MIOpen: north_star keeps img_backbone on stock PyTorch).
Counterparts of the two live backbones of the reference, with its parameter names:
ResNet18_wo_fpn (model/img_backbone.py:429-454 over ResNet :165-286 / BasicBlock :95-162) and
PatchProjection (:457-501).  The reference's dead/broken ResnetFPN path is not reproduced."""
import torch.nn as nn

from .model_utils import LayerNormProxy


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1, is_first_block=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU()
        self.downsample = None
        if is_first_block and stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride, 0),
                                            nn.BatchNorm2d(out_channels))

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(y + skip)


class ResNet(nn.Module):
    def __init__(self, ResBlock, n_blocks_list=(3, 4, 6, 3), out_channels_list=(64, 128, 256, 512),
                 stride_list=(1, 1, 1, 1), num_channels=3):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(num_channels, 64, 3, 2, 1), nn.BatchNorm2d(64), nn.ReLU(),
                                   nn.MaxPool2d(3, 2, 1))
        chans = [64] + [c * ResBlock.expansion for c in out_channels_list]
        for idx in range(4):
            blocks = [ResBlock(chans[idx], out_channels_list[idx], stride_list[idx], True)]
            blocks += [ResBlock(chans[idx + 1], out_channels_list[idx]) for _ in range(n_blocks_list[idx] - 1)]
            setattr(self, f"conv{idx + 2}_x", nn.Sequential(*blocks))

    def forward(self, x):
        x = self.conv1(x)
        for idx in range(4):
            x = getattr(self, f"conv{idx + 2}_x")(x)
        return x


class ResNet18_wo_fpn(nn.Module):
    """Stride-4 (bev_dim 56) or stride-8 (bev_dim 28) single-scale 64-channel feature map."""

    def __init__(self, bev_dim, logger=None, use_wandb=False):
        super().__init__()
        self.logger, self.use_wandb = logger, use_wandb
        strides = {56: (1, 1, 1, 1), 28: (1, 2, 1, 1)}.get(bev_dim, (1, 1, 1, 1))
        self.resnet = ResNet(BasicBlock, (2, 2, 2, 2), (64, 64, 64, 64), strides)

    def forward(self, x):
        return self.resnet(x)


class PatchProjection(nn.Module):
    def __init__(self, embed_dim, patch_size, logger=None, use_wandb=False):
        super().__init__()
        self.logger, self.use_wandb = logger, use_wandb
        half = embed_dim // 2
        n_mid = {4: 0, 8: 1, 16: 2}[patch_size]
        layers = [nn.Conv2d(3, half, 3, 2, 1), LayerNormProxy(half), nn.GELU()]
        for _ in range(n_mid):
            layers += [nn.Conv2d(half, half, 3, 2, 1), LayerNormProxy(half), nn.GELU()]
        layers += [nn.Conv2d(half, embed_dim, 3, 2, 1), LayerNormProxy(embed_dim)]
        self.patch_projection = nn.Sequential(*layers)

    def forward(self, x):
        return self.patch_projection(x)
