"""Glue layers around the hot path (stock PyTorch ops -> MIOpen/rocBLAS).
Counterpart of the reference's model/model_utils.py:6-59 (same class and parameter names)."""
import torch
import torch.nn.functional as F
from torch import nn


def depthwise_conv2d(x, conv: nn.Conv2d):
    """A depthwise (groups == channels, one filter per channel) stride-1 `nn.Conv2d` evaluated as k*k shifted
    multiply-adds.  Same arithmetic as the module; MIOpen runs the fp32 depthwise weight gradient through its
    naive reference kernel (59 ms per call at 200x200, C = 256), while these are plain bandwidth-bound
    elementwise passes.  Other configurations fall through to the module."""
    k = conv.kernel_size[0]
    if (conv.groups != conv.in_channels or conv.out_channels != conv.in_channels or conv.stride != (1, 1)
            or conv.kernel_size != (k, k) or conv.padding != (k // 2, k // 2) or conv.dilation != (1, 1) or k % 2 == 0):
        return conv(x)
    H, W = x.shape[-2:]
    xp = F.pad(x, (k // 2,) * 4)
    w = conv.weight
    out = None
    for dy in range(k):
        for dx in range(k):
            term = xp[..., dy:dy + H, dx:dx + W] * w[:, 0, dy, dx].view(1, -1, 1, 1)
            out = term if out is None else out + term
    if conv.bias is not None:
        out = out + conv.bias.view(1, -1, 1, 1)
    return out


def depthwise_conv2d_nhwc(x, conv: nn.Conv2d):
    """depthwise_conv2d for a channels-last (B, H, W, C) tensor (stride 1, odd kernel, 'same' padding)."""
    k = conv.kernel_size[0]
    assert conv.groups == conv.in_channels == conv.out_channels and conv.stride == (1, 1) and k % 2 == 1
    H, W = x.shape[1:3]
    xp = F.pad(x, (0, 0, k // 2, k // 2, k // 2, k // 2))
    w = conv.weight
    out = None
    for dy in range(k):
        for dx in range(k):
            term = xp[:, dy:dy + H, dx:dx + W, :] * w[:, 0, dy, dx]
            out = term if out is None else out + term
    return out if conv.bias is None else out + conv.bias


def pointwise_conv_nhwc(x, conv: nn.Conv2d):
    """A 1x1 `nn.Conv2d` applied to a channels-last tensor as the GEMM it is (rocBLAS).  MIOpen routes the fp32
    weight gradient of these 1x1 convolutions through its naive reference kernel (61 ms per call at 200x200)."""
    return F.linear(x, conv.weight.flatten(1), conv.bias)


class LayerNormProxy(nn.Module):
    """LayerNorm over the channel axis of an NCHW tensor (parameter path: `.norm.weight/.bias`)."""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        y = F.layer_norm(x.permute(0, 2, 3, 1), self.norm.normalized_shape, self.norm.weight, self.norm.bias,
                         self.norm.eps)
        return y.permute(0, 3, 1, 2)


class TransformerMLPWithConv(nn.Module):
    """1x1 expand -> (+ depthwise 3x3) -> GELU -> 1x1 project."""

    def __init__(self, channels, expansion, drop):
        super().__init__()
        self.dim1, self.dim2 = channels, channels * expansion
        self.linear1 = nn.Sequential(nn.Conv2d(self.dim1, self.dim2, 1, 1, 0))
        self.drop1 = nn.Dropout(drop)
        self.act = nn.GELU()
        self.linear2 = nn.Sequential(nn.Conv2d(self.dim2, self.dim1, 1, 1, 0))
        self.drop2 = nn.Dropout(drop)
        self.dwc = nn.Conv2d(self.dim2, self.dim2, 3, 1, 1, groups=self.dim2)

    def forward(self, x):
        xh = x.permute(0, 2, 3, 1)                                  # channels-last view (a no-copy view of the
        y = self.drop1(pointwise_conv_nhwc(xh, self.linear1[0]))    # LayerNormProxy output that feeds this block)
        y = self.act(y + depthwise_conv2d_nhwc(y, self.dwc))
        y = self.drop2(pointwise_conv_nhwc(y, self.linear2[0]))
        return y.permute(0, 3, 1, 2)


class LayerScale(nn.Module):
    def __init__(self, dim: int, inplace: bool = False, init_values: float = 1e-5):
        super().__init__()
        self.inplace = inplace
        self.weight = nn.Parameter(torch.full((dim,), init_values))

    def forward(self, x):
        w = self.weight.view(-1, 1, 1)
        return x.mul_(w) if self.inplace else x * w


def normalized_grid(H: int, W: int, dtype, device) -> torch.Tensor:
    """(H, W, 2) in (y, x), each axis k/(n-1)*2-1 (model/TSA_deform_attn.py:98-109)."""
    gy = torch.arange(H, dtype=dtype, device=device) / (H - 1.0) * 2.0 - 1.0
    gx = torch.arange(W, dtype=dtype, device=device) / (W - 1.0) * 2.0 - 1.0
    return torch.stack(torch.meshgrid(gy, gx, indexing="ij"), -1)


def trunc_normal_(t: torch.Tensor, std: float = 1.0):
    return nn.init.trunc_normal_(t, std=std)
