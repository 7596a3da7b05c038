"""Glue layers around the hot path (stock PyTorch ops -> MIOpen/rocBLAS).
Counterpart of the reference's model/model_utils.py:6-59 (same class and parameter names)."""
import torch
import torch.nn.functional as F
from torch import nn


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
        y = self.drop1(self.linear1(x))
        y = self.act(y + self.dwc(y))
        return self.drop2(self.linear2(y))


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
