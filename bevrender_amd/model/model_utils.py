"""Glue layers around the hot path (stock PyTorch ops -> MIOpen/rocBLAS).
Counterpart of the reference's model/model_utils.py:6-59 (same class and parameter names)."""
import torch
import torch.nn.functional as F
from torch import nn


def _dw_ok(x, conv: nn.Conv2d) -> bool:
    k = conv.kernel_size[0]
    return (conv.groups == conv.in_channels == conv.out_channels and conv.stride == (1, 1)
            and conv.kernel_size == (k, k) and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1)
            and k % 2 == 1 and k <= 5 and x.dtype == torch.float32)


def depthwise_conv2d(x, conv: nn.Conv2d):
    """A depthwise (groups == channels, one filter per channel) stride-1 'same' `nn.Conv2d` on an NCHW tensor through
    the HIP kernel (csrc/dwconv.hip): MIOpen runs the fp32 depthwise weight gradient through its naive reference
    kernel (59 ms per call at 200x200, C = 256).  Other configurations fall through to the module."""
    if not x.is_cuda or not _dw_ok(x, conv):   # glue layer: host tensors and odd configurations stay on the stock module
        return conv(x)
    from .. import ops
    return ops.depthwise_conv(x, conv.weight, conv.bias, nhwc=False)


def depthwise_conv2d_nhwc(x, conv: nn.Conv2d):
    """depthwise_conv2d for a channels-last (B, H, W, C) tensor (stride 1, odd kernel, 'same' padding)."""
    if not x.is_cuda or not _dw_ok(x, conv):
        return conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    from .. import ops
    return ops.depthwise_conv(x, conv.weight, conv.bias, nhwc=True)


def pointwise_conv_nhwc(x, conv: nn.Conv2d):
    """A 1x1 `nn.Conv2d` applied to a channels-last tensor as the GEMM it is (rocBLAS).  MIOpen routes the fp32
    weight gradient of these 1x1 convolutions through its naive reference kernel (61 ms per call at 200x200)."""
    from .. import ops
    return ops.linear_rows(x, conv.weight.flatten(1), conv.bias)


class LayerNormProxy(nn.Module):
    """LayerNorm over the channel axis of an NCHW tensor (parameter path: `.norm.weight/.bias`)."""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        xh = x.permute(0, 2, 3, 1)
        if x.is_cuda and x.dtype == torch.float32 and self.norm.elementwise_affine:
            from .. import ops
            if ops.layer_norm_supported(xh.shape[-1]):          # HIP: csrc/layernorm.hip
                return ops.layer_norm(xh, self.norm.weight, self.norm.bias, self.norm.eps).permute(0, 3, 1, 2)
        y = F.layer_norm(xh, self.norm.normalized_shape, self.norm.weight, self.norm.bias, self.norm.eps)
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
        from .. import ops
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none" and ops.dwconv_res_gelu_supported(y, self.dwc.weight) \
                and _dw_ok(y, self.dwc):
            y = ops.dwconv_res_gelu(y, self.dwc.weight, self.dwc.bias)      # act(y + dwc(y)) as one kernel
        else:
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


def attention_dropout(module):
    """(p, seed) for ops.attention_core(attn_drop=...) when `module` (an SCA / TSA attention module) is training with
    attn_drop_rate > 0, else None.  The seed comes from torch's CPU generator: reproducible under torch.manual_seed, no
    device synchronisation; the kernels turn (seed, problem, head, query, key) into the keep decision
    (ops.dropout_keep_mask), the same in the forward and the backward."""
    p = getattr(module, "attn_drop_rate", 0.0)
    if not (module.training and p > 0.0):
        return None
    return p, int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
