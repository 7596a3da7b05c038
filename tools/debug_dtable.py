"""Debug: table gradient, bf16 mode vs f32 mode, tiny case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bevrender_amd import _lib, ops
torch.manual_seed(0)
B, C, h, S, D, N = 1, 64, 2, 8, 1, 64
g = torch.Generator().manual_seed(1)
query = torch.randn(B, C, S, S, generator=g)
k, v = torch.randn(B, N, C, generator=g), torch.randn(B, N, C, generator=g)
pos = (torch.rand(B, N, 2, generator=g) * 2 - 1)
table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=g) * 0.3
cot = torch.randn(B, S * S, C, generator=g)
res = {}
for prec in (_lib.PREC_F32, _lib.PREC_BF16):
    ins = [t.clone().cuda().requires_grad_(True) for t in (query, k, v, pos, table)]
    out = ops.attention_core(*ins, heads=h, groups=1, views=1, precision=prec)
    out.backward(cot.cuda())
    res[prec] = ins[4].grad.cpu()
a, b = res[0], res[1]
print("max f32", a.abs().max().item(), "max bf16", b.abs().max().item(), "sum f32", a.sum().item(), "sum bf16", b.sum().item())
print("f32 head0 rows 5..9:\n", a[0, 5:10, :8])
print("bf16 head0 rows 5..9:\n", b[0, 5:10, :8])
print("colsum f32", a[0].sum(0)[:8], "\ncolsum bf16", b[0].sum(0)[:8])
print("rowsum f32", a[0].sum(1), "\nrowsum bf16", b[0].sum(1))
