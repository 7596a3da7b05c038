"""Cell kernels alone at the benchmark's shape: S=200, 66 000 cell-sorted keys per problem (the projector's pinned keys
with offsets over the learned range), B samples x 6 views x 2 heads, bf16.  Per-kernel times from HIP events.
  B=2 ITERS=3 python tools/prof_cell.py          (BEVRENDER_LIB=... selects a variant build)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevrender_amd import ops, _lib
torch.manual_seed(0)
S, C, h, D, V, B = 200, 64, 2, 5, 6, int(os.environ.get("B", "2"))
N = int(os.environ.get("N", "65984"))
dev = "cuda"
Wt = 2 * S * D - 1
gen = torch.Generator(device=dev).manual_seed(1)
P = B * V
a = (S - 1) + (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * 5.0
b = (Wt - 1) / 2.0 + (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * 2.5
order = ops.cell_order(a, b)
a, b = a.gather(1, order), b.gather(1, order)
pos = torch.stack((1 - a * 2 / (S - 1), 1 - b * 4 / (Wt - 1)), -1).requires_grad_(True)
q = torch.randn(B, C, S, S, device=dev, generator=gen, requires_grad=True)
k = torch.randn(P, N, C, device=dev, generator=gen, requires_grad=True)
v = torch.randn(P, N, C, device=dev, generator=gen, requires_grad=True)
table = (torch.randn(h, 2 * S - 1, Wt, device=dev, generator=gen) * 0.3).requires_grad_(True)
prec = {"bf16": _lib.PREC_BF16, "f16": _lib.PREC_F16, "bf16x3": _lib.PREC_BF16X3, "f32": _lib.PREC_F32}[os.environ.get("PREC", "bf16")]
for it in range(int(os.environ.get("ITERS", "3"))):
    ops.KERNEL_TIMER.start()
    out = ops.attention_core(q, k, v, pos, table, heads=h, groups=1, views=V, precision=prec, cell_split=0)
    out.square().mean().backward()
    r = ops.KERNEL_TIMER.stop()
pairs = P * h * S * S * N
print("TIMES", os.environ.get("BEVRENDER_LIB", "default").split("/")[-2:], {k_: round(v_["ms"], 2) for k_, v_ in r.items()},
      "Gpairs/s", {k_: round(pairs / v_["ms"] / 1e6, 0) for k_, v_ in r.items()})
