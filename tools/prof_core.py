"""Micro-workload for profiling the attention kernels alone (SCA-like keys: 65 % pinned to one pixel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevrender_amd import ops, _lib
torch.manual_seed(0)
B, V, C, h, S, D = 1, 4, 64, 2, 200, 5
N = (S // 2) * S * D
dev = 'cuda'
q = torch.randn(B, C, S, S, device=dev, requires_grad=True)
k = torch.randn(B * V, N, C, device=dev, requires_grad=True)
v = torch.randn(B * V, N, C, device=dev, requires_grad=True)
pos = torch.full((B * V, N, 2), -1.0, device=dev)
n_in = int(N * 0.35)
base = (torch.rand(B * V, n_in // 64 + 1, 1, 2, device=dev) * 2 - 1).expand(-1, -1, 64, -1).reshape(B * V, -1, 2)[:, :n_in]
pos[:, :n_in] = base + torch.randn(B * V, n_in, 2, device=dev) * torch.tensor([0.02, 0.003], device=dev)
pos = (pos + torch.randn_like(pos) * 0.005).requires_grad_(True)
table = (torch.randn(h, 2 * S - 1, 2 * S * D - 1, device=dev) * 0.1).requires_grad_(True)
for it in range(int(os.environ.get("ITERS", "2"))):
    ops.KERNEL_TIMER.start()
    out = ops.attention_core(q, k, v, pos, table, heads=h, groups=1, views=V, precision=_lib.PREC_BF16)
    out.square().mean().backward()
    r = ops.KERNEL_TIMER.stop()
print('TIMES', {k_: round(v_['ms'], 1) for k_, v_ in r.items()})
