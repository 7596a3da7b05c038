"""Timing of the correlation head's kernels at the benchmark shape (B = 8: one (16 x 2.56 M) embedding matrix against itself,
normalised, forward + backward), HIP events around each call; GB/s on SURVEY 8d's bytes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevrender_amd import ops
torch.manual_seed(0)
n, E = 16, 64 * 200 * 200
emb = torch.randn(n, E, device="cuda", requires_grad=True)
for it in range(3):
    ops.KERNEL_TIMER.start()
    D = ops.pairwise_corr(emb, emb, normalize=True)
    D.square().sum().backward()
    r = ops.KERNEL_TIMER.stop()
print({k: (round(v["ms"] * 1e3, 1), "us", round(v["bytes"] / v["ms"] / 1e6, 0), "GB/s") for k, v in r.items()})
