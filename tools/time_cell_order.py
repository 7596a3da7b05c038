"""Timing of ops.cell_order (the per-call sort of the pinned keys by rpe-table cell) at the benchmark shape, 48 x 64 192 keys."""
import os
import torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevrender_amd import ops
torch.manual_seed(0)
a = (torch.rand(48, 64192, device="cuda") * 30 + 100)
b = (torch.rand(48, 64192, device="cuda") * 40 + 500)
for n_tail in (0, 128):
    o = ops.cell_order(a, b, n_tail)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        o = ops.cell_order(a, b, n_tail)
    e1.record(); torch.cuda.synchronize()
    print("n_tail", n_tail, "ms per call", e0.elapsed_time(e1) / 10, o.dtype, o.shape)
