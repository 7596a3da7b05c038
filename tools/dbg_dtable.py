import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bevrender_amd import _lib, ops
from test_gpu_ops import _core_problem, _oracle_core
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
cfg = (1, 1, 64, 2, 1, 8, 1, 64)
B, V, C, h, g, S, D, N = cfg
query, k, v, pos, table = _core_problem(*cfg, seed=5, spread=0.9)
ins_cpu = [t.clone().requires_grad_(True) for t in (query, k, v, pos, table)]
want = _oracle_core(*ins_cpu, h, g, V)
cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
want.backward(cot)
for prec in (0, 1):
    ins = [t.clone().cuda().requires_grad_(True) for t in (query, k, v, pos, table)]
    got = ops.attention_core(*ins, heads=h, groups=g, views=V, precision=prec)
    got.backward(cot.cuda())
    e = (ins[4].grad.cpu() - ins_cpu[4].grad)
    print("prec", prec, "table grad max err", float(e.abs().max()), "max", float(ins_cpu[4].grad.abs().max()), "query err", float((ins[0].grad.cpu()-ins_cpu[0].grad).abs().max()))
    bad = (e[0].abs() > 1e-3 * ins_cpu[4].grad.abs().max()).nonzero()
    print("  n bad", len(bad), "of", e[0].numel(), " rows", sorted(set(bad[:, 0].tolist()))[:40], " cols", sorted(set(bad[:, 1].tolist()))[:40])
    if len(bad):
        y, x = bad[0].tolist()
        print("  first bad", y, x, "got", float(ins[4].grad[0, y, x]), "want", float(ins_cpu[4].grad[0, y, x]))
        print("  ratio sample", (ins[4].grad.cpu()[0] / (ins_cpu[4].grad[0] + 1e-12))[y:y+3, max(0,x-2):x+4])
