"""Debug: which keys carry the d(pos) discrepancy at the cfg1 SCA geometry?  Compares the kernels (f32 mode) with the
oracle in float32 AND float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O
from test_gpu_fullsize import ring_rig

S, D, C, h = 50, 5, 64, 2
T, K = [ring_rig(1, 128, 128)[0][0]], [np.array([[100., 0, 64, 0], [0, 100., 64, 0], [0, 0, 1, 0]])]
gen = torch.Generator().manual_seed(50)
pts = O.sample_3d_points({"X": 20, "Y": 10, "Z": 2}, S, D, -1.0)
ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, 128, 128, 128, 128), 1)[0].reshape(1, -1, 2)[..., (1, 0)]
N = ref.shape[1]
rng = torch.tensor([1.0 / (S // 2 - 1.0), 1.0 / (S * D - 1.0)]) * 5.0
pos = (ref + torch.tanh(torch.randn(1, N, 2, generator=gen)) * rng)
order = torch.from_numpy(ops.kd_key_order(ref[0].double().numpy(), S, 2 * S * D - 1))
pos = pos[:, order].contiguous()
pinned = (ref[0, order] == -1).all(-1)
Wt = 2 * S * D - 1
query = torch.randn(1, C, S, S, generator=gen)
k, v = torch.randn(1, N, C, generator=gen), torch.randn(1, N, C, generator=gen)
table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * 0.3
cot = torch.randn(1, S * S, C, generator=gen)
c = C // h
res = {}
for dt in (torch.float32, torch.float64):
    cpu = [t.clone().to(dt).requires_grad_(True) for t in (query, k, v, pos, table)]
    o = O.attention_core(cpu[0][0].reshape(h, c, S * S), cpu[1][0].reshape(N, h, c).permute(1, 2, 0),
                         cpu[2][0].reshape(N, h, c).permute(1, 2, 0), cpu[3], cpu[4], S, S, 1, c ** -0.5)
    o.reshape(C, S * S).t().backward(cot[0].to(dt))
    res[dt] = cpu[3].grad[0].double()
gpu = [t.clone().cuda().requires_grad_(True) for t in (query, k, v, pos, table)]
got = ops.attention_core(*gpu, heads=h, groups=1, views=1, precision=_lib.PREC_F32)
got.backward(cot.cuda())
g = gpu[3].grad[0].double().cpu()
w32, w64 = res[torch.float32], res[torch.float64]
print("max|w64|", w64.abs().max().item())
print("kernel vs f64", (g - w64).abs().max().item(), " oracle32 vs f64", (w32 - w64).abs().max().item())
for comp, nm in ((0, "y"), (1, "x")):
    e = (g[:, comp] - w64[:, comp]).abs()
    idx = torch.argsort(e, descending=True)[:8]
    print("component", nm, "max|want|", w64[:, comp].abs().max().item())
    for i in idx.tolist():
        a = (1 - pos[0, i, 0].item()) * (S - 1) / 2
        b = (1 - pos[0, i, 1].item()) * (Wt - 1) / 4
        print(f"  key {i}: err {e[i].item():.3e} got {g[i, comp].item():+.4e} w64 {w64[i, comp].item():+.4e} w32 {w32[i, comp].item():+.4e}"
              f" pinned {bool(pinned[i])} a {a:.5f} b {b:.5f} pos {pos[0, i].tolist()}")
