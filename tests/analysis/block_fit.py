"""How many 256-key blocks of the bench SCA geometry fit the bwd_k LDS ring (host-side estimate, static keys)."""
import os, sys, math, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import ring_rig
from oracle import bevrender_oracle as orc
from bevrender_amd import ops
S, D, V = 200, 5, 6
T, K = ring_rig(V, 704, 256)
pts = orc.sample_3d_points({"X": 50, "Y": 50, "Z": 2}, S, D, -1.0)
p2d = orc.bev_grid_to_camera(pts, T, K, 704, 256, 704, 256)
ref = orc.sca_reference_points(p2d, 1)[0]          # (V, h, w*d, 2) (x, y)
Wt = 2 * S * D - 1
Sp = 32 * ((S + 31) // 32)
for blk, wcap in ((384, 30720), (192, 14336), (192, 20480)):
    fit = tot = 0
    needs, rows = [], []
    for v in range(V):
        xy = ref[v].reshape(-1, 2).numpy().astype(np.float64)
        yx = xy[:, ::-1]
        yx = np.clip(yx, -3, 3)
        order = ops.kd_key_order(yx, S, Wt)
        a = ((1 - yx[:, 0]) * (S - 1) / 2)[order]
        b = ((1 - yx[:, 1]) * (Wt - 1) / 4)[order]
        a = np.clip(a, -(Sp + 1), 2 * S); b = np.clip(b, -(Wt // 2 + 2), Wt + 1)
        for k0 in range(0, len(a), blk):
            A = np.floor(a[k0:k0 + blk]); bb = b[k0:k0 + blk]
            r = int(A.max() - A.min()) + Sp + 1
            pitch = r | 1
            ncw = wcap // pitch - 1
            need = int(math.floor(bb.max() - bb.min())) + 4 + 6 + 1
            tot += 1; fit += ncw >= need
            needs.append(need); rows.append(r)
    print(f"blk {blk} wcap {wcap}: fit {fit}/{tot} = {fit/tot:.3f}; need cols p50 {np.percentile(needs,50)} p90 {np.percentile(needs,90)} max {max(needs)}; rows p50 {np.percentile(rows,50)} p90 {np.percentile(rows,90)}")
