"""Fraction of 64-key steps (and of their 32-key halves) whose tap box fits the query-stationary LDS windows (static keys)."""
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
ref = orc.sca_reference_points(p2d, 1)[0]
Wt = 2 * S * D - 1
rx = (Wt - 1) / (2.0 * (S - 1))
for leaf in (64, 32):
    for ncol, cap in ((8, 64), (8, 88), (8, 56)):
        tot = fit64 = fit32 = fit_either = 0
        fr, fc = [], []
        for v in range(V):
            xy = ref[v].reshape(-1, 2).numpy().astype(np.float64)
            yx = np.clip(xy[:, ::-1], -3, 3)
            order = ops.kd_key_order(yx, S, Wt, leaf=leaf)
            a = ((1 - yx[:, 0]) * (S - 1) / 2)[order]
            b = ((1 - yx[:, 1]) * (Wt - 1) / 4)[order]
            def ok(A, bb):
                rows = 32 + int(A.max() - A.min())
                cols = int(math.floor((ncol - 1) * rx + bb.max())) + 2 - (int(math.floor(bb.min())) - 1) + 1
                return rows <= 63 and cols <= cap
            for k0 in range(0, len(a) - 63, 64):
                A = np.floor(a[k0:k0 + 64]); bb = b[k0:k0 + 64]
                tot += 1
                f = ok(A, bb)
                if not f:
                    fr.append(32 + int(A.max() - A.min())); fc.append(float(bb.max() - bb.min()))
                fit64 += f
                h0, h1 = ok(A[:32], bb[:32]), ok(A[32:], bb[32:])
                fit32 += (h0 + h1) / 2
        print(f"leaf {leaf} ncol {ncol} cap {cap}: 64-key steps fit {fit64/tot:.4f}; 32-key halves fit {fit32/tot:.4f}; failing: rows p50 {np.percentile(fr,50)} p90 {np.percentile(fr,90)} bspread p50 {np.percentile(fc,50):.0f} p90 {np.percentile(fc,90):.0f}")
