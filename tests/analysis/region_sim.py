"""CPU simulation of the query-stationary kernels' table-region logic at the benchmark geometry (cfg2: ring rig,
S=200, D=5, V=6, keys = static projections + offsets over the full learned range, k-d order), and the histogram of
32-key leaf boxes VERDICT r01 item 3 asks for (is the bias a small-K GEMM?).

  python tests/analysis/region_sim.py            (a minute on 8 cores)
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fullsize import lift_problem  # noqa: E402

S, D, V = 200, 5, 6
Wt = 2 * S * D - 1
rx = (Wt - 1) / (2.0 * (S - 1))
p = lift_problem(S, D, V, 64, 2, 704, 256, {"X": 50, "Y": 50, "Z": 2}, seed=2024)
pos = p["pos"].double().numpy()
a_all = (1 - pos[..., 0]) * (S - 1) / 2
b_all = (1 - pos[..., 1]) * (Wt - 1) / 4
N = a_all.shape[1]
print(f"pinned share {p['pinned']:.3f}, N {N}")

# ---- leaf boxes: K = rows x cols of the taps of one 32-key leaf for ONE query column ---------------------------
Ks = []
for v in range(V):
    A = np.floor(a_all[v]).astype(int)
    for k0 in range(0, N, 32):
        Ah, bh = A[k0:k0 + 32], b_all[v, k0:k0 + 32]
        R = Ah.max() - Ah.min() + 2
        Cw = int(math.floor(bh.max()) - math.floor(bh.min())) + 2 + 1      # +1: frac(j rx) shifts the box by < 1
        Ks.append(R * Cw)
Ks = np.array(Ks)
qs = [50, 75, 90, 95, 99]
print("leaf-box K = rows x cols:", {q: int(np.percentile(Ks, q)) for q in qs},
      " share with K<=64: %.3f, K<=96: %.3f, K<=128: %.3f" % ((Ks <= 64).mean(), (Ks <= 96).mean(), (Ks <= 128).mean()))


# ---- region walk (attn_tile.h semantics) --------------------------------------------------------------------------
def walk(cap, ncol=8, rows_max=63):
    tot_half = win_half = moves = steps_whole = 0
    for v in range(V):
        A = np.floor(a_all[v]).astype(int)
        b = b_all[v]
        nstep = N // 64
        amin = A[:nstep * 64].reshape(nstep, 2, 32).min(2)
        amax = A[:nstep * 64].reshape(nstep, 2, 32).max(2)
        bmin = b[:nstep * 64].reshape(nstep, 2, 32).min(2)
        bmax = b[:nstep * 64].reshape(nstep, 2, 32).max(2)
        for cb in range(0, S // ncol, 6):                       # every 6th column block: the statistics are smooth
            jlo, jhi = cb * ncol * rx, (cb * ncol + ncol - 1) * rx
            rg = None                                            # (ax0, ay0)

            def info(lo_a, hi_a, lo_b, hi_b):
                nrows = 32 + hi_a - lo_a
                xlo = int(math.floor(jlo + lo_b)) - 1
                xhi = int(math.floor(jhi + hi_b)) + 2
                return lo_a, nrows, xlo, xhi - xlo + 1, (nrows <= rows_max and xhi - xlo + 1 <= cap)

            def contains(rg, wi):
                return wi[2] >= rg[0] and wi[2] + wi[3] <= rg[0] + cap and wi[0] >= rg[1] and wi[0] + wi[1] <= rg[1] + rows_max

            def anchor(wi):
                return (wi[2] - (cap - wi[3]) // 2, wi[0] - (rows_max - wi[1]) // 2)

            for s in range(nstep):
                whole = info(amin[s].min(), amax[s].max(), bmin[s].min(), bmax[s].max())
                if whole[4]:
                    steps_whole += 1
                    tot_half += 2
                    win_half += 2
                    if rg is None or not contains(rg, whole):
                        rg = anchor(whole)
                        moves += 1
                    continue
                for hf in range(2):
                    wi = info(amin[s, hf], amax[s, hf], bmin[s, hf], bmax[s, hf])
                    tot_half += 1
                    if wi[4]:
                        win_half += 1
                        if rg is None or not contains(rg, wi):
                            rg = anchor(wi)
                            moves += 1
    return win_half / tot_half, moves / (tot_half / 2), steps_whole / (tot_half / 2)


for cap in (40, 48, 52, 56, 64, 88):
    f, m, w = walk(cap)
    print(f"CAP {cap:3d} columns: halves served from the LDS region {f:.4f}; region moves per step {m:.3f}; whole-step windows {w:.3f}")


# ---- look-ahead anchoring: on a move, centre the region on the union of the longest run of following boxes that
# still fits (instead of on the box that forced the move) -------------------------------------------------------------
def walk_lookahead(cap, ncol=8, rows_max=63, max_run=64):
    tot = moves = 0
    for v in range(V):
        A = np.floor(a_all[v]).astype(int)
        b = b_all[v]
        nh = N // 32
        amin = A[:nh * 32].reshape(nh, 32).min(1); amax = A[:nh * 32].reshape(nh, 32).max(1)
        bmin = b[:nh * 32].reshape(nh, 32).min(1); bmax = b[:nh * 32].reshape(nh, 32).max(1)
        wb = cap - (ncol - 1) * rx - 5.5
        for cb in range(0, S // ncol, 6):
            jlo, jhi = cb * ncol * rx, (cb * ncol + ncol - 1) * rx
            rg = None

            def info(lo_a, hi_a, lo_b, hi_b):
                nrows = 32 + hi_a - lo_a
                xlo = int(math.floor(jlo + lo_b)) - 1
                xhi = int(math.floor(jhi + hi_b)) + 2
                return lo_a, nrows, xlo, xhi - xlo + 1, (nrows <= rows_max and xhi - xlo + 1 <= cap)

            for s in range(nh):
                wi = info(amin[s], amax[s], bmin[s], bmax[s])
                if not wi[4]:
                    continue                                  # grouped / fallback: not modelled here
                tot += 1
                if rg is not None and wi[2] >= rg[0] and wi[2] + wi[3] <= rg[0] + cap and wi[0] >= rg[1] and wi[0] + wi[1] <= rg[1] + rows_max:
                    continue
                moves += 1
                la, ha, lb, hb = amin[s], amax[s], bmin[s], bmax[s]
                for t in range(s + 1, min(nh, s + max_run)):
                    na, xa, nb, xb = min(la, amin[t]), max(ha, amax[t]), min(lb, bmin[t]), max(hb, bmax[t])
                    if xa - na > 31 or xb - nb > wb:
                        break
                    la, ha, lb, hb = na, xa, nb, xb
                u = info(la, ha, lb, hb)
                rg = (u[2] - (cap - u[3]) // 2, u[0] - (rows_max - u[1]) // 2)
    return moves / (tot / 2)


for cap in (56, 64, 88):
    print(f"CAP {cap:3d} look-ahead anchoring: region moves per step {walk_lookahead(cap):.3f}")
