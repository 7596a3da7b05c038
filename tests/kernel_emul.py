"""Torch (CPU, any float dtype) emulation of the ALGORITHM the HIP attention kernels implement, on the
kernels' packed layouts: integer-shift table coordinates, clamped key coordinates, zero-padded
transposed table.  Test-only: lets the packed formulation be checked against the oracle without a GPU
and localises GPU failures (packing vs kernel)."""
import torch


def emul_attn_fwd(Qp, Kp, Vp, key_a, key_b, Tt, geom):
    g = geom
    dt = Qp.dtype
    dev = Qp.device
    Mp, Np, Sp = g.Mp, g.Np, g.Sp
    mq = torch.arange(Mp, device=dev)
    i = (mq % Sp).to(dt)
    j = (mq // Sp).to(dt)
    rx = (g.Wt - 1) / (2.0 * (g.S - 1))
    O = torch.zeros(g.n_prob, g.heads, Mp, 32, dtype=dt, device=dev)
    LSE = torch.zeros(g.n_prob, g.heads, Mp, dtype=dt, device=dev)
    hpg = g.heads // g.groups
    kmask = torch.arange(Np, device=dev) >= g.N
    for prob in range(g.n_prob):
        for hd in range(g.heads):
            grp = hd // hpg
            a = key_a[prob * g.groups + grp].clamp(-(Sp + 1.0), g.Ht + 1.0)
            b = key_b[prob * g.groups + grp].clamp(-(g.Wt // 2 + 2.0), g.Wt + 1.0)
            A = torch.floor(a)
            fy = a - A
            tx = j[:, None] * rx + b[None, :]
            X = torch.floor(tx)
            fx = tx - X
            yi = (A[None, :] + i[:, None]).long() + g.y_off
            xi = X.long() + g.x_off
            T = Tt[hd]
            t00, t01 = T[xi, yi], T[xi, yi + 1]
            t10, t11 = T[xi + 1, yi], T[xi + 1, yi + 1]
            u0 = t00 * (1 - fy) + t01 * fy
            u1 = t10 * (1 - fy) + t11 * fy
            bias = u0 + fx * (u1 - u0)
            s = Qp[prob // g.q_div, hd] @ Kp[prob, hd].t() + bias
            s = s.masked_fill(kmask[None, :], float("-inf"))
            m = s.max(dim=1, keepdim=True).values
            p = torch.exp2(s - m)
            l = p.sum(1, keepdim=True)
            O[prob, hd] = (p / l) @ Vp[prob, hd]
            LSE[prob, hd] = (m + torch.log2(l)).squeeze(1)
    return O, LSE
