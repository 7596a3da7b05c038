"""Tap kernels (csrc/attn_tap*.hip) against a float64 restatement of their definition, and their timing at the benchmark's
shape.   python tools/tap_check.py check        small random cases, forward (+ backward when built)
         B=2 python tools/tap_check.py time     S=200, 65 984 keys per problem, B samples x 6 views x 2 heads"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevrender_amd import ops, _lib  # noqa: E402

dev = "cuda"


def hat(u):
    return (1 - u.abs()).clamp_min(0)


def tap_w(ys, xs, dtype=torch.float64):
    """(P, N) feature-pixel positions -> (P, N, 16) slot weights (slot 14 = 0: live, 15 = 1)."""
    ys, xs = ys.to(dtype), xs.to(dtype)
    w = torch.zeros(*ys.shape, 16, dtype=dtype, device=ys.device)
    for r in range(4):
        for c in range(3):
            w[..., r * 3 + c] = hat(r - ys) * hat(c - xs)
    w[..., 15] = 1
    return w


def bias_ref(T2, a, b, S, Wt, j, mimic=None):
    """T2 (h, Ht, Wt) float64, a, b (N,) -> bias (h, S rows i, N) for BEV column j: bilinear at (i + a, j rx + b), zeros outside."""
    h, Ht, _ = T2.shape
    rx = (Wt - 1) / (2.0 * (S - 1))
    i = torch.arange(S, device=a.device, dtype=torch.float64)
    ty = i[:, None] + a[None, :]
    tx = (torch.tensor(j * rx, dtype=torch.float32, device=a.device) + b.float()).double()[None, :].expand(S, -1)
    y0, x0 = torch.floor(ty), torch.floor(tx)
    fy, fx = ty - y0, tx - x0
    out = torch.zeros(h, S, a.numel(), dtype=torch.float64, device=a.device)
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = (y0 + dy).long(), (x0 + dx).long()
            wgt = (fy if dy else 1 - fy) * (fx if dx else 1 - fx)
            if mimic is not None:
                wgt = wgt.float().to(mimic).double()
            ok = (yy >= 0) & (yy < Ht) & (xx >= 0) & (xx < Wt)
            v = T2[:, yy.clamp(0, Ht - 1), xx.clamp(0, Wt - 1)]
            out += torch.where(ok, wgt, torch.zeros_like(wgt))[None] * v
    return out


def reference(G, Gb, a, b, ys, xs, T2, S, Wt, N, mimic=None):
    """float64 definition: returns Rn (P, h, S(j), S(i), 16) normalised, LSE (P, h, S, S).
    G (P, h, Mp, 16), Gb (P, h, Mp) as handed to the kernel (values), a, b, ys, xs (P, Np)."""
    P, h, Mp, _ = G.shape
    Sp = Mp // S
    Gd = G.double().reshape(P, h, S, Sp, 16)[:, :, :, :S]
    Gbd = Gb.double().reshape(P, h, S, Sp)[:, :, :, :S]
    w = tap_w(ys[:, :N], xs[:, :N])
    if mimic is not None:
        w = w.float().to(mimic).double()
        T2 = T2.float().to(mimic).double()
    Rn = torch.zeros(P, h, S, S, 16, dtype=torch.float64, device=G.device)
    LSE = torch.zeros(P, h, S, S, dtype=torch.float64, device=G.device)
    for p in range(P):
        for j in range(S):
            bia = bias_ref(T2, a[p, :N].double(), b[p, :N], S, Wt, j, mimic)                   # (h, S, N)
            lg = torch.einsum("hit,nt->hin", Gd[p, :, j, :, :12], w[p, :, :12]) + Gbd[p, :, j, :, None] + bia
            m = lg.amax(-1, keepdim=True)
            pr = torch.exp2(lg - m)
            l = pr.sum(-1, keepdim=True)
            LSE[p, :, j] = (m + torch.log2(l))[..., 0]
            Rn[p, :, j] = torch.einsum("hin,nt->hit", pr / l, w[p])
    return Rn, LSE


def grad_reference(G, Gb, H, Hc, a, b, ys, xs, T2, S, Wt, N, LSE):
    """float64 gradients through autograd of a pseudo-loss: with P, dS held constant, L = sum dS S + sum P (w . H) / ln2 has
    dL/dG = sum_n w dS, dL/dT2 = the table gradient, and dL/d(a, b, ys, xs) = the key-side gradients (the second term
    is the path through V_n = sum_t w_t Vpix_t).  Returns dG (P, h, S(j), S(i), 16), dT2, da, db, dys, dxs."""
    P, h, Mp, _ = G.shape
    Sp = Mp // S
    Gd = G.double().reshape(P, h, S, Sp, 16)[:, :, :, :S]
    Hd = H.double().reshape(P, h, S, Sp, 16)[:, :, :, :S]
    Gbd = Gb.double().reshape(P, h, S, Sp)[:, :, :, :S]
    Hcd = Hc.double().reshape(P, h, S, Sp)[:, :, :, :S]
    leaves = [t[:, :N].double().clone().requires_grad_(True) for t in (a, b, ys, xs)]
    al, bl, yl, xl = leaves
    T2l = T2.clone().requires_grad_(True)
    dG = torch.zeros(P, h, S, S, 16, dtype=torch.float64, device=G.device)
    for p in range(P):
        w = tap_w(yl[p], xl[p])
        for j in range(S):
            bia = bias_ref(T2l, al[p], bl[p], S, Wt, j)
            lg = torch.einsum("hit,nt->hin", Gd[p, :, j, :, :12], w[:, :12]) + Gbd[p, :, j, :, None] + bia
            Pm = torch.exp2(lg - LSE[p, :, j, :, None]).detach()
            dpl = torch.einsum("hit,nt->hin", Hd[p, :, j, :, :12], w[:, :12])
            dS = (Pm * (dpl + Hcd[p, :, j, :, None])).detach()
            dG[p, :, j] = torch.einsum("hin,nk->hik", dS, w.detach())
            ((dS * lg).sum() + (Pm * dpl).sum() * ops.LOG2E).backward(retain_graph=True)     # H carries ln2: the value path does not
    return dG, T2l.grad, al.grad, bl.grad, yl.grad, xl.grad


def set_offset(G, c):
    """slots 12, 13 of G <- hi, lo 16-bit parts of c; returns the value the kernels will see (hi + lo, float)."""
    hi = c.to(G.dtype)
    lo = (c - hi.float()).to(G.dtype)
    G[..., 12], G[..., 13] = hi, lo
    return hi.float() + lo.float()


def make_case(P, h, S, N, Wt, spread=(5.0, 2.5), gscale=4.0, seed=0, prec=_lib.PREC_BF16, sort=True, n_dead_logit=False):
    gen = torch.Generator(device=dev).manual_seed(seed)
    ed = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float16
    geom = ops.AttnGeom(n_prob=P, q_div=1, heads=h, groups=1, S=S, N=N, Wt=Wt, precision=prec)
    a = (S - 1) + (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * spread[0]
    b = (Wt - 1) / 2.0 + (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * spread[1]
    ys = (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * 1.6
    xs = (torch.rand(P, N, device=dev, generator=gen) * 2 - 1) * 0.45
    if sort:
        order = ops.cell_order(a, b)
        a, b, ys, xs = (t.gather(1, order) for t in (a, b, ys, xs))
    pad = geom.Np - N
    a, b, ys, xs = (torch.nn.functional.pad(t, (0, pad)).contiguous() for t in (a, b, ys, xs))
    G = torch.zeros(P, h, geom.Mp, 16, device=dev)
    G[..., :12] = torch.randn(P, h, geom.Mp, 12, device=dev, generator=gen) * gscale
    valid = (torch.arange(geom.Mp, device=dev) % geom.Sp) < S
    G = G * valid[None, None, :, None]
    G[..., 14] = -1.0e30 if prec == _lib.PREC_BF16 else -60000.0
    Gb = torch.randn(P, h, geom.Mp, device=dev, generator=gen) * valid
    T = torch.randn(h, 2 * S - 1, Wt, device=dev, generator=gen) * 0.5
    return geom, a, b, ys, xs, G.to(ed), Gb.contiguous(), T


def run_fwd(geom, a, b, ys, xs, G, Gb, T, headroom=64.0, timer=None):
    L = _lib.lib()
    d = geom.desc()
    Tt = ops.pack_table(T.float(), geom)
    pair = torch.stack((Tt[..., :-1], Tt[..., 1:]), dim=-1).contiguous()
    ws = torch.empty(L.bevr_attn_tap_ws_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
    _lib.check(L.bevr_attn_tap_prep(C.byref(d), ops._ptr(a), ops._ptr(b), ops._ptr(ys), ops._ptr(xs), ops._ptr(ws),
                                    ops._stream()), "tap_prep")
    tmax = (T.float() * ops.LOG2E).amax((1, 2)).clamp_min(0)                                 # per head
    U = G[..., :12].float().amax(-1).clamp_min(0) * 1.01 + Gb + tmax[None, :, None] * 1.01 + 0.01
    G = G.clone()
    mref = (Gb - set_offset(G, Gb - (U - headroom))).contiguous()      # the reference the kernel works against
    R = torch.empty(geom.n_prob, geom.heads, geom.Mp, 16, device=dev, dtype=torch.float32)
    flags = torch.zeros(geom.n_prob * geom.heads, geom.S, device=dev, dtype=torch.int32)
    args = (C.byref(d), ops._ptr(G), ops._ptr(ws), ops._ptr(pair), ops._ptr(mref), ops._ptr(R),
            ops._ptr(flags), ops._stream())
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.bevr_attn_tap_fwd(*args), "tap_fwd")
        e1.record()
        torch.cuda.synchronize()
        timer.append(e0.elapsed_time(e1))
    else:
        _lib.check(L.bevr_attn_tap_fwd(*args), "tap_fwd")
    return R, mref, flags, ws, pair


def run_bwd_q(geom, G, Gc, H, Hc, ws, pair, timer=None):
    L = _lib.lib()
    d = geom.desc()
    G, H = G.clone(), H.clone()
    set_offset(G, Gc)
    set_offset(H, Hc)
    dG = torch.empty(geom.n_prob, geom.heads, geom.Mp, 16, device=dev, dtype=torch.float32)
    dT = torch.zeros(geom.heads, geom.Wp, geom.Hp + 1, device=dev, dtype=torch.float32)
    args = (C.byref(d), ops._ptr(G), ops._ptr(H), ops._ptr(ws), ops._ptr(pair), ops._ptr(dG), ops._ptr(dT), ops._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.bevr_attn_tap_bwd_q(*args), "tap_bwd_q")
    e1.record()
    torch.cuda.synchronize()
    if timer is not None:
        timer.append(e0.elapsed_time(e1))
    return dG, dT


def run_bwd_k(geom, G, Gc, H, Hc, ws, Tt, timer=None):
    L = _lib.lib()
    d = geom.desc()
    G, H = G.clone(), H.clone()
    set_offset(G, Gc)
    set_offset(H, Hc)
    outs = [torch.zeros(geom.n_prob, geom.Np, device=dev, dtype=torch.float32) for _ in range(4)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.bevr_attn_tap_bwd_k(C.byref(d), ops._ptr(G), ops._ptr(H), ops._ptr(ws), ops._ptr(Tt), *[ops._ptr(t) for t in outs],
                                     ops._stream()), "tap_bwd_k")
    e1.record()
    torch.cuda.synchronize()
    if timer is not None:
        timer.append(e0.elapsed_time(e1))
    return outs


def relerr(got, want):
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def header_table(T, geom):
    """The `table` operand of bevr_attn_tap_bwd_k built from the words of include/bevrender_hip.h alone:
    table[h][x][y] = log2(e) * rpe_table[h][y - y_off][x - x_off], zero in the padding, shape [heads][Wp][Hp + 1]."""
    t = torch.zeros(geom.heads, geom.Wp, geom.Hp + 1, device=T.device, dtype=torch.float32)
    t[:, geom.x_off:geom.x_off + geom.Wt, geom.y_off:geom.y_off + geom.Ht] = (T.float() * ops.LOG2E).transpose(1, 2)
    return t.contiguous()


def check_case(name, **kw):
    headroom = kw.pop("headroom", 64.0)
    from_header = kw.pop("from_header", False)
    geom, a, b, ys, xs, G, Gb, T = make_case(**kw)
    R, mref, flags, _, _ = run_fwd(geom, a, b, ys, xs, G, Gb, T, headroom)
    torch.cuda.synchronize()
    S, Sp = geom.S, geom.Sp
    Rg = R.double().reshape(geom.n_prob, geom.heads, S, Sp, 16)[:, :, :, :S]
    mr = mref.double().reshape(geom.n_prob, geom.heads, S, Sp)[:, :, :, :S]
    l = Rg[..., 15]
    Rn_got = Rg / l[..., None]
    lse_got = mr + torch.log2(l)
    mim = torch.bfloat16 if geom.precision == _lib.PREC_BF16 else torch.float16
    T2 = T.double() * ops.LOG2E
    out = {}
    for tag, mm in (("exact", None), ("mimic", mim)):
        Rn, LSE = reference(G.float(), Gb, a, b, ys, xs, T2, S, geom.Wt, geom.N, mm)
        sel = [t for t in range(16) if t not in (12, 13)]
        out[tag] = ((Rn_got[..., sel] - Rn[..., sel]).abs().max().item(), (lse_got - LSE).abs().max().item())
    print(f"{name:28s} flagged {int(flags.sum())}  Rn err exact {out['exact'][0]:.2e} mimic {out['mimic'][0]:.2e}   "
          f"LSE err exact {out['exact'][1]:.2e} mimic {out['mimic'][1]:.2e}  dead-slot max {Rn_got[..., 14].abs().max().item():.1e}")
    # ---- backward, query side: H, Hc random; LSE of the float64 reference
    gen = torch.Generator(device=dev).manual_seed(99)
    ed = G.dtype
    valid = ((torch.arange(geom.Mp, device=dev) % Sp) < S)
    H = torch.zeros(geom.n_prob, geom.heads, geom.Mp, 16, device=dev)
    H[..., :12] = torch.randn(geom.n_prob, geom.heads, geom.Mp, 12, device=dev, generator=gen)
    H = (H * valid[None, None, :, None]).to(ed)
    Hc = (torch.randn(geom.n_prob, geom.heads, geom.Mp, device=dev, generator=gen) * valid).contiguous()
    _, LSE = reference(G.float(), Gb, a, b, ys, xs, T2, S, geom.Wt, geom.N, None)
    lse_p = torch.full((geom.n_prob, geom.heads, S, Sp), 1.0e30 if geom.precision == _lib.PREC_BF16 else 30000.0, device=dev, dtype=torch.float64)
    lse_p[:, :, :, :S] = LSE
    Gc = (Gb.double() - lse_p.reshape(geom.n_prob, geom.heads, geom.Mp)).float().contiguous()
    _, _, _, ws, pair = run_fwd(geom, a, b, ys, xs, G, Gb, T, headroom)
    dG, dT = run_bwd_q(geom, G, Gc, H, Hc, ws, pair)
    wdG, wdT, wda, wdb, wdy, wdx = grad_reference(G.float(), Gb, H.float(), Hc, a, b, ys, xs, T2, S, geom.Wt, geom.N, LSE)
    dGg = dG.double().reshape(geom.n_prob, geom.heads, S, Sp, 16)[:, :, :, :S]
    dTg = dT.double()[:, geom.x_off:geom.x_off + geom.Wt, geom.y_off:geom.y_off + geom.Ht].transpose(1, 2)
    leak = dT.double().abs().sum() - dTg.abs().sum()
    print(f"{'':28s} bwd_q: dG {relerr(dGg[..., :12], wdG[..., :12]):.2e}  dGb {relerr(dGg[..., 15], wdG[..., 15]):.2e}  "
          f"dtable {relerr(dTg, wdT):.2e} (outside the table {leak.item():.1e})")
    Tt = header_table(T, geom) if from_header else ops.pack_table(T.float(), geom).contiguous()
    da, db, dy, dx = [t.double()[:, :geom.N] for t in run_bwd_k(geom, G, Gc, H, Hc, ws, Tt)]
    print(f"{'':28s} bwd_k: da {relerr(da, wda):.2e}  db {relerr(db, wdb):.2e}  dys {relerr(dy, wdy):.2e}  dxs {relerr(dx, wdx):.2e}")
    out.update(flagged=int(flags.sum()), dG=relerr(dGg[..., :12], wdG[..., :12]), dGb=relerr(dGg[..., 15], wdG[..., 15]),
               dtable=relerr(dTg, wdT), da=relerr(da, wda), db=relerr(db, wdb), dys=relerr(dy, wdy), dxs=relerr(dx, wdx),
               dead=Rn_got[..., 14].abs().max().item())
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        check_case("sorted S=24 N=500", P=2, h=2, S=24, N=500, Wt=2 * 24 * 3 - 1)
        check_case("ragged S=20 N=333", P=3, h=2, S=20, N=333, Wt=2 * 20 * 5 - 1, seed=1)
        check_case("unsorted wide", P=1, h=2, S=18, N=200, Wt=2 * 18 * 5 - 1, spread=(12.0, 30.0), sort=False, seed=2)
        check_case("S=40 (3 blocks) N=1000", P=1, h=1, S=40, N=1000, Wt=2 * 40 * 5 - 1, seed=3)
        check_case("big logits (exact pass)", P=1, h=2, S=16, N=300, Wt=2 * 16 * 3 - 1, gscale=200.0, seed=4)
        check_case("fp16", P=1, h=2, S=24, N=400, Wt=2 * 24 * 3 - 1, seed=5, prec=_lib.PREC_F16, gscale=2.0, headroom=8.0)
    else:
        B = int(os.environ.get("B", "2"))
        S, h, D, V = 200, 2, 5, 6
        N = int(os.environ.get("N", "65984"))
        geom, a, b, ys, xs, G, Gb, T = make_case(P=B * V, h=h, S=S, N=N, Wt=2 * S * D - 1)
        ts = []
        for _ in range(int(os.environ.get("ITERS", "3"))):
            R, mref, flags, _, _ = run_fwd(geom, a, b, ys, xs, G, Gb, T, timer=ts)
        pairs = B * V * h * S * S * N
        print("TAP fwd ms", [round(t, 2) for t in ts], "Tpairs/s", round(pairs / min(ts) / 1e9, 2), "flagged", int(flags.sum()))
        _, _, _, ws, pair = run_fwd(geom, a, b, ys, xs, G, Gb, T)
        lse = (mref + torch.log2(R[..., 15].clamp_min(1e-37)))
        H = (torch.randn_like(G.float()) * (torch.arange(16, device=dev) < 12)).to(G.dtype)
        Hc = torch.randn_like(Gb)
        tq = []
        for _ in range(int(os.environ.get("ITERS", "3"))):
            run_bwd_q(geom, G, (Gb - lse).contiguous(), H, Hc, ws, pair, timer=tq)
        print("TAP bwd_q ms", [round(t, 2) for t in tq], "Tpairs/s", round(pairs / min(tq) / 1e9, 2))
        tk = []
        for _ in range(int(os.environ.get("ITERS", "3"))):
            run_bwd_k(geom, G, (Gb - lse).contiguous(), H, Hc, ws, ops.pack_table(T.float(), geom).contiguous(), timer=tk)
        print("TAP bwd_k ms", [round(t, 2) for t in tk], "Tpairs/s", round(pairs / min(tk) / 1e9, 2))


if __name__ == "__main__":
    main()
