"""Parity at the sizes the benchmark runs (BASELINE.json configs 1-3 and the S=400 geometry of config 5).

The reference materialises (M x N) tensors: 4e9 entries per head at S = 200, so nothing of that size can be
compared whole.  Attention rows are independent of each other, so the oracle is evaluated on a random SUBSET of the
query rows (oracle.attention_core(rows=...), the reference's materialised arithmetic restricted to those rows)
and the kernels' full output is compared on those rows; the cotangent is zero outside the subset, so dK, dV,
d(pos) and d(rpe_table) of the kernels' full backward equal the oracle's.  Geometry is the benchmark's: ring rig of
6 cameras at 256x704, 200x200 BEV, D = 5 height bins, keys in the static k-d order, offsets over the full learned
range (tanh * 5 / (Hk - 1)).  Needs a real MI355X.
"""
import math
import os

import numpy as np
import pytest
import torch

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


def ring_rig(V, img_w, img_h):
    """bench.py's rig: V cameras on a ring (yaw 360 v / V, pitch 0, 1.5 m up), fx = fy = 0.8 W."""
    R0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    T, K = [], []
    for v in range(V):
        a = 2 * math.pi * v / V
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        M = np.eye(4)
        M[:3, :3] = Rz @ R0
        M[:3, 3] = (0, 0, 1.5)
        T.append(M)
        K.append(np.array([[0.8 * img_w, 0, img_w / 2, 0], [0, 0.8 * img_w, img_h / 2, 0], [0, 0, 1, 0]]))
    return T, K


def lift_problem(S, D, V, C, h, img_w, img_h, bound, seed, table_std=0.3):
    """Query, projected keys/values, key positions (static projection + learned-range offsets, k-d ordered) and
    table of one sample's SCA attention at the benchmark's geometry.  CPU tensors."""
    gen = torch.Generator().manual_seed(seed)
    T, K = ring_rig(V, img_w, img_h)
    pts = O.sample_3d_points(bound, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, img_w, img_h, img_w, img_h), 1)[0]   # (V,Hk,Wk,2) xy
    Hk, Wk = S // 2, S * D
    N = Hk * Wk
    ref = ref.reshape(V, N, 2)[..., (1, 0)]                                   # (y, x)
    pinned = (ref == -1.0).all(-1).float().mean().item()
    rng = torch.tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)]) * 5.0
    pos = ref + torch.tanh(torch.randn(V, N, 2, generator=gen)) * rng
    Wt = 2 * S * D - 1
    order = torch.stack([torch.from_numpy(ops.kd_key_order(ref[v].double().numpy(), S, Wt)) for v in range(V)], 0)
    pos = torch.gather(pos, 1, order[..., None].expand(-1, -1, 2)).contiguous()
    ref_o = torch.gather(ref, 1, order[..., None].expand(-1, -1, 2))
    pinned_mask = (ref_o == -1.0).all(-1)                                     # (V, N), in the returned key order
    query = torch.randn(1, C, S, S, generator=gen)
    k = torch.randn(V, N, C, generator=gen)
    v = torch.randn(V, N, C, generator=gen)
    table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * table_std
    return dict(query=query, k=k, v=v, pos=pos, table=table, pinned=pinned, pinned_mask=pinned_mask)


def cell_split_perm(p, S):
    """What the SCA module does with a view's keys, as a permutation of lift_problem's (k-d ordered) keys: the
    projector's pinned keys go to the end, [split, N) -- every view the same number, a multiple of 64 -- sorted by
    rpe-table cell of their CURRENT positions (ops.cell_order); the other keys keep their k-d order.
    Returns (perm (V, N) long, split)."""
    pm = p["pinned_mask"]
    V, N = pm.shape
    Wt = p["table"].shape[-1]
    n_b = (int(pm.sum(1).min()) // 64) * 64
    perms = []
    for v in range(V):
        ip = torch.nonzero(pm[v]).flatten()
        seg_b = ip[len(ip) - n_b:]
        keep = torch.ones(N, dtype=torch.bool)
        keep[seg_b] = False
        seg_a = torch.nonzero(keep).flatten()
        a, b = ops.key_coords(p["pos"][v:v + 1, seg_b], S, Wt, n_b)
        perms.append(torch.cat((seg_a, seg_b[ops.cell_order(a, b)[0]])))
    return torch.stack(perms, 0), N - n_b


def permute_keys(t, perm):
    """(V, N, ...) gathered along the key axis by perm (V, N)."""
    idx = perm.reshape(perm.shape + (1,) * (t.dim() - 2)).expand(-1, -1, *t.shape[2:])
    return torch.gather(t, 1, idx)


def oracle_rows(p, h, rows, cot, want_grads=True, dtype=torch.float64):
    """oracle.attention_core on the query subset `rows`, one view at a time (each view is an independent softmax
    problem sharing the query).  Returns out (V, R, C) and the gradients for the cotangent `cot` (V, R, C).
    float64 by default: d(pos) is discontinuous where a table coordinate crosses an integer, and a float32 oracle
    lands on either side of such a kink by its own rounding (see check_dpos)."""
    query, k, v, pos, table = (p[n].clone().to(dtype).requires_grad_(want_grads)
                               for n in ("query", "k", "v", "pos", "table"))
    if cot is not None:
        cot = cot.to(dtype)
    V, N, C = k.shape
    S = query.shape[-1]
    c = C // h
    outs = []
    for vi in range(V):
        q = query[0].reshape(h, c, S * S)
        kk = k[vi].reshape(N, h, c).permute(1, 2, 0)
        vv = v[vi].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[vi:vi + 1], table, S, S, 1, c ** -0.5, rows=rows)     # (h, c, R)
        o = o.reshape(C, len(rows)).t()
        if want_grads:
            (o * cot[vi]).sum().backward()
        outs.append(o.detach())
    grads = None
    if want_grads:
        grads = dict(query=query.grad, k=k.grad, v=v.grad, pos=pos.grad, table=table.grad)
    return torch.stack(outs, 0), grads


def pick_rows(S, R, seed):
    """R query indices m = i*S + j: random, plus the grid corners and the last (ragged) 32-row block."""
    g = torch.Generator().manual_seed(seed)
    rows = torch.randperm(S * S, generator=g)[:R - 8]
    extra = torch.tensor([0, S - 1, (S - 1) * S, S * S - 1, (S - 3) * S + 7, 31 * S + 5, 32 * S + 5, (S // 2) * S])
    return torch.unique(torch.cat((rows, extra)))


def rel_err(got, want):
    return (got.double() - want.double()).abs().max().item() / (want.abs().max().item() + 1e-30)


def kink_distance(pos, S, Wt, cols=None):
    """Per key: how close a table coordinate of the key comes to an integer on the query grid.  The bias is
    piecewise bilinear in (ty, tx) = (i + a_n, j rx + b_n): continuous, but its derivative with respect to the key
    position jumps where ty or tx crosses an integer.  frac(ty) = frac(a_n) for every row i; tx is tested per column.
    cols: the BEV columns j that carry cotangent (default: all).  A column without cotangent contributes nothing to
    d(pos), so its kinks do not matter: with the cotangent on a few sampled rows only those rows' columns are tested.
    (At S = 400, j rx mod 1 sweeps the unit interval four times over the grid, so EVERY key is within 2e-3 of a kink of
    SOME column and the unrestricted test left no key to check: VERDICT r02.)"""
    pos = pos.double()
    a = (1 - pos[..., 0]) * (S - 1) / 2
    b = (1 - pos[..., 1]) * (Wt - 1) / 4
    rx = (Wt - 1) / (2.0 * (S - 1))
    dist = (a - a.round()).abs()
    for j in (range(S) if cols is None else sorted(set(int(c) for c in cols))):
        t = b + j * rx
        dist = torch.minimum(dist, (t - t.round()).abs())
    return dist


def check_dpos(got, want, pos, S, Wt, lim, tag, cols=None, min_clean=0.0):
    """d(pos) against the float64 oracle.  Keys whose table coordinate passes within float32 rounding of an integer
    for some query column take the derivative of the neighbouring bilinear cell for that column (in this kernel as in
    the float32 reference: at cfg1 the reference's own float32 and float64 evaluations differ by 6 % of the largest
    entry on 4 of 6 250 keys, tools/debug_dpos.py).  So: (1) every key OUTSIDE the kink neighbourhood must meet the
    limit; (2) keys that miss it must be few and must all be kink-adjacent; (3) the error in the 2-norm is small."""
    got, want = got.double().cpu(), want.double()
    err = (got - want).abs().amax(-1)
    tol = lim * want.abs().max().item()
    bad = err > tol
    kd = kink_distance(pos, S, Wt, cols)
    n_bad, n = int(bad.sum()), bad.numel()
    clean = kd >= 2e-3
    l2 = ((got - want)[clean].norm() / want[clean].norm()).item() if clean.any() else 0.0
    worst_clean = (err[clean].max().item() / want.abs().max().item()) if clean.any() else 0.0
    print(f"[{tag}] d(pos): {n_bad} of {n} keys over {lim:.0e} x max (all within "
          f"{kd[bad].max().item() if n_bad else 0:.1e} of a kink); away from kinks ({int(clean.sum())} keys): worst "
          f"{worst_clean:.3e}, 2-norm rel err {l2:.3e}")
    assert clean.float().mean().item() >= min_clean, f"{tag}: only {int(clean.sum())} of {n} keys away from kinks: vacuous"
    assert n_bad <= 0.02 * n, f"{tag}: {n_bad} of {n} keys differ"
    assert n_bad == 0 or kd[bad].max().item() < 2e-3, f"{tag}: a key away from any kink differs"
    assert l2 < lim, f"{tag}: 2-norm rel err away from kinks {l2:.3e}"


@pytest.fixture(scope="module")
def cfg2():
    """One sample of BASELINE config 2's SCA attention: S=200, V=6, D=5, C=64, h=2 (M=40 000, N=100 000 per view)."""
    torch.manual_seed(0)
    S, D, V, C, h = 200, 5, 6, 64, 2
    p = lift_problem(S, D, V, C, h, 704, 256, {"X": 50, "Y": 50, "Z": 2}, seed=2024)
    rows = pick_rows(S, 256, 1)
    cot = torch.randn(V, len(rows), C, generator=torch.Generator().manual_seed(5))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, grads = oracle_rows(p, h, rows, cot)
    return dict(p=p, rows=rows, cot=cot, want=want, grads=grads, S=S, V=V, C=C, h=h)


# observed on MI355X (printed by the test, -s): see DESIGN.md section 3 for the table.
# out: max |err| relative to max |want| over the subset; gradients: max |err| / max |want| per tensor.
LIMITS = {
    _lib.PREC_F32: dict(out=2e-4, query=5e-4, k=5e-4, v=5e-4, pos=1e-3, table=2.5e-4),
    _lib.PREC_BF16: dict(out=1.5e-2, query=3e-2, k=3e-2, v=2e-2, pos=4e-2, table=3e-2),
    # fp16 operands (BASELINE config 5): 11-bit significands against bf16's 8 -- between the two tables above
    _lib.PREC_F16: dict(out=2.5e-3, query=5e-3, k=5e-3, v=4e-3, pos=8e-3, table=5e-3),
}
LIMITS[_lib.PREC_BF16X3] = LIMITS[_lib.PREC_F32]   # split-bf16 products: the f32 limits
TAG = {_lib.PREC_F32: "f32", _lib.PREC_BF16: "bf16", _lib.PREC_F16: "f16", _lib.PREC_BF16X3: "bf16x3"}
ALL_PREC = [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16]


@pytest.mark.parametrize("prec", ALL_PREC)
def test_cfg2_attention_rows_and_gradients(cfg2, prec):
    z = cfg2
    p, rows, h, V, S, C = z["p"], z["rows"], z["h"], z["V"], z["S"], z["C"]
    assert 0.2 < p["pinned"] < 0.7          # the rig really pins a large share of the keys to (-1, -1)
    ins = {n: p[n].clone().to(DEV).requires_grad_(True) for n in ("query", "k", "v", "pos", "table")}
    out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1, views=V,
                             precision=prec)                                       # (V, S*S, C)
    cot_full = torch.zeros_like(out)
    cot_full[:, rows.to(DEV)] = z["cot"].to(DEV)
    out.backward(cot_full)
    torch.cuda.synchronize()
    lim = LIMITS[prec]
    tag = TAG[prec]
    e_out = rel_err(out.detach()[:, rows.to(DEV)].cpu(), z["want"])
    print(f"\n[cfg2 {tag}] out rel err {e_out:.3e}")
    assert e_out < lim["out"], f"out: {e_out:.3e}"
    assert torch.isfinite(out).all()
    for n in ("query", "k", "v", "table"):
        e = rel_err(ins[n].grad.cpu(), z["grads"][n])
        print(f"[cfg2 {tag}] grad {n:6s} rel err {e:.3e}  (max |want| {z['grads'][n].abs().max().item():.3e})")
        assert e < lim[n], f"grad {n}: {e:.3e}"
    check_dpos(ins["pos"].grad, z["grads"]["pos"], p["pos"], S, p["table"].shape[-1], lim["pos"], f"cfg2 {tag}")


@pytest.mark.parametrize("prec", ALL_PREC)
def test_cfg2_split_region_and_cell_kernels(cfg2, prec):
    """The same sample as the SCA module runs it: the projector's pinned keys (two thirds) cell-sorted and attended
    through the cell kernels, the rest through the region kernels, one softmax.  The oracle result of the fixture is
    re-used: out, d(query), d(table) are invariant to the key order; d(k), d(v), d(pos) are the fixture's, permuted."""
    z = cfg2
    p, rows, h, V, S = z["p"], z["rows"], z["h"], z["V"], z["S"]
    perm, split = cell_split_perm(p, S)
    assert split < p["pos"].shape[1] * 0.5                                   # most keys go to the cell kernels
    ins = {n: p[n].clone() for n in ("query", "k", "v", "pos", "table")}
    for n in ("k", "v", "pos"):
        ins[n] = permute_keys(ins[n], perm)
    ins = {n: t.to(DEV).requires_grad_(True) for n, t in ins.items()}
    out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1, views=V,
                             precision=prec, cell_split=split)
    cot_full = torch.zeros_like(out)
    cot_full[:, rows.to(DEV)] = z["cot"].to(DEV)
    out.backward(cot_full)
    torch.cuda.synchronize()
    lim = LIMITS[prec]
    tag = TAG[prec]
    e_out = rel_err(out.detach()[:, rows.to(DEV)].cpu(), z["want"])
    print(f"\n[cfg2 split {tag}] out rel err {e_out:.3e} (cell segment: {p['pos'].shape[1] - split} of {p['pos'].shape[1]} keys)")
    assert e_out < lim["out"], f"out: {e_out:.3e}"
    for n in ("query", "k", "v", "table"):
        want = z["grads"][n] if n in ("query", "table") else permute_keys(z["grads"][n], perm)
        e = rel_err(ins[n].grad.cpu(), want)
        print(f"[cfg2 split {tag}] grad {n:6s} rel err {e:.3e}")
        assert e < lim[n], f"grad {n}: {e:.3e}"
    cols = (rows % S).tolist()
    check_dpos(ins["pos"].grad, permute_keys(z["grads"]["pos"], perm), permute_keys(p["pos"], perm), S,
               p["table"].shape[-1], lim["pos"], f"cfg2 split {tag}", cols=cols, min_clean=0.3)


def test_cfg3_batch8_launch_n_prob_48(cfg2):
    """One launch of config 3's shape: B = 8 samples x V = 6 views = 48 softmax problems per head (the bench's SCA call;
    every other test launches at most 6): the XCD remap of the problem index, the key workspace offsets and the
    per-problem cell sort at n_prob = 48.  Samples 0..6 are random; sample 7 is the fixture, whose rows are compared
    with the float64 oracle; every problem is compared between the bf16 and the f32 kernels.  Region kernels alone
    and the region / cell split."""
    z = cfg2
    p, rows, h, V, S, C = z["p"], z["rows"], z["h"], z["V"], z["S"], z["C"]
    B = 8
    N = p["pos"].shape[1]
    gen = torch.Generator(device=DEV).manual_seed(48)
    Hk, Wk = S // 2, S * 5
    rng = torch.tensor([1.0 / (Hk - 1.0), 1.0 / (Wk - 1.0)], device=DEV) * 5.0
    ref_o = p["pos"].to(DEV)          # any positions with the fixture's statistics: re-draw the offsets around the same refs
    perm, split = cell_split_perm(p, S)
    for use_split in (False, True):
        query = torch.cat((torch.randn(B - 1, C, S, S, device=DEV, generator=gen), p["query"].to(DEV)), 0)
        k = torch.cat((torch.randn((B - 1) * V, N, C, device=DEV, generator=gen), p["k"].to(DEV)), 0)
        v = torch.cat((torch.randn((B - 1) * V, N, C, device=DEV, generator=gen), p["v"].to(DEV)), 0)
        jit = torch.tanh(torch.randn(B - 1, V, N, 2, device=DEV, generator=gen)) * rng * 0.2
        pos = torch.cat(((ref_o[None] + jit).reshape((B - 1) * V, N, 2), ref_o), 0)
        cs = None
        if use_split:
            # every problem: fixture permutation first (pinned keys last), then its own cell sort of the cell segment
            pm = perm.to(DEV).repeat(B, 1)
            k, v, pos = permute_keys(k, pm), permute_keys(v, pm), permute_keys(pos, pm)
            a, b = ops.key_coords(pos[:, split:], S, p["table"].shape[-1], N - split)
            dyn = ops.cell_order(a, b) + split
            dyn = torch.cat((torch.arange(split, device=DEV)[None].expand(B * V, -1), dyn), 1)
            dyn[-V:] = torch.arange(N, device=DEV)[None]      # the fixture's problems are already sorted by perm
            k, v, pos = permute_keys(k, dyn), permute_keys(v, dyn), permute_keys(pos, dyn)
            cs = split
        res = {}
        for prec in (_lib.PREC_BF16, _lib.PREC_F32):
            ins = [t.clone().requires_grad_(True) for t in (query, k, v, pos, p["table"].to(DEV))]
            out = ops.attention_core(*ins, heads=h, groups=1, views=V, precision=prec, cell_split=cs)
            cot_full = torch.zeros_like(out)
            cot_full[-V:, rows.to(DEV)] = z["cot"].to(DEV)
            cot_full[:-V] = torch.randn(cot_full[:-V].shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)) * 0.05
            out.backward(cot_full)
            torch.cuda.synchronize()
            res[prec] = (out.detach(), [t.grad for t in ins])
            e = rel_err(out.detach()[-V:][:, rows.to(DEV)].cpu(), z["want"])
            print(f"\n[cfg3 n_prob=48 split={use_split} prec={prec}] fixture rows: out rel err {e:.3e}")
            assert e < LIMITS[prec]["out"]
            del ins, out, cot_full
        (ob, gb), (of, gf) = res[_lib.PREC_BF16], res[_lib.PREC_F32]
        e = rel_err(ob, of)
        print(f"[cfg3 n_prob=48 split={use_split}] bf16 vs f32, all 48 problems: out {e:.3e} "
              + " ".join(f"d{n} {rel_err(a, b_):.3e}" for n, a, b_ in zip(("query", "k", "v", "pos", "table"), gb, gf)))
        assert e < 1.5e-2
        for n, a, b_, l in zip(("query", "k", "v", "pos", "table"), gb, gf, (3e-2, 3e-2, 2e-2, 6e-2, 3e-2)):
            assert rel_err(a, b_) < l, n
        del res, query, k, v, pos
        torch.cuda.empty_cache()


@pytest.mark.parametrize("prec", ALL_PREC)
def test_tsa_geometry_bev200_rows_and_gradients(prec):
    """The TSA attention of configs 2-4 at full size: S = 200, N = 40 000 keys on the regular grid (offsets over the
    learned range 0.5 / (S - 1)), table 399 x 399 (rx = 1: different region-move and ring statistics from SCA's
    rx ~ 5), keys in the static k-d order.  256 query rows against the float64 oracle, forward and every gradient."""
    S, C, h = 200, 64, 2
    gen = torch.Generator().manual_seed(200)
    grid = O.normalized_grid(S, S, torch.float32).reshape(1, -1, 2)
    N = S * S
    pos = grid + torch.tanh(torch.randn(1, N, 2, generator=gen)) * (0.5 / (S - 1.0))
    order = torch.from_numpy(ops.kd_key_order(grid[0].double().numpy(), S, 2 * S - 1))
    p = dict(query=torch.randn(1, C, S, S, generator=gen), k=torch.randn(1, N, C, generator=gen),
             v=torch.randn(1, N, C, generator=gen), pos=pos[:, order].contiguous(),
             table=torch.randn(h, 2 * S - 1, 2 * S - 1, generator=gen) * 0.3)
    rows = pick_rows(S, 256, 3)
    cot = torch.randn(1, len(rows), C, generator=torch.Generator().manual_seed(7))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, grads = oracle_rows(p, h, rows, cot)
    ins = {n: p[n].clone().to(DEV).requires_grad_(True) for n in ("query", "k", "v", "pos", "table")}
    out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1, views=1,
                             precision=prec)
    cot_full = torch.zeros_like(out)
    cot_full[:, rows.to(DEV)] = cot.to(DEV)
    out.backward(cot_full)
    torch.cuda.synchronize()
    lim = LIMITS[prec]
    e = rel_err(out.detach()[:, rows.to(DEV)].cpu(), want)
    print(f"\n[tsa S=200 prec={prec}] out rel err {e:.3e}")
    assert e < lim["out"]
    for n in ("query", "k", "v", "table"):
        e = rel_err(ins[n].grad.cpu(), grads[n])
        print(f"[tsa S=200 prec={prec}] grad {n:6s} rel err {e:.3e}")
        assert e < lim[n], f"grad {n}: {e:.3e}"
    check_dpos(ins["pos"].grad, grads["pos"], p["pos"], S, 2 * S - 1, lim["pos"], f"tsa S=200 prec={prec}",
               cols=(rows % S).tolist(), min_clean=0.3)


def test_cfg2_bf16_agrees_with_f32_kernels_on_all_rows(cfg2):
    """The two precision modes against each other on EVERY query row (the oracle covers a subset): out and all
    gradients with a dense cotangent, so every region move / ring slide / window flush of the launch contributes."""
    z = cfg2
    p, h, V = z["p"], z["h"], z["V"]
    res = {}
    cot = None
    for prec in (_lib.PREC_F32, _lib.PREC_BF16):
        ins = {n: p[n].clone().to(DEV).requires_grad_(True) for n in ("query", "k", "v", "pos", "table")}
        out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1,
                                 views=V, precision=prec)
        if cot is None:
            cot = torch.randn(out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        out.backward(cot)
        torch.cuda.synchronize()
        res[prec] = dict(out=out.detach(), **{n: ins[n].grad for n in ins})
    a, b = res[_lib.PREC_F32], res[_lib.PREC_BF16]
    lim = dict(out=1.5e-2, query=3e-2, k=3e-2, v=2e-2, pos=4e-2, table=3e-2)
    for n in lim:
        e = rel_err(b[n], a[n])
        print(f"\n[cfg2 bf16 vs f32, all rows] {n:6s} rel diff {e:.3e}")
        assert e < lim[n], f"{n}: {e:.3e}"


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
def test_cfg1_bev50_full_rows(prec):
    """BASELINE config 1's BEV (S=50, one front camera at 128x128, batch 2): every query row against the oracle,
    SCA geometry (N = 6 250) and TSA geometry (regular grid, N = 2 500), forward and all gradients."""
    S, D, C, h = 50, 5, 64, 2
    T, K = [ring_rig(1, 128, 128)[0][0]], [np.array([[100., 0, 64, 0], [0, 100., 64, 0], [0, 0, 1, 0]])]
    gen = torch.Generator().manual_seed(50)
    pts = O.sample_3d_points({"X": 20, "Y": 10, "Z": 2}, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, 128, 128, 128, 128), 1)[0].reshape(1, -1, 2)[..., (1, 0)]
    cases = []
    N = ref.shape[1]
    rng = torch.tensor([1.0 / (S // 2 - 1.0), 1.0 / (S * D - 1.0)]) * 5.0
    pos = (ref + torch.tanh(torch.randn(2, N, 2, generator=gen)) * rng)
    order = torch.from_numpy(ops.kd_key_order(ref[0].double().numpy(), S, 2 * S * D - 1))
    cases.append(("sca", pos[:, order].contiguous(), 2 * S * D - 1))
    grid = O.normalized_grid(S, S, torch.float32).reshape(1, -1, 2)
    rng_t = torch.tensor([1.0 / (S - 1.0), 1.0 / (S - 1.0)]) * 0.5
    pos_t = grid + torch.tanh(torch.randn(2, S * S, 2, generator=gen)) * rng_t
    order_t = torch.from_numpy(ops.kd_key_order(grid[0].double().numpy(), S, 2 * S - 1))
    cases.append(("tsa", pos_t[:, order_t].contiguous(), 2 * S - 1))
    lim = LIMITS[prec]
    for name, pos, Wt in cases:
        B, N = pos.shape[:2]
        query = torch.randn(B, C, S, S, generator=gen)
        k, v = torch.randn(B, N, C, generator=gen), torch.randn(B, N, C, generator=gen)
        table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * 0.3
        cot = torch.randn(B, S * S, C, generator=gen)
        cpu = [t.clone().double().requires_grad_(True) for t in (query, k, v, pos, table)]
        c = C // h
        outs = []
        for b in range(B):
            o = O.attention_core(cpu[0][b].reshape(h, c, S * S), cpu[1][b].reshape(N, h, c).permute(1, 2, 0),
                                 cpu[2][b].reshape(N, h, c).permute(1, 2, 0), cpu[3][b:b + 1], cpu[4], S, S, 1, c ** -0.5)
            outs.append(o.reshape(C, S * S).t())
        want = torch.stack(outs, 0)
        want.backward(cot.double())
        gpu = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
        got = ops.attention_core(*gpu, heads=h, groups=1, views=1, precision=prec)
        got.backward(cot.to(DEV))
        torch.cuda.synchronize()
        e = rel_err(got.detach().cpu(), want.detach())
        print(f"\n[cfg1 {name} prec={prec}] out {e:.3e}")
        assert e < lim["out"]
        for n, a, b_ in zip(("query", "k", "v", "pos", "table"), gpu, cpu):
            if n == "pos":
                check_dpos(a.grad, b_.grad, pos, S, Wt, lim["pos"], f"cfg1 {name} prec={prec}")
                continue
            e = rel_err(a.grad.cpu(), b_.grad)
            print(f"[cfg1 {name} prec={prec}] grad {n} {e:.3e}")
            assert e < lim[n], f"{name} grad {n}: {e:.3e}"


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
def test_encoder_layer_s56_matches_reference(prec):
    """G4b on the GPU: this repo's EncoderLayer (HIP kernels) at S=56, C=64, D=5 against sampled outputs and
    gradients of the reference's own EncoderLayer (tests/golden/enclayer_s56.npz)."""
    from test_oracle_golden import check_sampled, enclayer_s56_setup, load
    z = load("enclayer_s56.npz")
    mg, c, T, K, layer = enclayer_s56_setup(device=DEV, precision=prec)
    layer = layer.to(DEV).train()
    bev_query, prev_bev, img_feat, cot = mg.enclayer_s56_inputs()
    ins = {"bev_query": bev_query.to(DEV).requires_grad_(True), "prev_bev": prev_bev.to(DEV).requires_grad_(True),
           "img_feat": img_feat.to(DEV).requires_grad_(True)}
    out, _ = layer(ins["bev_query"], ins["img_feat"], ins["prev_bev"], torch.zeros(c["B"], 2, 3, device=DEV),
                   torch.tensor(0), {}, False)
    out.backward(cot.to(DEV))
    torch.cuda.synchronize()
    f32 = prec == _lib.PREC_F32
    got = out.detach().flatten().cpu().numpy()[z["out_idx"]]
    scale = np.abs(z["out_val"]).max()
    e = np.abs(got - z["out_val"]).max() / scale
    print(f"\n[enclayer s56 prec={prec}] out err/max {e:.3e}")
    assert e < (3e-4 if f32 else 2e-2)
    params = dict(layer.named_parameters())
    worst = 0.0
    for name in z["names"]:
        name = str(name)
        g = ins[name[8:]].grad if name.startswith("grad_in.") else params[name[len("grad_param."):]].grad
        assert g is not None, name
        e = check_sampled(z, name, g, rtol=2e-3 if f32 else 5e-2, atol_frac=1e-3 if f32 else 5e-2,
                          floor_frac=2e-5 if f32 else 2e-3)
        if e is not None:                 # analytically-zero gradients are held to a noise floor, not to a ratio
            worst = max(worst, e)
    print(f"[enclayer s56 prec={prec}] worst sampled gradient err/absmax {worst:.3e}")
    # the check is not vacuous: a zeroed small gradient must fail it
    small = "grad_param.spatial_cross_attn.spatial_deform_attn.rpe_table"
    with pytest.raises(AssertionError):
        check_sampled(z, small, torch.zeros_like(params[small[len("grad_param."):]]), rtol=2e-3 if f32 else 5e-2,
                      atol_frac=1e-3 if f32 else 5e-2, floor_frac=2e-5 if f32 else 2e-3)


# BF16X3: the f32-layout cell kernels at 14 waves per workgroup run their 1024-thread instantiation here
@pytest.mark.parametrize("prec", [_lib.PREC_F16, _lib.PREC_BF16, _lib.PREC_BF16X3])
def test_cfg5_geometry_bev400_rows_and_gradients(prec):
    """The S = 400 geometry of BASELINE config 5 (M = 160 000, N = 400 000 per view, table 799 x 3999), one view, in
    the fp16 operand mode the config names (and in bf16):
    64 random query rows against the float64 oracle -- forward and, with a cotangent that is zero elsewhere, every
    gradient -- and constant V => every row returns that constant."""
    S, D, C, h = 400, 5, 64, 2
    p = lift_problem(S, D, 1, C, h, 1408, 512, {"X": 50, "Y": 50, "Z": 2}, seed=400, table_std=0.3)
    rows = pick_rows(S, 64, 2)
    cot = torch.randn(1, len(rows), C, generator=torch.Generator().manual_seed(6))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, grads = oracle_rows(p, h, rows, cot)
    ins = {n: p[n].clone().to(DEV).requires_grad_(True) for n in ("query", "k", "v", "pos", "table")}
    out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1, views=1,
                             precision=prec)
    cot_full = torch.zeros_like(out)
    cot_full[:, rows.to(DEV)] = cot.to(DEV)
    out.backward(cot_full)
    torch.cuda.synchronize()
    lim = LIMITS[prec]
    e = rel_err(out.detach()[:, rows.to(DEV)].cpu(), want)
    print(f"\n[cfg5 geometry {TAG[prec]}] out rel err {e:.3e}")
    assert e < lim["out"]
    for n in ("query", "k", "v", "table"):
        e = rel_err(ins[n].grad.cpu(), grads[n])
        print(f"[cfg5 geometry {TAG[prec]}] grad {n:6s} rel err {e:.3e}  (max |want| {grads[n].abs().max().item():.3e})")
        assert e < lim[n], f"grad {n}: {e:.3e}"
    cols = (rows % S).tolist()
    check_dpos(ins["pos"].grad, grads["pos"], p["pos"], S, p["table"].shape[-1], lim["pos"], f"cfg5 {TAG[prec]}", cols=cols,
               min_clean=0.3)
    # the same sample with the pinned keys split off to the cell kernels (13 row blocks per BEV column: 13 waves)
    perm, split = cell_split_perm(p, S)
    ins2 = {n: p[n].clone() for n in ("query", "k", "v", "pos", "table")}
    for n in ("k", "v", "pos"):
        ins2[n] = permute_keys(ins2[n], perm)
    ins2 = {n: t.to(DEV).requires_grad_(True) for n, t in ins2.items()}
    out2 = ops.attention_core(ins2["query"], ins2["k"], ins2["v"], ins2["pos"], ins2["table"], heads=h, groups=1, views=1,
                              precision=prec, cell_split=split)
    out2.backward(cot_full)
    torch.cuda.synchronize()
    e = rel_err(out2.detach()[:, rows.to(DEV)].cpu(), want)
    print(f"[cfg5 geometry {TAG[prec]}, split at {split}] out rel err {e:.3e}")
    assert e < lim["out"]
    for n in ("query", "k", "v", "table"):
        w = grads[n] if n in ("query", "table") else permute_keys(grads[n], perm)
        e = rel_err(ins2[n].grad.cpu(), w)
        print(f"[cfg5 geometry {TAG[prec]}, split] grad {n:6s} rel err {e:.3e}")
        assert e < lim[n], f"split grad {n}: {e:.3e}"
    check_dpos(ins2["pos"].grad, permute_keys(grads["pos"], perm), permute_keys(p["pos"], perm), S, p["table"].shape[-1],
               lim["pos"], f"cfg5 {TAG[prec]} split", cols=cols, min_clean=0.3)
    del ins2, out2
    with torch.no_grad():
        pat = torch.randn(C, generator=torch.Generator().manual_seed(1)).to(DEV)
        vconst = pat[None, None, :].expand_as(ins["v"]).contiguous()
        out = ops.attention_core(ins["query"].detach(), ins["k"].detach(), vconst, ins["pos"].detach(),
                                 ins["table"].detach(), heads=h, groups=1, views=1, precision=prec)
        torch.cuda.synchronize()
        assert (out - pat).abs().max().item() < (2e-2 if prec == _lib.PREC_BF16 else 4e-3)


def _randomize(mod, seed, scale=0.25):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in sorted(mod.named_parameters()):
            if name.endswith("norm.weight"):
                prm.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))
            elif "rpe_table" in name:
                prm.copy_(0.3 * torch.randn(prm.shape, generator=g))
            else:
                fan = prm[0].numel() if prm.dim() > 1 else 1
                prm.copy_(torch.randn(prm.shape, generator=g) * (scale if prm.dim() == 1 else 1.0 / math.sqrt(fan)))


def _check_module_grads(tag, named_got, want, f32):
    # f32: the precision code (0 f32, 1 bf16, 2 fp16) -- the limits are per code
    """per-tensor: max |err| <= lim * max |want of this tensor|; the analytically-zero proj_k.bias to a noise floor.
    The offset heads' gradients are sums of d(pos) over the keys, and d(pos) jumps at kinks (check_dpos: a float32
    evaluation lands on the other side of a kink for ~0.5 % of the keys of this geometry, by up to 6 % of the largest
    entry): their limit in f32 mode is that of a sum with such outliers, not of the smooth tensors."""
    lim = {0: 3e-3, 1: 6e-2, 2: 1e-2}[f32]
    lim_off = {0: 1.5e-2, 1: 6e-2, 2: 2e-2}[f32]
    big = max(w.abs().max().item() for w in want.values())
    worst = 0.0
    for name, w in want.items():
        g = named_got[name]
        assert g is not None, name
        if name.endswith("proj_k.bias"):
            assert g.abs().max().item() <= {0: 2e-4, 1: 2e-2, 2: 4e-3}[f32] * big, f"{tag} {name}: analytically zero"
            continue
        e = rel_err(g.cpu(), w)
        worst = max(worst, e)
        assert e < (lim_off if "conv_offset" in name else lim), f"{tag} grad {name}: {e:.3e}"
    print(f"[{tag}] worst gradient err / max|want| over {len(want)} tensors: {worst:.3e}")


@pytest.mark.parametrize("prec", ALL_PREC)
def test_sca_module_bev200_six_views_rows(prec):
    """The whole SCA module at the benchmark's size (S = 200, V = 6, D = 5, 64 x 176 features), through the
    SpatialCrossAttn wrapper: projector, offset heads, even / odd row split, static key order with the pinned keys
    split off and cell-sorted, feature sampling, the K | V GEMM, operand packing, both attention paths, proj_out --
    against oracle.sca_forward(rows=...) (float64) on 128 BEV positions: output and every parameter / input gradient."""
    _sca_module_rows(prec, S=200, img_w=704, img_h=256, n_rows=128, max_split=50000)


def test_cfg5_sca_module_bev400_six_views_fp16_rows():
    """BASELINE config 5 at the MODULE level (VERDICT r04 'weak' 3): SpatialCrossAttn at S = 400, V = 6, fp16 operands on
    128 x 352 feature maps (6 cameras 512 x 1408) -- kv_project on the large maps, key positions for 6 x 400 000 keys, the
    per-call cell sort, the attention segments and the sampling backward -- against oracle.sca_forward(rows=...) in
    float64 on 32 BEV positions, output and every parameter / input gradient, at the fp16 limits."""
    _sca_module_rows(_lib.PREC_F16, S=400, img_w=1408, img_h=512, n_rows=32, max_split=200000)


def _sca_module_rows(prec, S, img_w, img_h, n_rows, max_split):
    from bevrender_amd.model.SCA import SpatialCrossAttn
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    D, V, C, h, B, Hi, Wi = 5, 6, 64, 2, 1, img_h // 4, img_w // 4
    T, K = ring_rig(V, img_w, img_h)
    proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=img_w, img_height=img_h,
                               ori_img_width=img_w, ori_img_height=img_h, device=DEV)
    bound = {"X": 50, "Y": 50, "Z": 2}
    sca = SpatialCrossAttn(bound, proj, S, D, -1.0, C, h, 1, 1, 3, B, True, n_views=V, precision=prec)
    _randomize(sca, 11)
    gen = torch.Generator().manual_seed(12)
    query, x = torch.randn(B, C, S, S, generator=gen), torch.randn(B, V, C, Hi, Wi, generator=gen)
    rows = pick_rows(S, n_rows, 5)
    cot = torch.randn(B, C, len(rows), generator=gen)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    att = sca.spatial_deform_attn
    used = [n for n, _ in att.named_parameters() if not n.startswith(("proj_q", "proj_views"))]
    pw = {k_: v_.detach().clone().double().requires_grad_(k_ in used) for k_, v_ in att.state_dict().items()}
    qc, xc = query.clone().double().requires_grad_(True), x.clone().double().requires_grad_(True)
    pts = O.sample_3d_points(bound, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, img_w, img_h, img_w, img_h), B).double()
    want = O.sca_forward(pw, xc, qc, ref, n_heads=h, depth_dim=D, rows=rows)
    (want * cot.double()).sum().backward()
    sca = sca.to(DEV)
    qg, xg = query.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    out, _ = sca(qg, xg.reshape(B * V, C, Hi, Wi), torch.tensor(0), None, False)
    assert sca.reference_points(0, qg.device)[2] < max_split      # the split is in use: most keys on the cell / tap kernels
    cot_full = torch.zeros(B, C, S * S, device=DEV)
    cot_full[:, :, rows.to(DEV)] = cot.to(DEV)
    out.backward(cot_full.reshape(B, C, S, S))
    torch.cuda.synchronize()
    f32 = 0 if int(prec) == _lib.PREC_BF16X3 else int(prec)   # the split-bf16 mode is held to the f32 limits
    e = rel_err(out.detach().reshape(B, C, S * S)[:, :, rows.to(DEV)].cpu(), want.detach())
    print(f"\n[sca module S={S} V=6 prec={prec}] out rel err {e:.3e}")
    assert e < {0: 3e-4, 1: 2e-2, 2: 3e-3}[f32]
    got = {n: prm.grad for n, prm in att.named_parameters()}
    got.update({"in.query": qg.grad, "in.x": xg.grad})
    wnt = {n: pw[n].grad for n in used}
    wnt.update({"in.query": qc.grad, "in.x": xc.grad})
    _check_module_grads(f"sca module S={S} V=6 prec={prec}", got, wnt, f32)


@pytest.mark.parametrize("prec", ALL_PREC)
def test_tsa_module_bev200_rows(prec):
    """The whole TSA module at S = 200 (N = 40 000 grid keys, depthwise 3x3 offset head, sampling of prev_bev, K | V GEMM,
    packing, attention, proj_out) against oracle.tsa_forward(rows=...) (float64) on 128 BEV positions."""
    _tsa_module_rows(prec, 200, 128)


def test_cfg5_tsa_module_bev400_fp16_rows():
    """BASELINE config 5 at the module level: TSADeformableAttention at S = 400 (N = 160 000 grid keys), fp16 operands,
    against oracle.tsa_forward(rows=...) in float64 on 128 BEV positions (with 32 the offset head's gradients -- sums of
    d(pos) over the keys, which jumps at kinks -- were 2.5e-2 off against their 2e-2 limit: too few rows for that sum)."""
    _tsa_module_rows(_lib.PREC_F16, 400, 128)


def _tsa_module_rows(prec, S, n_rows):
    from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
    C, h, B = 64, 2, 1
    tsa = TSADeformableAttention(S, C, h, 1, 1, 3, True, B, n_views=1, precision=prec)
    _randomize(tsa, 21)
    gen = torch.Generator().manual_seed(22)
    query, prev = torch.randn(B, C, S, S, generator=gen), torch.randn(B, C, S, S, generator=gen)
    rows = pick_rows(S, n_rows, 6)
    cot = torch.randn(B, C, len(rows), generator=gen)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    used = [n for n, _ in tsa.named_parameters() if not n.startswith(("proj_q", "proj_views"))]
    pw = {k_: v_.detach().clone().double().requires_grad_(k_ in used) for k_, v_ in tsa.state_dict().items()}
    qc, pc = query.clone().double().requires_grad_(True), prev.clone().double().requires_grad_(True)
    want = O.tsa_forward(pw, qc, pc, n_heads=h, rows=rows)
    (want * cot.double()).sum().backward()
    tsa = tsa.to(DEV)
    qg, pg = query.to(DEV).requires_grad_(True), prev.to(DEV).requires_grad_(True)
    out, _ = tsa(pg, qg, None, False)
    cot_full = torch.zeros(B, C, S * S, device=DEV)
    cot_full[:, :, rows.to(DEV)] = cot.to(DEV)
    out.backward(cot_full.reshape(B, C, S, S))
    torch.cuda.synchronize()
    f32 = 0 if int(prec) == _lib.PREC_BF16X3 else int(prec)   # the split-bf16 mode is held to the f32 limits
    e = rel_err(out.detach().reshape(B, C, S * S)[:, :, rows.to(DEV)].cpu(), want.detach())
    print(f"\n[tsa module S={S} prec={prec}] out rel err {e:.3e}")
    assert e < {0: 3e-4, 1: 2e-2, 2: 3e-3}[f32]
    got = {n: prm.grad for n, prm in tsa.named_parameters()}
    got.update({"in.query": qg.grad, "in.prev": pg.grad})
    wnt = {n: pw[n].grad for n in used}
    wnt.update({"in.query": qc.grad, "in.prev": pc.grad})
    _check_module_grads(f"tsa module S={S} prec={prec}", got, wnt, f32)


@pytest.mark.parametrize("prec", [_lib.PREC_F16, _lib.PREC_BF16])
def test_cfg5_tsa_geometry_bev400_rows_and_gradients(prec):
    """TSA at config 5's BEV: S = 400, N = 160 000 grid keys, table 799 x 799, bf16 operands; 64 rows against the
    float64 oracle, forward and every gradient."""
    S, C, h = 400, 64, 2
    gen = torch.Generator().manual_seed(401)
    grid = O.normalized_grid(S, S, torch.float32).reshape(1, -1, 2)
    N = S * S
    pos = grid + torch.tanh(torch.randn(1, N, 2, generator=gen)) * (0.5 / (S - 1.0))
    order = torch.from_numpy(ops.kd_key_order(grid[0].double().numpy(), S, 2 * S - 1))
    p = dict(query=torch.randn(1, C, S, S, generator=gen), k=torch.randn(1, N, C, generator=gen),
             v=torch.randn(1, N, C, generator=gen), pos=pos[:, order].contiguous(),
             table=torch.randn(h, 2 * S - 1, 2 * S - 1, generator=gen) * 0.3)
    rows = pick_rows(S, 64, 4)
    cot = torch.randn(1, len(rows), C, generator=torch.Generator().manual_seed(8))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, grads = oracle_rows(p, h, rows, cot)
    ins = {n: p[n].clone().to(DEV).requires_grad_(True) for n in ("query", "k", "v", "pos", "table")}
    out = ops.attention_core(ins["query"], ins["k"], ins["v"], ins["pos"], ins["table"], heads=h, groups=1, views=1,
                             precision=prec)
    cot_full = torch.zeros_like(out)
    cot_full[:, rows.to(DEV)] = cot.to(DEV)
    out.backward(cot_full)
    torch.cuda.synchronize()
    lim = LIMITS[prec]
    e = rel_err(out.detach()[:, rows.to(DEV)].cpu(), want)
    print(f"\n[cfg5 tsa S=400 {TAG[prec]}] out rel err {e:.3e}")
    assert e < lim["out"]
    for n in ("query", "k", "v", "table"):
        e = rel_err(ins[n].grad.cpu(), grads[n])
        print(f"[cfg5 tsa S=400 {TAG[prec]}] grad {n:6s} rel err {e:.3e}")
        assert e < lim[n], f"grad {n}: {e:.3e}"
    check_dpos(ins["pos"].grad, grads["pos"], p["pos"], S, 2 * S - 1, lim["pos"], f"cfg5 tsa S=400 {TAG[prec]}",
               cols=(rows % S).tolist(), min_clean=0.3)


def test_correlation_head_at_config3_width():
    """The Gram, the contrastive and the lifted-structure loss at the width the benchmark runs them: B = 8,
    E = C S^2 = 64 * 200 * 200 = 2.56 M (the flattened BEV; every other parity case stops at E = 50 176).  Value and
    both gradients against the float64 oracle restatement (PARITY UNPINNED as a restatement of pytorch_metric_learning)."""
    from bevrender_amd.loss.contrastive_loss import ContrastiveLoss
    from bevrender_amd.loss.lift_loss import LiftedStructureLoss
    B, E = 8, 64 * 200 * 200
    gen = torch.Generator().manual_seed(256)
    cam = torch.randn(B, E, generator=gen)
    mp = cam + torch.linspace(3.0, 40.0, B)[:, None] * torch.randn(B, E, generator=gen)
    want_D = O.pairwise_corr(cam.double(), mp.double())
    got_D = ops.pairwise_corr(cam.to(DEV), mp.to(DEV))
    torch.cuda.synchronize()
    e = rel_err(got_D.cpu(), want_D)
    print(f"\n[corr E=2.56M] Gram 2 - 2 cam map^T rel err {e:.3e}")
    assert e < 2e-5
    for mod, fn in ((ContrastiveLoss(), O.contrastive_loss), (LiftedStructureLoss(), O.lifted_structure_loss)):
        cc, mc = cam.clone().double().requires_grad_(True), mp.clone().double().requires_grad_(True)
        want = fn(cc, mc)
        want.backward()
        cg, mg = cam.clone().to(DEV).requires_grad_(True), mp.clone().to(DEV).requires_grad_(True)
        got = mod.get_loss(cg, mg)
        got.backward()
        torch.cuda.synchronize()
        name = type(mod).__name__
        print(f"[corr E=2.56M] {name}: {got.item():.6f} vs {want.item():.6f}")
        assert abs(got.item() - want.item()) < 5e-5 * max(1.0, abs(want.item())), name
        for a, b_ in ((cg, cc), (mg, mc)):
            if b_.grad.abs().max() == 0:
                assert a.grad.abs().max().item() < 1e-9
            else:
                assert rel_err(a.grad.cpu().double(), b_.grad) < 5e-4, name


@pytest.mark.parametrize("B", [2, 8])
def test_retrieval_losses_match_oracle(B):
    """ContrastiveLoss / LiftedStructureLoss / TripletLossMetricLearning (HIP Gram + margins) against the oracle's
    restatement at config 3's sizes (B = 8, E = 64*28*28): value and both gradients.  The oracle functions are
    PARITY UNPINNED (pytorch_metric_learning absent); this pins product == oracle."""
    from bevrender_amd.loss.contrastive_loss import ContrastiveLoss
    from bevrender_amd.loss.lift_loss import LiftedStructureLoss
    from bevrender_amd.loss.triplet_loss_metric import TripletLossMetricLearning
    E = 64 * 28 * 28
    gen = torch.Generator().manual_seed(B)
    cam = torch.randn(B, E, generator=gen)
    # noise levels from "clearly matched" to "barely matched": some triplets are semi-hard with a non-zero loss
    mp = cam + torch.linspace(3.0, 40.0, B)[:, None] * torch.randn(B, E, generator=gen)
    for mod, fn in ((ContrastiveLoss(), O.contrastive_loss), (LiftedStructureLoss(), O.lifted_structure_loss),
                    (TripletLossMetricLearning(), O.triplet_margin_loss)):
        cc, mc = cam.clone().double().requires_grad_(True), mp.clone().double().requires_grad_(True)
        want = fn(cc, mc)
        want.backward()
        cg, mg = cam.clone().to(DEV).requires_grad_(True), mp.clone().to(DEV).requires_grad_(True)
        got = mod.get_loss(cg, mg)
        assert got.ndim == 0
        got.backward()
        torch.cuda.synchronize()
        name = type(mod).__name__
        assert abs(got.item() - want.item()) < 2e-5 * max(1.0, abs(want.item())), f"{name}: {got.item()} vs {want.item()}"
        for a, b_ in ((cg, cc), (mg, mc)):
            if b_.grad.abs().max() == 0:
                assert a.grad.abs().max().item() < 1e-9
            else:
                assert rel_err(a.grad.cpu().double(), b_.grad) < 2e-4, name
