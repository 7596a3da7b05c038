"""Tap kernels (csrc/attn_tap*.hip): the projector-pinned keys attended without K and V.

Three levels: the entry points against a float64 restatement of their definition (tools/tap_check.py); the host path
ops.attention_core(tap_source=True) -- region kernels on the scattered keys, tap kernels on the pinned ones, merged through
(O, LSE) -- against the oracle's materialised attention on sampled + projected keys (reference
model/SCA_deform_attn.py:290-413), forward and every gradient; and against the cell kernels on the same keys."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(got, want):
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)


KERNEL_CASES = {
    "sorted": dict(P=2, h=2, S=24, N=500, Wt=2 * 24 * 3 - 1),
    "ragged": dict(P=3, h=2, S=20, N=333, Wt=2 * 20 * 5 - 1, seed=1),
    # keys all over the table, not sorted: every 32-key tile is emitted in several masked passes (forward, query side) and
    # takes the per-pair gather on the key side
    "unsorted_wide": dict(P=1, h=2, S=18, N=200, Wt=2 * 18 * 5 - 1, spread=(12.0, 30.0), sort=False, seed=2),
    "three_row_blocks": dict(P=1, h=1, S=40, N=1000, Wt=2 * 40 * 5 - 1, seed=3),
    "fp16": dict(P=1, h=2, S=24, N=400, Wt=2 * 24 * 3 - 1, seed=5, prec=_lib.PREC_F16, gscale=2.0, headroom=8.0),
}


@pytest.mark.parametrize("name", list(KERNEL_CASES))
def test_tap_entry_points_match_their_float64_definition(name):
    import tap_check
    r = tap_check.check_case(name, **KERNEL_CASES[name])
    f16 = KERNEL_CASES[name].get("prec") == _lib.PREC_F16
    lim_f, lim_g = (1e-3, 4e-3) if f16 else (4e-3, 3e-2)
    assert r["flagged"] == 0 and r["dead"] == 0.0
    assert r["exact"][0] < lim_f and r["exact"][1] < 10 * lim_f, r          # Rn, LSE (log2 units)
    for k in ("dG", "dGb", "dtable", "da", "db", "dys", "dxs"):
        assert r[k] < lim_g, (k, r)


def test_tap_bwd_k_through_the_c_abi_as_the_header_describes():
    """include/bevrender_hip.h declared bevr_attn_tap_bwd_k's table operand as `table_pair`; the kernel reads
    the PLAIN packed table [heads][Wp][Hp + 1].  The header now says so; this test builds the operand from the header's
    words alone (tap_check.header_table, not ops.pack_table) and holds d(key_a), d(key_b) -- the two outputs that read it --
    to the float64 definition, and checks that the pair table (what the old declaration asked for) does NOT pass."""
    import tap_check
    kw = dict(P=2, h=2, S=24, N=500, Wt=2 * 24 * 3 - 1, seed=7)
    r = tap_check.check_case("header table", from_header=True, **kw)
    for k in ("da", "db", "dys", "dxs"):
        assert r[k] < 3e-2, (k, r)
    geom, a, b, ys, xs, G, Gb, T = tap_check.make_case(**kw)
    assert tuple(tap_check.header_table(T, geom).shape) == (geom.heads, geom.Wp, geom.Hp + 1)
    assert torch.equal(tap_check.header_table(T, geom), ops.pack_table(T.float(), geom).contiguous())


def test_tap_forward_recomputes_rows_whose_weights_underflow_the_static_reference():
    """logit scale 200: the upper bound of a row's logits is hundreds of binades above the logits that carry its mass,
    every weight underflows against the static reference, the column is flagged and recomputed with an online maximum.
    Compared with the restatement on the SAME rounded operands (at this scale the bf16 rounding of G is worth ~1 in log2)."""
    import tap_check
    r = tap_check.check_case("big logits", P=1, h=2, S=16, N=300, Wt=2 * 16 * 3 - 1, gscale=200.0, seed=4)
    assert r["flagged"] > 0
    assert r["mimic"][0] < 2e-3 and r["mimic"][1] < 1e-2, r


def _tap_problem(B, V, C, h, S, D, Hi, Wi, n_pin, seed, feat_scale=1.0):
    """An SCA-shaped call: per view N keys, the first N - n_pin scattered over the image, the last n_pin pinned to pixel
    (0, 0) and moved by offsets inside the learned range (tanh * 5 / (Hk - 1), 5 / (Wk - 1)), cell-sorted."""
    gen = torch.Generator().manual_seed(seed)
    Hk, Wk = S // 2, S * D
    N = Hk * Wk
    P = B * V
    query = torch.randn(B, C, S, S, generator=gen)
    feat = torch.randn(P, Hi, Wi, C, generator=gen) * feat_scale
    Wkv = torch.randn(2 * C, C, generator=gen) * C ** -0.5
    bkv = torch.randn(2 * C, generator=gen) * 0.3
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    scat = (torch.rand(P, N - n_pin, 2, generator=gen) * 2 - 1) * 1.05
    off = torch.tanh(torch.randn(P, n_pin, 2, generator=gen) * 1.5) * torch.tensor([5.0 / (Hk - 1), 5.0 / (Wk - 1)])
    pin = off - 1.0
    a, b = ops.key_coords(pin, S, 2 * S * D - 1, n_pin)
    order = ops.cell_order(a, b)
    pin = pin.gather(1, order[..., None].expand(-1, -1, 2))
    pos = torch.cat((scat, pin), 1)
    return query, feat, Wkv, bkv, pos, table, N - n_pin


def _oracle_chain(query, feat, Wkv, bkv, pos, table, h, V):
    """sample -> proj_k | proj_v -> materialised attention, per view, in the oracle's arithmetic."""
    B, C, S, _ = query.shape
    c = C // h
    P, N, _ = pos.shape
    grid = pos[:, None, :, (1, 0)]                                                  # (P, 1, N, 2) in (x, y)
    xs = F.grid_sample(feat.permute(0, 3, 1, 2), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    kv = F.linear(xs[:, :, 0].permute(0, 2, 1), Wkv, bkv)                             # (P, N, 2C)
    outs = []
    for p in range(P):
        q = query[p // V].reshape(h, c, S * S)
        kk = kv[p, :, :C].reshape(N, h, c).permute(1, 2, 0)
        vv = kv[p, :, C:].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[p:p + 1], table, S, S, 1, c ** -0.5)
        outs.append(o.reshape(C, S * S).t())
    return torch.stack(outs, 0)


TAP_CFGS = [
    # B, V, C, h, S, D, Hi, Wi, pinned keys per view.  The tap contract: 2.5 (Hi - 1) / (S / 2 - 1) < 3 and
    # 2.5 (Wi - 1) / (S D - 1) < 2 (what SCADeformableAttention._pinned_keys_tap checks), or an image inside the tap grid
    (1, 2, 64, 2, 12, 3, 6, 20, 128),
    (2, 1, 32, 1, 16, 5, 6, 10, 448),
    (1, 1, 64, 2, 34, 2, 16, 44, 704),       # three 16-row blocks per column, ragged
    (1, 2, 16, 2, 10, 3, 3, 2, 64),          # an image smaller than the 4 x 3 tap grid
]


@pytest.mark.parametrize("cfg", TAP_CFGS)
def test_attention_core_with_tap_source_matches_the_oracle(cfg):
    B, V, C, h, S, D, Hi, Wi, n_pin = cfg
    ins = _tap_problem(B, V, C, h, S, D, Hi, Wi, n_pin, seed=sum(cfg))
    split = ins[-1]
    cpu = [t.clone().double().requires_grad_(True) for t in ins[:-1]]
    want = _oracle_chain(*cpu, h, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    want.backward(cot)
    gpu = [t.clone().to(DEV).requires_grad_(True) for t in ins[:-1]]
    query, feat, Wkv, bkv, pos, table = gpu
    got = ops.attention_core(query, None, None, pos, table, heads=h, groups=1, views=V, precision=_lib.PREC_BF16,
                             kv_source=(feat, Wkv, bkv), cell_split=split, tap_source=True)
    got.backward(cot.float().to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().float().numpy(), rtol=3e-2, atol=1.5e-2)
    for n, a, b in zip(["query", "feat", "Wkv", "bkv", "pos", "table"], gpu, cpu):
        e = rel_err(a.grad.double().cpu(), b.grad)
        assert e < 3e-2, f"grad {n}: rel err {e:.3e}"


@pytest.mark.parametrize("cfg", TAP_CFGS[:3])
def test_attention_core_with_tap_source_in_fp16_matches_the_oracle(cfg):
    """BASELINE config 5's operand type on the tap route (round 5: headroom 8, cotangents scaled by a power of two): the
    same comparison at the fp16 limits of tests/test_gpu_fullsize.py, with a mean-type loss (cotangents ~1e-5: fp16
    subnormals without the scale)."""
    B, V, C, h, S, D, Hi, Wi, n_pin = cfg
    ins = _tap_problem(B, V, C, h, S, D, Hi, Wi, n_pin, seed=sum(cfg) + 1)
    split = ins[-1]
    cpu = [t.clone().double().requires_grad_(True) for t in ins[:-1]]
    want = _oracle_chain(*cpu, h, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 1e-5
    want.backward(cot)
    gpu = [t.clone().to(DEV).requires_grad_(True) for t in ins[:-1]]
    query, feat, Wkv, bkv, pos, table = gpu
    assert ops.tap_supported(_lib.PREC_F16, 1)
    got = ops.attention_core(query, None, None, pos, table, heads=h, groups=1, views=V, precision=_lib.PREC_F16,
                             kv_source=(feat, Wkv, bkv), cell_split=split, tap_source=True)
    got.backward(cot.float().to(DEV))
    torch.cuda.synchronize()
    e = rel_err(got.detach().double().cpu(), want.detach())
    assert e < 2.5e-3, f"out: rel err {e:.3e}"
    for n, a, b in zip(["query", "feat", "Wkv", "bkv", "pos", "table"], gpu, cpu):
        e = rel_err(a.grad.double().cpu(), b.grad)
        assert e < 5e-3, f"grad {n}: rel err {e:.3e}"


def test_tap_and_cell_kernels_agree_on_the_same_keys():
    """the same call with the pinned keys on the tap kernels and on the cell kernels (K, V formed): two routes to one
    softmax, both in bf16 operands."""
    B, V, C, h, S, D, Hi, Wi, n_pin = 1, 2, 64, 2, 34, 3, 12, 30, 1024
    ins = _tap_problem(B, V, C, h, S, D, Hi, Wi, n_pin, seed=9)
    split = ins[-1]
    res = []
    for tap in (True, False):
        gpu = [t.clone().to(DEV).requires_grad_(True) for t in ins[:-1]]
        query, feat, Wkv, bkv, pos, table = gpu
        out = ops.attention_core(query, None, None, pos, table, heads=h, groups=1, views=V, precision=_lib.PREC_BF16,
                                 kv_source=(feat, Wkv, bkv), cell_split=split, tap_source=tap)
        out.square().mean().backward()
        res.append([out.detach()] + [t.grad for t in gpu])
    torch.cuda.synchronize()
    for n, a, b in zip(["out", "query", "feat", "Wkv", "bkv", "pos", "table"], *res):
        assert rel_err(a, b) < 3e-2, f"{n}: {rel_err(a, b):.3e}"


def test_tap_path_with_a_loose_logit_bound_takes_the_exact_pass():
    """pixel (3, 2) of the feature map -- never sampled by a pinned key (xs < 0.6 here) -- carries features 300x
    the others: the static bound of every row is set by a logit no key has, the tap forward's weights underflow against
    it and the exact pass must repair every column.  Same comparison as above."""
    B, V, C, h, S, D, Hi, Wi = 1, 1, 32, 1, 16, 3, 8, 12
    n_pin = (S // 2) * S * D        # every key pinned: no scattered key samples the inflated pixel either
    ins = list(_tap_problem(B, V, C, h, S, D, Hi, Wi, n_pin, seed=21))
    ins[1][:, 3, 2, :] *= 300.0
    split = ins[-1]
    cpu = [t.clone().double().requires_grad_(True) for t in ins[:-1]]
    want = _oracle_chain(*cpu, h, V)
    want.square().mean().backward()
    gpu = [t.clone().to(DEV).requires_grad_(True) for t in ins[:-1]]
    query, feat, Wkv, bkv, pos, table = gpu
    got = ops.attention_core(query, None, None, pos, table, heads=h, groups=1, views=V, precision=_lib.PREC_BF16,
                             kv_source=(feat, Wkv, bkv), cell_split=split, tap_source=True)
    got.square().mean().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().float().numpy(), rtol=3e-2, atol=1.5e-2)
    for n, a, b in zip(["query", "pos", "table"], (query, pos, table), (cpu[0], cpu[4], cpu[5])):
        e = rel_err(a.grad.double().cpu(), b.grad)
        assert e < 3e-2, f"grad {n}: rel err {e:.3e}"
