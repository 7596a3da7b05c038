"""Slab-stationary query-side backward (csrc/attn_slab_bwd_q.hip, bevr_attn_slab_bwd_q) against the query-tile kernel it
replaces on the product path (bevr_attn_bwd_q) and against the oracle's materialised attention
(reference model/SCA_deform_attn.py:331-413 / model/TSA_deform_attn.py:245-333 differentiated by autograd).

Both kernels evaluate the same arithmetic per (query, key) pair -- the same 16-bit operands, 16-bit packed tap weights,
64-bit fixed-point table-gradient cells -- up to ONE rounding: the table column coordinate is j rx + b here and
j rx + (b - region origin) there, so a tap weight can round to the neighbouring 16-bit value (2^-9 of the weight in bf16,
2^-12 in fp16, on some pairs).  Observed differences: bf16 8e-5 ... 4e-3, fp16 3e-5 ... 6e-4 of the largest entry -- the size
of the modes' own errors.  So the A/B limits are the modes' (1e-2 / 2e-3), and what pins the kernel is the second test:
against the float64 oracle the slab kernel must be as accurate as the query-tile kernel (within 1.5x + a floor) and inside
half the bf16 limit of tests/test_gpu_ops.py.  Cases: ragged sizes, several row blocks, a
table wider than a slab by two orders of magnitude, channel groups, keys far outside the table (the clamped body), a
single key, fp16."""
import os

import pytest
import torch

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(got, want):
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)


def _problem(B, V, C, h, g, S, N, Wt, seed, spread=1.05, far=0.0):
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen)
    kv = torch.randn(B * V, N, 2 * C, generator=gen)
    pos = (torch.rand(B * V * g, N, 2, generator=gen) * 2 - 1) * spread
    if far > 0:      # a share of the keys far outside [-1, 1]: table rows / columns beyond the table's ends
        m = torch.rand(B * V * g, N, 1, generator=gen) < far
        pos = torch.where(m, pos * 4.0, pos)
    table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * 0.3
    return query, kv, pos, table


def _run(ins, h, g, V, prec, slab):
    os.environ["BEVR_SLAB"] = "2" if slab else "0"      # 2: the slab kernel for every table width (ops.slab_supported)
    try:
        query, kv, pos, table = (t.clone().to(DEV).requires_grad_(True) for t in ins)
        out = ops.attention_core(query, None, None, pos, table, heads=h, groups=g, views=V, precision=prec, kv=kv)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(77)).to(DEV)
        out.backward(cot)
        torch.cuda.synchronize()
        return out.detach(), query.grad, table.grad, kv.grad, pos.grad
    finally:
        os.environ.pop("BEVR_SLAB", None)


CASES = {
    #            B  V  C   h  g  S   N     Wt                 extra
    "tsa_small": (2, 1, 64, 2, 1, 12, 144, 23, {}),
    "sca_small": (1, 2, 64, 2, 1, 12, 6 * 36, 2 * 12 * 3 - 1, {}),
    "ragged": (1, 1, 32, 1, 1, 21, 333, 2 * 21 * 5 - 1, {}),
    "three_row_blocks": (1, 1, 64, 2, 1, 70, 1500, 2 * 70 * 2 - 1, {}),
    "groups": (1, 1, 64, 4, 2, 16, 400, 2 * 16 * 3 - 1, {}),
    "far_keys": (1, 2, 64, 2, 1, 24, 700, 2 * 24 * 3 - 1, {"far": 0.3}),
    "one_key": (1, 1, 64, 2, 1, 9, 1, 2 * 9 * 5 - 1, {}),
    "wide_table": (1, 1, 64, 2, 1, 40, 3000, 2 * 40 * 5 - 1, {"spread": 1.2}),
}


@pytest.mark.parametrize("prec", [_lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("name", list(CASES))
def test_slab_bwd_q_agrees_with_the_query_tile_kernel(name, prec):
    B, V, C, h, g, S, N, Wt, extra = CASES[name]
    ins = _problem(B, V, C, h, g, S, N, Wt, seed=len(name) + 7 * S, **extra)
    assert ops.slab_supported(prec, S)
    o1, dq1, dt1, dkv1, dp1 = _run(ins, h, g, V, prec, slab=True)
    o0, dq0, dt0, dkv0, dp0 = _run(ins, h, g, V, prec, slab=False)
    assert torch.equal(o1, o0)                      # the forward does not depend on the switch
    # nor does the key side (float atomics in the position gradient: equal up to their order)
    assert rel_err(dkv1, dkv0) < 1e-5 and rel_err(dp1, dp0) < 1e-4
    e_q, e_t = rel_err(dq1, dq0), rel_err(dt1, dt0)
    print(f"[slab vs tile {name} prec={prec}] dQ {e_q:.2e} d(table) {e_t:.2e}")
    lim = 1e-2 if prec == _lib.PREC_BF16 else 2e-3
    assert e_q < lim and e_t < lim, (e_q, e_t)
    assert dt1.abs().sum() > 0 or N == 1


@pytest.mark.parametrize("name", ["sca_small", "ragged", "three_row_blocks", "far_keys", "groups", "wide_table"])
def test_slab_bwd_q_is_as_accurate_as_the_query_tile_kernel_against_the_oracle(name):
    """the oracle's materialised attention in float64; bf16 operands on both kernels"""
    B, V, C, h, g, S, N, Wt, extra = CASES[name]
    ins = _problem(B, V, C, h, g, S, N, Wt, seed=3 + S, **extra)
    _, dq1, dt1, _, _ = _run(ins, h, g, V, _lib.PREC_BF16, slab=True)
    _, dq0, dt0, _, _ = _run(ins, h, g, V, _lib.PREC_BF16, slab=False)
    query, kv, pos, table = (t.clone().double().requires_grad_(True) for t in ins)
    c = C // h
    outs = []
    for p in range(B * V):
        q = query[p // V].reshape(h, c, S * S)
        kk = kv[p, :, :C].reshape(N, h, c).permute(1, 2, 0)
        vv = kv[p, :, C:].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[p * g:(p + 1) * g], table, S, S, g, c ** -0.5)
        outs.append(o.reshape(C, S * S).t())
    want = torch.stack(outs, 0)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(77)).double()
    want.backward(cot)
    e1 = (rel_err(dq1.cpu().double(), query.grad), rel_err(dt1.cpu().double(), table.grad))
    e0 = (rel_err(dq0.cpu().double(), query.grad), rel_err(dt0.cpu().double(), table.grad))
    print(f"[vs oracle {name}] slab dQ {e1[0]:.2e} d(table) {e1[1]:.2e} | tile dQ {e0[0]:.2e} d(table) {e0[1]:.2e}")
    for a, b in zip(e1, e0):
        assert a < 1.5e-2 and a < 1.5 * b + 1e-3, (e1, e0)
