"""The attention output's way into proj_out and the tall-and-thin linears around it.

bevr_merge_views_fwd / _bwd (csrc/merge.hip, ops.merge_views): the packed attention output -> the layout proj_out
contracts, with the merge of two key segments' softmax halves -- against the stock-op chain it replaces
(ops.unpack_out_views after logaddexp2 / exp2 / mul / add; reference model/SCA_deform_attn.py:415-420,
model/TSA_deform_attn.py:325-333), forward and every input's gradient.  float32 both ways: the limits are rounding only."""
import pytest
import torch

from bevrender_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(got, want):
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-30)


def _chain(O_r, L_r, O_c, L_c, S, c, views):
    if O_c is not None:
        L_t = torch.logaddexp2(L_r, L_c)
        O_r = torch.exp2(L_r - L_t)[..., None] * O_r + torch.exp2(L_c - L_t)[..., None] * O_c
    return ops.unpack_out_views(O_r, S, c, views)


@pytest.mark.parametrize("two", [False, True])
@pytest.mark.parametrize("B,V,h,S,c", [(2, 3, 2, 21, 32), (1, 1, 4, 40, 16), (3, 2, 1, 33, 8), (1, 6, 2, 64, 32)])
def test_merge_views_matches_the_stock_chain(B, V, h, S, c, two):
    gen = torch.Generator().manual_seed(S * 7 + c)
    Sp = 32 * ((S + 31) // 32)
    Mp = S * Sp

    def mk(*shape, scale=1.0):
        return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)
    O_r, O_c = mk(B * V, h, Mp, 32), mk(B * V, h, Mp, 32)
    L_r, L_c = mk(B * V, h, Mp, scale=6.0), mk(B * V, h, Mp, scale=6.0)       # weights from 2^-30 to 1 - 2^-30
    ins = (O_r, L_r, O_c, L_c) if two else (O_r,)
    got = ops.merge_views(O_r, S, c, V, *( (L_r, O_c, L_c) if two else ()))
    want = _chain(O_r, L_r if two else None, O_c if two else None, L_c if two else None, S, c, V)
    assert got.shape == want.shape == (B, S * S, V * h * c)
    assert rel_err(got, want) < 2e-6
    cot = torch.randn(want.shape, generator=gen).to(DEV)
    g_got = torch.autograd.grad(got, ins, cot)
    g_want = torch.autograd.grad(want, ins, cot)
    for name, a, b in zip(("dO_r", "dL_r", "dO_c", "dL_c") if two else ("dO_r",), g_got, g_want):
        assert a.shape == b.shape
        e = rel_err(a, b)
        print(f"[merge {B=} {V=} {h=} {S=} {c=} {two=}] {name} {e:.1e}")
        assert e < (2e-5 if name.startswith("dL") else 2e-6), (name, e)
        # the padding (rows i >= S, channels >= c) carries exact zeros
        if name.startswith("dO"):
            pad = a.reshape(B * V, h, S, Sp, 32)
            assert Sp == S or pad[:, :, :, S:, :].abs().max().item() == 0.0
            assert c == 32 or pad[..., c:].abs().max().item() == 0.0


def test_merge_views_ignores_what_the_padding_rows_hold():
    """rows i >= S of the packed layout may hold anything (the kernels leave -inf / stale values there)"""
    B, V, h, S, c = 1, 2, 2, 20, 32
    Sp, gen = 32, torch.Generator().manual_seed(5)
    O_r = torch.randn(B * V, h, S * Sp, 32, generator=gen).to(DEV)
    O_c = torch.randn(B * V, h, S * Sp, 32, generator=gen).to(DEV)
    L_r = torch.randn(B * V, h, S * Sp, generator=gen).to(DEV)
    L_c = torch.randn(B * V, h, S * Sp, generator=gen).to(DEV)
    clean = ops.merge_views(O_r, S, c, V, L_r, O_c, L_c)
    for t, bad in ((O_r, float("nan")), (O_c, float("inf")), (L_r, float("-inf")), (L_c, float("-inf"))):
        t.reshape(B * V, h, S, Sp, -1)[:, :, :, S:] = bad
    O_r.requires_grad_(True)
    L_r.requires_grad_(True)
    out = ops.merge_views(O_r, S, c, V, L_r, O_c, L_c)
    assert torch.equal(out, clean)
    out.sum().backward()
    assert torch.isfinite(O_r.grad).all() and torch.isfinite(L_r.grad).all()


@pytest.mark.parametrize("rows,K,N,bias", [(70000, 64, 256, True), (4096 * 17, 384, 64, True), (66001, 256, 64, False)])
def test_linear_rows_matches_f_linear(rows, K, N, bias):
    """ops.linear_rows: F.linear with the weight gradient's long contraction split into partial products"""
    gen = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(N, K, generator=gen) * 0.1).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=gen).to(DEV).requires_grad_(True) if bias else None
    ins = (x, w, b) if bias else (x, w)
    got = ops.linear_rows(x, w, b)
    want = torch.nn.functional.linear(x, w, b)
    assert torch.equal(got, want)
    cot = torch.randn(want.shape, generator=gen).to(DEV)
    for name, a, c in zip(("dx", "dw", "db"), torch.autograd.grad(got, ins, cot), torch.autograd.grad(want, ins, cot)):
        e = rel_err(a, c)
        print(f"[linear_rows {rows}x{K}->{N}] {name} {e:.1e}")
        assert e < 1e-5, (name, e)      # float32 sums of 70 000 terms in two orders


@pytest.mark.parametrize("prec_name", ["bf16", "f16"])
@pytest.mark.parametrize("with_lse", [False, True])
def test_attn_bwd_prep_matches_the_stock_chain(prec_name, with_lse):
    """bevr_attn_bwd_prep (csrc/attn_bwd_prep.hip): the rounded cotangent (rows + permuted transpose), delta and the two
    maxima, against the stock passes of ops._AttnCore.backward it replaces"""
    import ctypes as C
    from bevrender_amd import _lib
    prec = _lib.PREC_BF16 if prec_name == "bf16" else _lib.PREC_F16
    ed = torch.bfloat16 if prec_name == "bf16" else torch.float16
    gen = torch.Generator().manual_seed(11)
    n_prob, h, Mp = 3, 2, 7 * 32
    dO = (torch.randn(n_prob, h, Mp, 32, generator=gen) * 3.0).to(DEV)
    O = torch.randn(n_prob, h, Mp, 32, generator=gen).to(DEV)
    scale = torch.tensor(4.0, device=DEV) if prec_name == "f16" else None
    dLSE = torch.randn(n_prob, h, Mp, generator=gen).to(DEV) if with_lse else None
    LSE0 = torch.randn(n_prob, h, Mp, generator=gen).to(DEV)
    LSE0[0, 1, 5] = float("-inf")                                    # a row without finite LSE takes no dLSE
    sc = 4.0 if scale is not None else 1.0
    # the stock chain
    dOe_w = (dO * sc).to(ed)
    dOr = dOe_w.float()
    delta_w = (dOr * O).sum(-1)
    if with_lse:
        delta_w = delta_w - torch.where(torch.isfinite(LSE0), dLSE, torch.zeros_like(dLSE)) * ops.LOG2E * sc
    dOt_w = ops._perm_t(dOe_w)
    # the kernel
    dOe = torch.empty_like(dOe_w)
    dOt = torch.empty(n_prob, h, 32, Mp, device=DEV, dtype=ed)
    delta = torch.empty(n_prob, h, Mp, device=DEV)
    stats = torch.zeros(2, device=DEV)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = _lib.lib().bevr_attn_bwd_prep(p(dO), p(O), p(scale), p(dLSE), p(LSE0) if with_lse else None, p(dOe), p(dOt), p(delta),
                                       p(stats), n_prob * h, Mp, prec, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(dOe, dOe_w) and torch.equal(dOt, dOt_w)
    assert rel_err(delta, delta_w) < 2e-6
    assert abs(stats[0].item() - dOr.pow(2).sum(-1).max().item()) < 1e-5 * stats[0].item()
    assert abs(stats[1].item() - delta_w.abs().max().item()) < 1e-5 * stats[1].item()
    # argument contract
    assert _lib.lib().bevr_attn_bwd_prep(p(dO), p(O), None, None, None, p(dOe), p(dOt), p(delta), p(stats), n_prob * h, Mp,
                                         _lib.PREC_F32, None) == -3
    assert _lib.lib().bevr_attn_bwd_prep(p(dO), p(O), None, None, None, p(dOe), p(dOt), p(delta), p(stats), n_prob * h, Mp - 1,
                                         prec, None) == -2


@pytest.mark.parametrize("shape", [(2, 9, 7, 8), (1, 40, 40, 64), (2, 10, 13, 256), (1, 203, 200, 12)])
def test_dwconv_res_gelu_matches_the_stock_chain(shape):
    """ops.dwconv_res_gelu (bevr_dwconv_res_gelu): gelu(x + depthwise3x3(x) + bias), the middle of the layer MLPs
    (reference model/model_utils.py:51-59), against conv2d + add + F.gelu -- forward and all three gradients"""
    B, H, W, Cc = shape
    gen = torch.Generator().manual_seed(H * 3 + Cc)
    x = torch.randn(B, H, W, Cc, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(Cc, 1, 3, 3, generator=gen) * 0.3).to(DEV).requires_grad_(True)
    b = torch.randn(Cc, generator=gen).to(DEV).requires_grad_(True)
    got = ops.dwconv_res_gelu(x, w, b)
    xn = x.permute(0, 3, 1, 2)
    want = torch.nn.functional.gelu(xn + torch.nn.functional.conv2d(xn, w, b, padding=1, groups=Cc)).permute(0, 2, 3, 1)
    assert rel_err(got, want) < 2e-6
    cot = torch.randn(want.shape, generator=gen).to(DEV)
    for name, a, c in zip(("dx", "dw", "db"), torch.autograd.grad(got, (x, w, b), cot), torch.autograd.grad(want, (x, w, b), cot)):
        e = rel_err(a, c)
        print(f"[dwconv_res_gelu {shape}] {name} {e:.1e}")
        assert e < 2e-5, (name, e)


@pytest.mark.parametrize("B,V,h,S,c", [(2, 3, 2, 21, 32), (1, 2, 4, 40, 16), (1, 6, 2, 64, 32)])
def test_merge_tap_matches_the_stock_chain(B, V, h, S, c):
    """ops.merge_tap (bevr_merge_tap_fwd / _bwd): merge_views with O_c = Rn Vp + bv formed on the way, against
    matmul + add + the stock merge chain; forward and the gradients of all six inputs"""
    gen = torch.Generator().manual_seed(S + c)
    Sp = 32 * ((S + 31) // 32)
    Mp = S * Sp

    def mk(*shape, scale=1.0):
        return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)
    O_r, Rn = mk(B * V, h, Mp, 32), mk(B * V, h, Mp, 12)
    L_r, L_c = mk(B * V, h, Mp, scale=5.0), mk(B * V, h, Mp, scale=5.0)
    Vp0 = torch.randn(B * V, h, 12, 32, generator=gen)
    bv0 = torch.randn(h, 32, generator=gen)
    Vp0[..., c:] = 0.0                       # padded head channels are zero (ops.attention_core pads them)
    bv0[..., c:] = 0.0
    Vp, bv = Vp0.to(DEV).requires_grad_(True), bv0.to(DEV).requires_grad_(True)
    ins = (O_r, L_r, Rn, L_c, Vp, bv)
    got = ops.merge_tap(O_r, L_r, Rn, L_c, Vp, bv, S, c, V)
    want = _chain(O_r, L_r, torch.matmul(Rn, Vp) + bv[None, :, None, :], L_c, S, c, V)
    assert got.shape == want.shape and rel_err(got, want) < 3e-6
    cot = torch.randn(want.shape, generator=gen).to(DEV)
    g_got = torch.autograd.grad(got, ins, cot)
    g_want = torch.autograd.grad(want, ins, cot)
    for name, a, w_ in zip(("dO_r", "dL_r", "dRn", "dL_c", "dVp", "dbv"), g_got, g_want):
        if name in ("dVp", "dbv"):           # the stock chain's gradient reaches the padded channels through nothing: zero there too
            a, w_ = a[..., :c], w_[..., :c]
        e = rel_err(a, w_)
        print(f"[merge_tap {B=} {V=} {h=} {S=} {c=}] {name} {e:.1e}")
        assert e < 3e-5, (name, e)
