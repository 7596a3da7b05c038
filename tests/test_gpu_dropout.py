"""Attention dropout (reference model/SCA_deform_attn.py:155-156,402-409,420; model/TSA_deform_attn.py:90-91,323,336).

The kernels do not store a mask: the keep decision of a (problem-head, query, key) pair is a hash of (seed, ph, mq, n)
(csrc/bevr_common.h:bevr_drop_keep) evaluated alike in the forward and in both backward kernels.  The tests rebuild that
mask on the host (ops.dropout_keep_mask), hand it to the oracle's materialised attention as the multiplier nn.Dropout
applies (0 or 1 / (1 - p)) and compare forward and every gradient."""
import numpy as np
import pytest
import torch

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O
from test_gpu_ops import CORE_CFGS, GRAD_LIM, TOL, _core_problem, rel_err

DEV = "cuda"


def test_keep_mask_is_a_pure_function_with_the_requested_rate():
    thr = int(round(0.3 * 65536))
    a = ops.dropout_keep_mask(77, thr, 4, 12, 500)
    b = ops.dropout_keep_mask(77, thr, 4, 12, 500)
    c = ops.dropout_keep_mask(78, thr, 4, 12, 500)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.float().mean().item() - 0.7) < 5e-3
    # no structure along any axis: per problem-head, per query and per key the rate holds
    for dim in ((1, 2), (0, 2), (0, 1)):
        assert (a.float().mean(dim) - 0.7).abs().max().item() < 0.08


def _oracle_core_drop(query, k, v, pos, table, h, g, V, keep):
    B, C, S, _ = query.shape
    c = C // h
    Bp, N, _ = k.shape
    outs = []
    for bp in range(Bp):
        q = query[bp // V].reshape(h, c, S * S)
        kk = k[bp].reshape(N, h, c).permute(1, 2, 0)
        vv = v[bp].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[bp * g:(bp + 1) * g], table, S, S, g, c ** -0.5, keep=keep[bp * h:(bp + 1) * h])
        outs.append(o.reshape(C, S * S).t())
    return torch.stack(outs, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("cfg", [CORE_CFGS[1], CORE_CFGS[4], CORE_CFGS[5]])
def test_attention_core_with_dropout_matches_the_oracle_with_the_same_mask(cfg, prec):
    B, V, C, h, g, S, D, N = cfg
    p, seed = 0.3, 0x5eed1234
    thr = int(round(p * 65536))
    query, k, v, pos, table = _core_problem(B, V, C, h, g, S, D, N, seed=sum(cfg))
    keep = ops.dropout_keep_mask(seed, thr, B * V * h, S, N).to(torch.float32) * (65536.0 / (65536.0 - thr))
    ins_cpu = [t.clone().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core_drop(*ins_cpu, h, g, V, keep)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(cot)
    ins_gpu = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
    got = ops.attention_core(*ins_gpu, heads=h, groups=g, views=V, precision=prec, attn_drop=(p, seed))
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), **TOL[prec])
    for n, a, b in zip(["query", "k", "v", "pos", "table"], ins_gpu, ins_cpu):
        e = rel_err(a.grad.cpu(), b.grad)
        assert e < GRAD_LIM[prec], f"grad {n}: rel err {e:.3e}"
    # and the mask did something: without it the output differs
    plain = ops.attention_core(*[t.detach() for t in ins_gpu], heads=h, groups=g, views=V, precision=prec)
    assert rel_err(plain.cpu(), want.detach()) > 0.05


@pytest.mark.gpu
def test_modules_accept_dropout_rates_and_apply_them_in_training_mode_only():
    from bevrender_amd.model.SCA_deform_attn import SCADeformableAttention
    from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
    torch.manual_seed(3)
    B, C, h, S, D, Hi, Wi = 2, 64, 2, 12, 3, 8, 20
    mods = [TSADeformableAttention(S, C, h, 1, 1, 3, True, B, n_views=1, attn_drop_rate=0.2, proj_drop_rate=0.1,
                                   precision=_lib.PREC_F32).to(DEV),
            SCADeformableAttention(S, D, C, h, 1, 1, 3, True, B, n_views=2, attn_drop_rate=0.2, proj_drop_rate=0.1,
                                   precision=_lib.PREC_F32).to(DEV)]
    plain = [TSADeformableAttention(S, C, h, 1, 1, 3, True, B, n_views=1, precision=_lib.PREC_F32).to(DEV),
             SCADeformableAttention(S, D, C, h, 1, 1, 3, True, B, n_views=2, precision=_lib.PREC_F32).to(DEV)]
    q = torch.randn(B, C, S, S, device=DEV, requires_grad=True)
    prev = torch.randn(B, C, S, S, device=DEV)
    x = torch.randn(B, 2, C, Hi, Wi, device=DEV)
    ref = (torch.rand(1, 2, S // 2, S * D, 2, device=DEV) * 2.2 - 1.1).expand(B, -1, -1, -1, -1).contiguous()
    for m, pm, args in ((mods[0], plain[0], (prev, q, {}, False)), (mods[1], plain[1], (x, q, ref, {}, False))):
        with torch.no_grad():
            for t in m.parameters():
                t.copy_(torch.randn_like(t) * 0.1)
        pm.load_state_dict(m.state_dict())
        m.train()
        a, _ = m(*args)
        b, _ = m(*args)
        assert torch.isfinite(a).all() and not torch.allclose(a, b)          # a fresh mask per call
        a.square().mean().backward()
        assert torch.isfinite(q.grad).all() and m.rpe_table.grad.abs().sum() > 0
        m.eval()
        pm.eval()
        e, _ = m(*args)
        w, _ = pm(*args)
        assert torch.allclose(e, w, rtol=1e-5, atol=1e-6)                   # eval mode: dropout is the identity
        # training-mode output is an unbiased estimate of the plain one: the mean over many masks approaches it
        m.train()
        acc = torch.zeros_like(w)
        for _ in range(48):
            acc += m(*args)[0].detach()
        assert rel_err(acc / 48, w.detach()) < 0.35
