"""Seeded random sweep of the attention entry points against the CPU oracle (48 configurations by default, BEVR_SWEEP=n for
more: 400 pass on MI355X): ragged BEV sizes, key counts around the
64-key step and 384-key block boundaries, every supported head width, groups, several views, tables from narrow
to wider than any LDS region, key positions inside / outside / clustered -- forward and every gradient, both
precision modes.  Small cases (each well under a second on the GPU, a few seconds of oracle)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O
from test_gpu_fullsize import kink_distance
from test_gpu_ops import _oracle_core, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def draw(seed):
    r = np.random.RandomState(seed)
    h = int(r.choice([1, 2, 4]))
    c = int(r.choice([8, 16, 32]))
    g = int(r.choice([d for d in (1, 2) if h % d == 0]))
    V = int(r.choice([1, 1, 2, 3]))
    B = int(r.choice([1, 2]))
    S = int(r.choice([3, 5, 8, 13, 21, 31, 32, 33, 40]))
    D = int(r.choice([1, 2, 3, 5, 9]))
    N = int(r.choice([1, 7, 31, 32, 33, 63, 64, 65, 127, 200, 383, 384, 385, 450]))
    mode = int(r.choice([0, 1, 2, 3]))
    return B, V, h * c, h, g, S, D, N, mode


def make(cfg, seed):
    B, V, C, h, g, S, D, N, mode = cfg
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen)
    k = torch.randn(B * V, N, C, generator=gen)
    v = torch.randn(B * V, N, C, generator=gen)
    u = torch.rand(B * V * g, N, 2, generator=gen)
    if mode == 0:      # inside the grid
        pos = u * 2 - 1
    elif mode == 1:    # well outside too (clamped taps, zero bias)
        pos = (u * 2 - 1) * 1.6
    elif mode == 2:    # one tight cluster: every key in the same few table cells (pinned-key case)
        pos = -0.97 + 0.02 * u
    else:              # two distant clusters inside one step
        pos = torch.where(torch.arange(N)[None, :, None] % 2 == 0, -0.8 + 0.05 * u, 0.7 + 0.05 * u)
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    return query, k, v, pos, table


def table_gradient_terms(ins, cot, h, g, V):
    """sum over (query, key) of |w dS| per table cell (float64): the magnitude of the terms whose signed sum is the
    table gradient.  dS = P (dP - delta) restated from the oracle's materialised formulation; the weights w >= 0 are
    the bilinear taps, applied by differentiating the oracle's own bias lookup with |dS| as the cotangent."""
    query, k, v, pos, table = [t.detach() for t in ins]
    B, C, S, _ = query.shape
    c = C // h
    Bp, N, _ = k.shape
    M = S * S
    tab = table.clone().requires_grad_(True)
    q_grid = O.normalized_grid(S, S, torch.float64).reshape(1, M, 2)
    total = 0.0
    for bp in range(Bp):
        q = query[bp // V].reshape(h, c, M)
        kk = k[bp].reshape(N, h, c).permute(1, 2, 0)
        vv = v[bp].reshape(N, h, c).permute(1, 2, 0)
        disp = (q_grid.unsqueeze(2) - pos[bp * g:(bp + 1) * g].reshape(g, 1, N, 2)) * 0.5
        bias = F.grid_sample(tab.reshape(g, h // g, *tab.shape[-2:]), disp[..., (1, 0)], mode="bilinear",
                             align_corners=True).reshape(h, M, N)
        P = torch.softmax(torch.einsum("bcm,bcn->bmn", q, kk) * c ** -0.5 + bias.detach(), dim=2)
        dO = cot[bp].t().reshape(h, c, M)                                   # cot (Bp, M, C)
        dP = torch.einsum("bcm,bcn->bmn", dO, vv)
        dS = P * (dP - (P * dP).sum(2, keepdim=True))
        total = total + (bias * dS.abs()).sum()
    total.backward()
    return tab.grad


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("BEVR_SWEEP", "48")))))
def test_random_configuration(seed):
    cfg = draw(seed)
    B, V, C, h, g, S, D, N, mode = cfg
    query, k, v, pos, table = make(cfg, 1000 + seed)
    # float64 oracle: in float32 it lands on either side of the kinks of the piecewise-bilinear bias
    ins_cpu = [t.clone().double().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core(*ins_cpu, h, g, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(seed))
    want.backward(cot.double())
    for prec, lim_o, lim_g in ((_lib.PREC_F32, 2e-4, 5e-4), (_lib.PREC_BF16, 2.5e-2, 3e-2)):
        ins = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
        got = ops.attention_core(*ins, heads=h, groups=g, views=V, precision=prec)
        got.backward(cot.to(DEV))
        torch.cuda.synchronize()
        e = rel_err(got.detach().cpu().double(), want.detach())
        assert e < lim_o, f"cfg {cfg} prec {prec}: out {e:.3e}"
        for n, a, b in zip(("query", "k", "v", "pos", "table"), ins, ins_cpu):
            if N == 1 and n in ("query", "k", "pos", "table"):
                # one key: P = 1 and dS = P (dP - delta) = 0 analytically.  What the kernels return is the rounding of
                # dP against delta -- in BF16 mode dO enters dP rounded to bf16 and delta in f32, so |dS| ~ 2^-9 |dO||V|
                lim0 = 2e-3 if prec == _lib.PREC_F32 else 5e-2
                assert a.grad.abs().max().item() < lim0 and b.grad.abs().max().item() < 1e-9, (cfg, prec, n)
                continue
            if n == "pos":
                # kinks of the piecewise-bilinear bias: a key whose table coordinate comes within
                # float32 rounding of an integer for some query column takes the neighbouring cell's derivative for that
                # column; compare the other keys in the 2-norm
                clean = kink_distance(pos, S, 2 * S * D - 1) >= 1e-4
                dg, dw = a.grad.cpu().double()[clean], b.grad[clean]
                e = (dg - dw).norm().item() / max(dw.norm().item(), 2e-2 * max(dw.numel(), 1) ** 0.5)
                assert clean.float().mean().item() > 0.5, "kink neighbourhood too wide for this case"
                assert e < (5e-3 if prec == _lib.PREC_F32 else 6e-2), f"cfg {cfg} prec {prec}: grad pos 2-norm {e:.3e}"
                continue
            # a gradient that is analytically ~0 (one key: P = 1, dS = 0) is rounding noise of O(1) terms: floor the scale
            e = (a.grad.cpu().double() - b.grad).abs().max().item() / max(b.grad.abs().max().item(), 2e-2)
            lim = lim_g
            if n == "table" and mode == 2 and prec == _lib.PREC_BF16:
                # every key in the same few table cells: a cell's gradient is sum_{q,n} w dS[q, n] with sum_n dS[q, n] = 0,
                # i.e. what is left after cancellation, and the bf16 operands' rounding applies to the TERMS.  Each cell is
                # therefore held to the sum of the magnitudes of its own terms (float64, below): 2^-8 per term for the
                # roundings of P, dP and the weights, x 2 of slack -- or to the ordinary limit, whichever is wider.  A
                # scatter into the wrong cell or with the wrong weight breaks this for the cells it touches.
                terms = table_gradient_terms(ins_cpu, cot.double(), h, g, V)
                err = (a.grad.cpu().double() - b.grad).abs()
                bound = 2.0 * 2.0 ** -8 * terms + lim_g * max(b.grad.abs().max().item(), 2e-2)
                worst = (err / bound).max().item()
                assert worst < 1.0, f"cfg {cfg} prec {prec}: grad table {worst:.2f} x its term bound"
                continue
            assert e < lim, f"cfg {cfg} prec {prec}: grad {n} {e:.3e}"
