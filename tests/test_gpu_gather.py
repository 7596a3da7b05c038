"""The gather forward (csrc/attn_gather_fwd.hip: scattered keys, bias as a sparse matrix product over an LDS table window,
bf16 operands) against the float64 oracle and against the region forward it replaces, on the cases its own machinery
adds: windows that fit, tiles emitted as per-key strips, tables shorter than a window column, the exact pass behind a
useless static bound, key counts around the 32-key emission, the benchmark's BEV size.  Reference arithmetic:
model/SCA_deform_attn.py:331-413 of the reference, restated in oracle/bevrender_oracle.py (attention_core)."""
import numpy as np
import pytest
import torch

from bevrender_amd import _lib, ops
from test_gpu_ops import _oracle_core, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
# bf16 operands against the float64 oracle (the limits of tests/test_gpu_ops.py for this mode)
LIM_OUT, LIM_GRAD = 2.5e-2, 3e-2


def _problem(B, V, C, h, S, D, N, seed, pos_fn):
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen)
    k = torch.randn(B * V, N, C, generator=gen)
    v = torch.randn(B * V, N, C, generator=gen)
    pos = pos_fn(torch.rand(B * V, N, 2, generator=gen))
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    return query, k, v, pos, table


def _run(ins, h, V, monkeypatch, gather):
    monkeypatch.setenv("BEVR_GATHER", "1" if gather else "0")
    dev = [t.clone().to(DEV).requires_grad_(True) for t in ins]
    ops.KERNEL_TIMER.start()
    out = ops.attention_core(*dev, heads=h, groups=1, views=V, precision=_lib.PREC_BF16)
    used = set(ops.KERNEL_TIMER.stop())
    assert ("bevr_attn_gather_fwd" in used) == gather and ("bevr_attn_fwd" in used) == (not gather), used
    return out, dev


CASES = [
    # B, V, C, h, S, D, N, key positions
    ("window fits, ragged last emission", (1, 2, 64, 2, 21, 3, 77), lambda u: (u * 2 - 1) * 0.5),
    ("strips: two clusters far apart in every tile", (1, 1, 64, 2, 40, 9, 200),
     lambda u: torch.where(torch.arange(u.shape[1])[None, :, None] % 2 == 0, -0.8 + 0.05 * u, 0.7 + 0.05 * u)),
    ("strips: keys all over a table wider than any window", (1, 1, 32, 1, 8, 100, 130), lambda u: (u * 2 - 1) * 1.1),
    ("outside the grid too (clamped taps)", (2, 1, 32, 1, 13, 2, 33), lambda u: (u * 2 - 1) * 1.6),
    ("one key", (1, 1, 64, 2, 8, 1, 1), lambda u: u * 2 - 1),
    ("31 / 32 / 33 keys around one emission", (1, 3, 16, 1, 5, 2, 33), lambda u: u * 2 - 1),
]


@pytest.mark.parametrize("name,cfg,pos_fn", CASES, ids=[c[0] for c in CASES])
def test_gather_forward_against_the_oracle_and_the_region_forward(name, cfg, pos_fn, monkeypatch):
    B, V, C, h, S, D, N = cfg
    ins = _problem(B, V, C, h, S, D, N, 500 + N, pos_fn)
    ins_cpu = [t.clone().double().requires_grad_(True) for t in ins]
    want = _oracle_core(*ins_cpu, h, 1, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(3))
    want.backward(cot.double())
    got, dev = _run(ins, h, V, monkeypatch, gather=True)
    ref, _ = _run(ins, h, V, monkeypatch, gather=False)
    torch.cuda.synchronize()
    assert rel_err(got.detach().cpu().double(), want.detach()) < LIM_OUT
    # same operands, same rounding points (16-bit table pairs, weights, P): the two forwards differ by summation order only
    assert rel_err(got.detach().cpu().double(), ref.detach().cpu().double()) < 1.5e-2
    got.backward(cot.to(DEV))      # the backward kernels run from the gather forward's O and LSE
    torch.cuda.synchronize()
    if N > 1:
        for n, a, b in zip(("query", "k", "v", "table"), (dev[0], dev[1], dev[2], dev[4]), (ins_cpu[0], ins_cpu[1], ins_cpu[2], ins_cpu[4])):
            e = (a.grad.cpu().double() - b.grad).abs().max().item() / max(b.grad.abs().max().item(), 2e-2)
            assert e < LIM_GRAD, f"{name}: grad {n} {e:.3e}"


def test_exact_pass_behind_a_useless_static_bound(monkeypatch):
    """The static softmax reference comes from ||Q|| max ||K|| (ops.py): make that bound hundreds of binades looser than
    the logits -- huge, mutually orthogonal Q and K -- so that every weight underflows against the reference, the columns
    are flagged and the exact instantiation (online maximum) recomputes them.  Forward against the oracle."""
    B, V, C, h, S, D, N = 1, 1, 32, 1, 12, 2, 150
    gen = torch.Generator().manual_seed(9)
    query = torch.zeros(B, C, S, S)
    query[:, :16] = torch.randn(B, 16, S, S, generator=gen) * 60.0          # channels 0..15 only
    k = torch.zeros(B * V, N, C)
    k[..., 16:] = torch.randn(B * V, N, 16, generator=gen) * 60.0           # channels 16..31 only: Q . K = 0
    k[..., :16] = torch.randn(B * V, N, 16, generator=gen) * 0.02           # + a small live part
    v = torch.randn(B * V, N, C, generator=gen)
    pos = (torch.rand(B * V, N, 2, generator=gen) * 2 - 1) * 0.9
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    ins = (query, k, v, pos, table)
    want = _oracle_core(*[t.double() for t in ins], h, 1, V)
    got, _ = _run(ins, h, V, monkeypatch, gather=True)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert rel_err(got.detach().cpu().double(), want) < LIM_OUT


def test_benchmark_bev_size_one_view(monkeypatch):
    """S = 200 (13 row blocks, 2 per wave, the last wave's second block past the column), D = 5: the launch shape of the
    benchmark, one view, 1 500 keys.  Gather against region forward on all rows, and the rows past the grid stay zero."""
    B, V, C, h, S, D, N = 1, 1, 64, 2, 200, 5, 1500
    ins = _problem(B, V, C, h, S, D, N, 77, lambda u: (u * 2 - 1) * 0.9)
    got, _ = _run(ins, h, V, monkeypatch, gather=True)
    ref, _ = _run(ins, h, V, monkeypatch, gather=False)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    e = rel_err(got.detach().cpu().double(), ref.detach().cpu().double())
    assert e < 1.5e-2, e
