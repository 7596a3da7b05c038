"""Parity of the cell-sorted attention kernels (csrc/attn_cell_*.hip, through the C ABI) with the CPU oracle.

The cell kernels compute the same attention as the region kernels (reference: model/SCA_deform_attn.py:331-413) with
the relative-position bias as a matrix product over 4 x 4 table chunks; they are fast for key tiles that crowd a few
table cells (what sorting the projector's pinned keys by cell gives) and must be CORRECT for any keys (tiles that do
not fit one chunk take the slow pass).  Cases: clustered keys sorted by cell (fast pass), scattered keys (slow pass
only), a mix inside one segment, the two-segment chain region kernels -> cell kernels, padded keys, one and several
row blocks.  Needs a real MI355X.
"""
import numpy as np
import pytest
import torch

from bevrender_amd import _lib, ops
from test_gpu_fullsize import check_dpos
from test_gpu_ops import TOL, _oracle_core, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def clustered_pos(P, n, S, Wt, gen, a0=None, b0=None, da=5.0, db=2.5):
    """n keys per problem around table coordinates (a0, b0) (default: where the projector pins out-of-image points,
    pos = (-1, -1)), spread over +-da table rows x +-db table columns: the learned-offset range of the reference."""
    a0 = float(S - 1) if a0 is None else a0
    b0 = (Wt - 1) / 2.0 if b0 is None else b0
    a = a0 + (torch.rand(P, n, generator=gen) * 2 - 1) * da
    b = b0 + (torch.rand(P, n, generator=gen) * 2 - 1) * db
    py = 1 - a * 2 / (S - 1)
    px = 1 - b * 4 / (Wt - 1)
    return torch.stack((py, px), -1)


def sort_by_cell(pos, S, Wt, n0=0):
    """cell-sort the keys [n0, N) of every problem (what the SCA module does per call)."""
    a, b = ops.key_coords(pos, S, Wt, pos.shape[1])
    order = ops.cell_order(a[:, n0:], b[:, n0:]) + n0
    idx = torch.cat((torch.arange(n0)[None].expand(pos.shape[0], -1), order), 1)
    return torch.gather(pos, 1, idx[..., None].expand(-1, -1, 2))


def run_case(query, k, v, pos, table, h, g, V, prec, split, lim_f32=5e-4, lim_bf16=3e-2):
    """float64 oracle: d(pos) is discontinuous where a table coordinate crosses an integer, and a float32 oracle lands on
    either side of such a kink by its own rounding (test_gpu_fullsize.check_dpos)."""
    ins_cpu = [t.clone().double().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core(*ins_cpu, h, g, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(cot.double())
    ins_gpu = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
    got = ops.attention_core(*ins_gpu, heads=h, groups=g, views=V, precision=prec, cell_split=split)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), **TOL[prec])
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    lim = {_lib.PREC_F32: lim_f32, _lib.PREC_BF16X3: lim_f32, _lib.PREC_BF16: lim_bf16, _lib.PREC_F16: 6e-3}[prec]
    errs = {}
    S, Wt = query.shape[-1], table.shape[-1]
    for n, a, b in zip(["query", "k", "v", "pos", "table"], ins_gpu, ins_cpu):
        if n == "pos":
            check_dpos(a.grad, b.grad, pos, S, Wt, {_lib.PREC_F32: 1e-3, _lib.PREC_BF16X3: 1e-3, _lib.PREC_BF16: 4e-2, _lib.PREC_F16: 8e-3}[prec],
                       f"cell prec={prec} split={split}")
            continue
        errs[n] = rel_err(a.grad.cpu().double(), b.grad)
    print(f"[cell prec={prec} split={split}] out {rel_err(got.detach().cpu().double(), want.detach()):.2e} "
          + " ".join(f"d{n} {e:.2e}" for n, e in errs.items()))
    for n, e in errs.items():
        assert e < lim, f"grad {n}: rel err {e:.3e}"
    return got.detach(), [t.grad.detach() for t in ins_gpu]


def problem(B, V, C, h, S, D, seed):
    gen = torch.Generator().manual_seed(seed)
    Wt = 2 * S * D - 1
    query = torch.randn(B, C, S, S, generator=gen)
    table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * 0.3
    return gen, Wt, query, table


# B, V, C, h, S, D, N
CLUSTER_CFGS = [
    (1, 1, 64, 2, 8, 3, 640),       # one row block (one wave per workgroup)
    (2, 1, 64, 2, 34, 3, 1500),     # two row blocks, N not a multiple of 64 (padded keys in the last step)
    (1, 2, 32, 1, 40, 5, 2048),     # two views sharing the query, one head
    (1, 1, 64, 2, 70, 2, 1024),     # three row blocks (three waves: the builder role rotates unevenly)
]


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("cfg", CLUSTER_CFGS)
def test_cell_sorted_cluster_all_keys_through_the_cell_kernels(cfg, prec):
    """Keys crowding ~70 table cells (the projector's pinned keys with learned offsets), sorted by cell: nearly every
    tile fits one chunk (fast pass).  Forward and every gradient against the oracle."""
    B, V, C, h, S, D, N = cfg
    gen, Wt, query, table = problem(B, V, C, h, S, D, seed=sum(cfg))
    k, v = torch.randn(B * V, N, C, generator=gen), torch.randn(B * V, N, C, generator=gen)
    pos = sort_by_cell(clustered_pos(B * V, N, S, Wt, gen), S, Wt)
    run_case(query, k, v, pos, table, h, 1, V, prec, split=0)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
def test_cell_kernels_are_correct_for_scattered_keys(prec):
    """Keys all over the table, unsorted: no tile fits a chunk, everything runs in the slow pass (per-pair gather).
    The cell entry points must be correct for ANY key set."""
    B, V, C, h, S, D, N = 1, 1, 64, 2, 34, 2, 200
    gen, Wt, query, table = problem(B, V, C, h, S, D, seed=5)
    k, v = torch.randn(B * V, N, C, generator=gen), torch.randn(B * V, N, C, generator=gen)
    pos = (torch.rand(B * V, N, 2, generator=gen) * 2 - 1) * 1.1
    pos[0, 0] = torch.tensor([-7.0, 9.0])      # far outside: clamped coordinates, zero bias
    run_case(query, k, v, pos, table, h, 1, V, prec, split=0)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
def test_cell_segment_mixing_fast_and_slow_tiles(prec):
    """One segment whose tiles alternate between a tight cluster (fast pass) and scattered keys (slow pass), and a
    cluster that straddles the table's lower edge (zero padding): both passes of all three kernels contribute to
    the same rows, chained in place."""
    B, V, C, h, S, D = 1, 1, 64, 2, 34, 3
    gen, Wt, query, table = problem(B, V, C, h, S, D, seed=11)
    parts = []
    for t in range(6):
        if t % 2 == 0:
            parts.append(clustered_pos(1, 32, S, Wt, gen, a0=10.0 + 7 * t, b0=40.0 + 20 * t, da=0.9, db=0.9))
        else:
            parts.append((torch.rand(1, 32, 2, generator=gen) * 2 - 1))
    parts.append(clustered_pos(1, 64, S, Wt, gen, a0=2 * S - 2.5, b0=Wt - 1.5, da=1.4, db=1.4))   # past the table's corner
    pos = torch.cat(parts, 1)
    N = pos.shape[1]
    k, v = torch.randn(1, N, C, generator=gen), torch.randn(1, N, C, generator=gen)
    run_case(query, k, v, pos, table, h, 1, V, prec, split=0)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("n_a", [64, 300])
def test_two_segment_chain_region_then_cell(prec, n_a):
    """The SCA split: scattered keys [0, n_a) through the region kernels, the cell-sorted cluster [n_a, N) through the
    cell kernels, one softmax over both (forward chained through (O, LSE), backward through the shared LSE / delta and
    accumulated dQ / d(table)) -- against the oracle, and against the same keys run through the region kernels alone."""
    B, V, C, h, S, D, N = 2, 2, 64, 2, 34, 3, 1100
    gen, Wt, query, table = problem(B, V, C, h, S, D, seed=23 + n_a)
    k, v = torch.randn(B * V, N, C, generator=gen), torch.randn(B * V, N, C, generator=gen)
    pos = torch.cat(((torch.rand(B * V, n_a, 2, generator=gen) * 2 - 1),
                     clustered_pos(B * V, N - n_a, S, Wt, gen)), 1)
    pos = sort_by_cell(pos, S, Wt, n0=n_a)
    out_split, g_split = run_case(query, k, v, pos, table, h, 1, V, prec, split=n_a)
    out_reg, g_reg = run_case(query, k, v, pos, table, h, 1, V, prec, split=None)
    lim = {_lib.PREC_F32: 2e-4, _lib.PREC_BF16X3: 2e-4, _lib.PREC_BF16: 4e-2, _lib.PREC_F16: 8e-3}[prec]
    assert rel_err(out_split, out_reg) < lim
    for n, a, b in zip(["query", "k", "v", "pos", "table"], g_split, g_reg):
        if n != "pos":   # d(pos): both were held to the float64 oracle away from kinks (check_dpos)
            assert rel_err(a, b) < {_lib.PREC_F32: 5e-4, _lib.PREC_BF16X3: 5e-4, _lib.PREC_BF16: 6e-2, _lib.PREC_F16: 1.2e-2}[prec], n


def test_cell_order_sorts_by_cell_and_is_a_permutation():
    gen = torch.Generator().manual_seed(0)
    a = torch.rand(3, 500, generator=gen) * 9 + 100
    b = torch.rand(3, 500, generator=gen) * 5 + 900
    order = ops.cell_order(a.to(DEV), b.to(DEV)).cpu()
    assert (order.sort(1).values == torch.arange(500)[None]).all()
    A = torch.floor(torch.gather(a, 1, order)).long()
    Bc = torch.floor(torch.gather(b, 1, order)).long()
    assert (A[:, 1:] >= A[:, :-1]).all()                      # rows ascending
    same_row = A[:, 1:] == A[:, :-1]
    step = (Bc[:, 1:] - Bc[:, :-1])[same_row]
    assert step.abs().max() <= 2                              # columns walk cell by cell inside a row (an empty cell: 2)


def test_cell_order_tail_holds_the_keys_of_the_least_populated_cells():
    """n_tail > 0: a permutation whose first n_tail keys are those of the sparsest cells (ties: earliest in cell order) and
    whose two parts are each in cell order -- checked against a plain restatement (stable sorts on the host)."""
    gen = torch.Generator().manual_seed(3)
    P, N, n_tail = 4, 3000, 96
    a = torch.randn(P, N, generator=gen) * 2.5 + 100          # clustered: populations from 1 to dozens per cell
    b = torch.randn(P, N, generator=gen) * 3.0 + 900
    got = ops.cell_order(a.to(DEV), b.to(DEV), n_tail).cpu()
    base = ops.cell_order(a.to(DEV), b.to(DEV), 0).cpu()
    assert (got.sort(1).values == torch.arange(N)[None]).all()
    for p in range(P):
        cid = torch.floor(a[p]).long() * 100000 + torch.floor(b[p]).long()
        s = cid[base[p]]
        _, inv, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
        pop = counts[inv]                                         # population of each key's cell, in cell order
        tail_pos = torch.sort(pop, stable=True).indices[:n_tail]  # the restatement: a stable sort by population
        is_tail = torch.zeros(N, dtype=torch.bool)
        is_tail[tail_pos] = True
        want = torch.cat((base[p][is_tail], base[p][~is_tail]))
        assert torch.equal(got[p], want), p
