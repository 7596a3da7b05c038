"""CPU checks of the host logic: packed layouts + the kernels' algebraic reformulation of the RPE bias
(emulated in torch, float64) against the oracle's materialised formulation."""
import numpy as np
import pytest
import torch

from bevrender_amd import ops
from oracle import bevrender_oracle as O
from tests.kernel_emul import emul_attn_fwd


def _problem(B, V, C, h, g, S, D, N, seed, spread=1.2, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen, dtype=dtype)
    k = torch.randn(B * V, N, C, generator=gen, dtype=dtype)
    v = torch.randn(B * V, N, C, generator=gen, dtype=dtype)
    pos = (torch.rand(B * V * g, N, 2, generator=gen, dtype=dtype) * 2 - 1) * spread
    Wt = 2 * S * D - 1
    table = torch.randn(h, 2 * S - 1, Wt, generator=gen, dtype=dtype) * 0.3
    return query, k, v, pos, table


def _oracle_core(query, k, v, pos, table, h, g, V):
    B, C, S, _ = query.shape
    c = C // h
    outs = []
    Bp, N, _ = k.shape
    for bp in range(Bp):
        b = bp // V
        q = query[b:b + 1].reshape(h, c, S * S)
        kk = k[bp].reshape(N, h, c).permute(1, 2, 0)
        vv = v[bp].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[bp * g:(bp + 1) * g], table, S, S, g, c ** -0.5)  # (h, c, M)
        outs.append(o.reshape(C, S * S).t())
    return torch.stack(outs, 0)


@pytest.mark.parametrize("cfg", [
    # B, V, C, h, g, S, D, N
    (1, 1, 16, 2, 1, 8, 1, 64),      # TSA-like
    (2, 1, 16, 2, 1, 8, 3, 96),      # SCA fixture shape
    (1, 2, 16, 4, 2, 6, 2, 50),      # views + groups + ragged N, S not multiple of anything
    (1, 1, 64, 2, 1, 10, 5, 70),     # full head width 32, N not a multiple of 64
])
def test_packed_formulation_equals_reference_formulation(cfg):
    B, V, C, h, g, S, D, N = cfg
    query, k, v, pos, table = _problem(B, V, C, h, g, S, D, N, seed=sum(cfg))
    # a few keys far outside, a few exactly on the border
    pos[0, 0] = torch.tensor([-7.0, 9.0], dtype=pos.dtype)
    pos[0, 1] = torch.tensor([1.0, -1.0], dtype=pos.dtype)
    pos[0, 2] = torch.tensor([-1.0, 1.0], dtype=pos.dtype)
    want = _oracle_core(query, k, v, pos, table, h, g, V)

    geom = ops.AttnGeom(n_prob=B * V, q_div=V, heads=h, groups=g, S=S, N=N, Wt=table.shape[-1], precision=0)
    Qp = ops.pack_query(query, h)
    Kp, Vp = ops.pack_keys(k, h), ops.pack_keys(v, h)
    a, b = ops.key_coords(pos, S, geom.Wt, geom.Np)
    Tt = ops.pack_table(table, geom)
    assert Qp.shape == (B, h, geom.Mp, 32) and Kp.shape == (B * V, h, geom.Np, 32)
    assert Tt.shape == (h, geom.Wp, geom.Hp + 1)
    Op, _ = emul_attn_fwd(Qp, Kp, Vp, a, b, Tt, geom)
    got = ops.unpack_out(Op, S, C // h)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-9, atol=1e-10)


def test_geometry_matches_header_rules():
    g = ops.AttnGeom(n_prob=24, q_div=6, heads=2, groups=1, S=200, N=100000, Wt=1999, precision=1)
    assert (g.Sp, g.Mp, g.Np) == (224, 200 * 224, 100032)
    assert (g.Ht, g.Hp, g.y_off) == (399, 399 + 448 + 4, 226)
    assert (g.x_off, g.Wp) == (999 + 4, 1999 + 1998 + 9)
    assert g.Hp * g.Wp * 8 < 2 ** 31


def test_perm32_is_involution_and_matches_mfma_order():
    p = [ops.perm32(r) for r in range(32)]
    assert sorted(p) == list(range(32))
    assert [ops.perm32(x) for x in p] == list(range(32))
    # position 16 s + 8 hi + j must hold accumulator row crow(8 s + j, hi) = 16 s + 8 (j >> 2) + 4 hi + (j & 3)
    for s in range(2):
        for hi in range(2):
            for j in range(8):
                assert ops.perm32(16 * s + 8 * hi + j) == 16 * s + 8 * (j >> 2) + 4 * hi + (j & 3)
    idx = ops.perm_index(64, "cpu")
    assert idx[:32].tolist() == p and idx[32:].tolist() == [32 + x for x in p]


def test_pack_unpack_roundtrip():
    torch.manual_seed(0)
    S, h, c = 6, 2, 8
    x = torch.randn(3, h * c, S, S)
    Qp = ops.pack_query(x, h) / (c ** -0.5 * ops.LOG2E)
    back = ops.unpack_out(Qp, S, c)                         # (B, S*S, C) rows i*S + j
    np.testing.assert_allclose(back.numpy(), x.reshape(3, h * c, S * S).permute(0, 2, 1).numpy(), rtol=1e-6)


def test_kd_key_order_is_a_permutation_with_compact_leaves():
    rng = np.random.default_rng(0)
    S, D = 40, 5
    Wt = 2 * S * D - 1
    N = 5000 + 17
    pos = rng.uniform(-1, 1, size=(N, 2))
    pos[: N // 3] = -1.0                      # masked keys pinned to the corner, as the projector does
    order = ops.kd_key_order(pos, S, Wt)
    assert sorted(order.tolist()) == list(range(N))
    a = (1 - pos[order, 0]) * (S - 1) / 2
    b = (1 - pos[order, 1]) * (Wt - 1) / 4
    area_sorted, area_nat = [], []
    a0, b0 = (1 - pos[:, 0]) * (S - 1) / 2, (1 - pos[:, 1]) * (Wt - 1) / 4
    for k in range(0, N - 64, 64):
        area_sorted.append((np.ptp(a[k:k + 64]) + 1) * (np.ptp(b[k:k + 64]) + 1))
        area_nat.append((np.ptp(a0[k:k + 64]) + 1) * (np.ptp(b0[k:k + 64]) + 1))
    assert np.median(area_sorted) * 20 < np.median(area_nat)


def test_full_model_constructs_on_cpu_with_the_reference_state_dict_layout():
    """The G5 fixture was made by loading this model's state dict into the reference model with strict=True:
    452 entries, 1 990 436 parameters (tests/golden/make_golden_full.py).  Construction needs no GPU."""
    import importlib.util
    import logging
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(here, "golden", "make_golden_full.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    z = np.load(os.path.join(here, "golden", "full_bevrender.npz"))
    from bevrender_amd.model.bevrender import BEVRender
    torch.manual_seed(int(z["seed"]))
    model = BEVRender(g.full_config(), logging.getLogger("t"), "train")
    assert len(model.state_dict()) == int(z["n_state"]) == 452
    assert sum(p.numel() for p in model.parameters()) == 1990436


def test_render_decoder_matches_reference_golden():
    """G6: the render decoder (plain PyTorch / MIOpen module with the reference's parameter names) against the
    reference's decoder at BEV side 14 / 28 / 56 (tests/golden/decoder.npz), seeded weights as in G5."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(here, "golden", "make_golden_full.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    z = np.load(os.path.join(here, "golden", "decoder.npz"))
    from bevrender_amd.model.decoder_img_render import BEVImageRenderDecoder
    for S in (14, 28, 56):
        torch.manual_seed(int(z["seed"]) + S)
        dec = BEVImageRenderDecoder(bev_spatial_dim=S, model_dim=64, hid_dim=64)
        assert len(dec.state_dict()) == int(z[f"s{S}.n_state"])
        dec.train()
        out = dec(g.decoder_input(S)).detach()
        assert tuple(out.shape) == tuple(z[f"s{S}.shape"]) == (2, 3, 224, 224)
        np.testing.assert_allclose(out.flatten()[torch.tensor(z[f"s{S}.idx"])].numpy(), z[f"s{S}.val"], rtol=1e-4, atol=1e-5)
        assert abs(out.double().sum().item() - float(z[f"s{S}.sum"])) < 1e-3 * out.numel() * 1e-2
