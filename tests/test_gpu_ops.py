"""Parity of the HIP kernels (through the C ABI) with the CPU oracle.  Needs a real MI355X."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevrender_amd import _lib, ops
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"

# tolerances: F32 = exact-f32 MFMA, differences are summation order + the algebraic bias reformulation;
# BF16 = bf16 operands (Q, K, V, P, dO, dS), f32 accumulation.
# F16 = fp16 operands (11-bit significand against bf16's 8): its limits sit between the two.
TOL = {_lib.PREC_F32: dict(rtol=2e-4, atol=2e-5), _lib.PREC_BF16: dict(rtol=3e-2, atol=1.5e-2),
       _lib.PREC_F16: dict(rtol=6e-3, atol=3e-3)}
TOL[_lib.PREC_BF16X3] = TOL[_lib.PREC_F32]   # the split-bf16 mode is held to the f32 limits everywhere
GRAD_LIM = {_lib.PREC_F32: 5e-4, _lib.PREC_BF16X3: 5e-4, _lib.PREC_BF16: 3e-2, _lib.PREC_F16: 6e-3}


def rel_err(got, want):
    return (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)


def _core_problem(B, V, C, h, g, S, D, N, seed, spread=1.1):
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen)
    k = torch.randn(B * V, N, C, generator=gen)
    v = torch.randn(B * V, N, C, generator=gen)
    pos = (torch.rand(B * V * g, N, 2, generator=gen) * 2 - 1) * spread
    pos[0, 0] = torch.tensor([-7.0, 9.0])
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    return query, k, v, pos, table


def _oracle_core(query, k, v, pos, table, h, g, V):
    B, C, S, _ = query.shape
    c = C // h
    Bp, N, _ = k.shape
    outs = []
    for bp in range(Bp):
        q = query[bp // V].reshape(h, c, S * S)
        kk = k[bp].reshape(N, h, c).permute(1, 2, 0)
        vv = v[bp].reshape(N, h, c).permute(1, 2, 0)
        o = O.attention_core(q, kk, vv, pos[bp * g:(bp + 1) * g], table, S, S, g, c ** -0.5)
        outs.append(o.reshape(C, S * S).t())
    return torch.stack(outs, 0)


CORE_CFGS = [
    # B, V, C, h, g, S, D, N
    (1, 1, 64, 2, 1, 8, 1, 64),
    (2, 1, 16, 2, 1, 8, 3, 96),
    (1, 2, 16, 4, 2, 6, 2, 50),
    (1, 1, 64, 2, 1, 10, 5, 70),
    (1, 3, 64, 2, 1, 34, 2, 300),     # two row blocks, ragged columns, several views
    # table 1599 columns wide, keys all over it: no step fits the LDS table windows (forward / query-side backward
    # fall back to global gathers and atomics) and no key block fits the key-side ring (gather kernel)
    (1, 1, 32, 1, 1, 8, 100, 128),
]


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("cfg", CORE_CFGS)
def test_attention_core_forward_backward(cfg, prec):
    B, V, C, h, g, S, D, N = cfg
    query, k, v, pos, table = _core_problem(B, V, C, h, g, S, D, N, seed=sum(cfg))
    ins_cpu = [t.clone().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core(*ins_cpu, h, g, V)
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    want.backward(cot)

    ins_gpu = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
    got = ops.attention_core(*ins_gpu, heads=h, groups=g, views=V, precision=prec)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), **TOL[prec])
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    names = ["query", "k", "v", "pos", "table"]
    lim = GRAD_LIM[prec]
    for n, a, b in zip(names, ins_gpu, ins_cpu):
        e = rel_err(a.grad.cpu(), b.grad)
        assert e < lim, f"grad {n}: rel err {e:.3e}"


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
def test_mid_step_region_move_and_all_padding_half(prec):
    """Targeted shape for the conditional barriers of the query-stationary kernels (attn_fwd.hip move_region,
    attn_bwd_q.hip `if (kh) __syncthreads()`): N = 96 keys = one full 64-key step + one step whose second 32-key half
    is ALL padding.  In the full step the two halves sit 150 table columns apart, so the whole-step box fits no LDS
    region but each half does: the second half forces a MID-STEP region move (flush, re-anchor, refill between two
    barriers).  Forward and every gradient against the oracle."""
    B, V, C, h, g, S, D, N = 1, 1, 64, 2, 1, 34, 4, 96
    gen = torch.Generator().manual_seed(77)
    query = torch.randn(B, C, S, S, generator=gen)
    k, v = torch.randn(B, N, C, generator=gen), torch.randn(B, N, C, generator=gen)
    Wt = 2 * S * D - 1
    # table column b = (1 - px)(Wt-1)/4: clusters at b ~ 20, b ~ 170 (150 columns apart), and b ~ 60 for the tail
    def cluster(b0, n):
        px = 1 - (b0 + 3 * torch.rand(n, generator=gen)) * 4 / (Wt - 1)
        py = 0.1 + 0.05 * torch.rand(n, generator=gen)
        return torch.stack((py, px), -1)
    pos = torch.cat((cluster(20.0, 32), cluster(170.0, 32), cluster(60.0, 32)), 0)[None]
    table = torch.randn(h, 2 * S - 1, Wt, generator=gen) * 0.3
    ins_cpu = [t.clone().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core(*ins_cpu, h, g, V)
    cot = torch.randn(want.shape, generator=gen)
    want.backward(cot)
    ins_gpu = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
    got = ops.attention_core(*ins_gpu, heads=h, groups=g, views=V, precision=prec)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), **TOL[prec])
    lim = 5e-4 if prec == _lib.PREC_F32 else 5e-2
    for n, a, b in zip(["query", "k", "v", "pos", "table"], ins_gpu, ins_cpu):
        e = rel_err(a.grad.cpu(), b.grad)
        assert e < lim, f"grad {n}: rel err {e:.3e}"


def test_attention_rows_are_a_convex_combination_at_scale():
    """Size-independent property at a BEV side the oracle cannot materialise (S=100, N=25000):
    V = const channel pattern -> output equals that pattern for every query (softmax rows sum to 1)."""
    B, V, C, h, S, D = 1, 1, 64, 2, 100, 5
    N = (S // 2) * S * D
    gen = torch.Generator().manual_seed(3)
    query = torch.randn(B, C, S, S, generator=gen).to(DEV)
    k = torch.randn(B, N, C, generator=gen).to(DEV)
    pat = torch.randn(C, generator=gen)
    v = pat[None, None, :].expand(B, N, C).contiguous().to(DEV)
    pos = (torch.rand(B, N, 2, generator=gen) * 2 - 1).to(DEV)
    table = (torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3).to(DEV)
    out = ops.attention_core(query, k, v, pos, table, heads=h, groups=1, views=1, precision=_lib.PREC_F32)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), pat[None, None, :].expand_as(out).numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 16, 6, 10, 96, 1), (2, 16, 6, 10, 96, 2), (1, 64, 16, 44, 1000, 1)])
def test_sample_features_matches_grid_sample(shape):
    B, C, Hi, Wi, N, g = shape
    gen = torch.Generator().manual_seed(5)
    feat = torch.randn(B, C, Hi, Wi, generator=gen)
    pos = (torch.rand(B * g, N, 2, generator=gen) * 2 - 1) * 1.2
    pos[0, :8] = torch.tensor([[-1., -1.], [1., 1.], [-1., 1.], [0., 0.], [1.5, 0.], [0., -1.5], [.999, .999], [3., 3.]])
    fc, pc = feat.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    want = F.grid_sample(fc.reshape(B * g, C // g, Hi, Wi), pc[:, None, :, (1, 0)], mode="bilinear",
                         align_corners=True).reshape(B, C, N).permute(0, 2, 1)
    cot = torch.randn(want.shape, generator=gen)
    want.backward(cot)
    fg, pg = feat.clone().to(DEV).requires_grad_(True), pos.clone().to(DEV).requires_grad_(True)
    got = ops.sample_features(fg, pg, g)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fg.grad.cpu().numpy(), fc.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pg.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol=2e-5)


def test_sample_backward_with_most_keys_pinned_to_the_corner():
    """The benchmark rig pins 66 % of a view's keys to pixel (0, 0) (+ the learned offset): the feature gradient of the
    top-left corner is summed per workgroup instead of scattered per key (csrc/sample.hip).  20 000 keys, 70 % of them
    within the offset range of the corner (some taps outside the map), the rest anywhere; C = 64 and a C whose
    channel-quad count does not divide the block (the plain-scatter variant)."""
    for C in (64, 24):
        B, Hi, Wi, N = 3, 16, 44, 20000
        gen = torch.Generator().manual_seed(C)
        feat = torch.randn(B, C, Hi, Wi, generator=gen)
        pos = (torch.rand(B, N, 2, generator=gen) * 2 - 1) * 1.1
        npin = int(0.7 * N)
        rng = torch.tensor([5.0 / 99.0, 5.0 / 999.0])
        pos[:, :npin] = -1.0 + torch.tanh(torch.randn(B, npin, 2, generator=gen)) * rng
        fc, pc = feat.clone().double().requires_grad_(True), pos.clone().double().requires_grad_(True)
        want = F.grid_sample(fc, pc[:, None, :, (1, 0)], mode="bilinear", align_corners=True).reshape(B, C, N).permute(0, 2, 1)
        cot = torch.randn(want.shape, generator=gen)
        want.backward(cot.double())
        fg, pg = feat.clone().to(DEV).requires_grad_(True), pos.clone().to(DEV).requires_grad_(True)
        got = ops.sample_features(fg, pg, 1)
        got.backward(cot.to(DEV))
        torch.cuda.synchronize()
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=5e-5)   # f32 vs f64
        # corner entries are sums of ~10 000 terms in float32 (order differs): relative to the largest entry
        scale = fc.grad.abs().max().item()
        assert (fg.grad.cpu().double() - fc.grad).abs().max().item() < 2e-5 * scale
        np.testing.assert_allclose(pg.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol=1e-4 * pc.grad.abs().max().item())


@pytest.mark.parametrize("name", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "proj_*.npz"))))
def test_projector_matches_reference_golden(name):
    from bevrender_amd.model.SCA import pillar_grid
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    z = np.load(os.path.join(GOLDEN, name))
    S, D, X, Y, Z, zs, iw, ih, ow, oh = z["cfg"]
    proj = BEV2CameraProjector(imu_to_rgb={0: list(z["imu_to_rgb"])}, K={0: [k.copy() for k in z["K"]]},
                               vehicle_type_code=0, img_width=int(iw), img_height=int(ih), ori_img_width=int(ow),
                               ori_img_height=int(oh), device=DEV)
    pts = pillar_grid({"X": X, "Y": Y, "Z": Z}, int(S), int(D), float(zs))
    got = torch.stack(proj.bev_grid_to_camera(pts)[0], 0).cpu().numpy()
    want = z["points_2d"]
    # the committed fixtures: the integer-truncation mask is reproduced bit for bit (no point flips its in-bound decision);
    # the boundary tolerance of _check_projection is for the random rigs of the sweep only
    _check_projection(got, want, pts, list(z["imu_to_rgb"]), list(z["K"]), int(iw), int(ih), int(ow), int(oh), name,
                      max_flip_frac=0.0)


def _check_projection(got, want, pts, T, K, iw, ih, ow, oh, tag, max_flip_frac=1e-3):
    """Every point agrees to 1e-5, except points whose in-bound decision FLIPPED: the mask is an integer truncation of
    a float pixel, so a pixel within float rounding of one of the four bounds may legitimately land on either side.
    The flips are counted, printed, and each one must be such a boundary case (pixel recomputed in float64)."""
    ncam, _, P = got.reshape(got.shape[0], 2, -1).shape
    g, w = got.reshape(ncam, 2, P), want.reshape(ncam, 2, P)
    pin_g, pin_w = (g == -1.0).all(1), (w == -1.0).all(1)
    flip = pin_g != pin_w
    close = np.isclose(g, w, rtol=1e-5, atol=1e-5).all(axis=1)
    assert close[~flip].all(), f"{tag}: {(~close[~flip]).sum()} non-boundary points differ"
    n_flip = int(flip.sum())
    print(f"\n[{tag}] {n_flip} of {flip.size} points flipped their in-bound decision")
    if n_flip:
        p64 = pts.reshape(4, -1).double().numpy()
        for cam in range(ncam):
            Kc = np.array(K[cam], dtype=np.float64)
            Kc[0] *= iw / ow
            Kc[1] *= ih / oh
            uvw = Kc[:, :3] @ (np.linalg.inv(np.asarray(T[cam], dtype=np.float64)) @ p64)[:3]
            u, v = uvw[0] / uvw[2], uvw[1] / uvw[2]
            for i in np.nonzero(flip[cam])[0]:
                d = min(abs(u[i]), abs(u[i] - (iw - 1)), abs(v[i]), abs(v[i] - (ih - 1)))
                assert d < 1e-3, f"{tag}: cam {cam} point {i} flipped at pixel ({u[i]:.5f}, {v[i]:.5f}), not a boundary case"
    assert n_flip <= max_flip_frac * flip.size, f"{tag}: {n_flip} in-bound decisions flipped"


def test_projector_grey_pixel_mask(tmp_path):
    """remove_ref_in_gray (model/bev_cmr_proj.py:114-122): points whose truncated pixel is (128, 128, 128) in the
    camera's reference image are pinned like out-of-bound points.  Reference images are PNG files, read with PIL."""
    from PIL import Image
    from bevrender_amd.model.SCA import pillar_grid
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    z = np.load(os.path.join(GOLDEN, "proj_ring3_s28.npz"))
    S, D, X, Y, Z, zs, iw, ih, ow, oh = z["cfg"]
    iw, ih, ow, oh = int(iw), int(ih), int(ow), int(oh)
    rng = np.random.default_rng(3)
    paths, refs = [], []
    for cam in range(3):
        img = rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)
        img[img == 128] = 127                                   # no accidental grey ...
        img[ih // 3: 2 * ih // 3, iw // 4: 3 * iw // 4] = 128    # ... one grey rectangle (the vehicle's own hood)
        img[0, 0] = 128 if cam == 1 else img[0, 0]               # pixel (0, 0): where already-masked points are read
        p = str(tmp_path / f"ref{cam}.png")
        Image.fromarray(img).save(p)
        paths.append(p)
        refs.append(torch.from_numpy(img).permute(2, 0, 1))
    T, K = list(z["imu_to_rgb"]), [k.copy() for k in z["K"]]
    proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: [k.copy() for k in K]}, vehicle_type_code=0, img_width=iw,
                               img_height=ih, ori_img_width=ow, ori_img_height=oh, remove_ref_in_gray=True,
                               bound_check_img_paths=paths, device=DEV)
    pts = pillar_grid({"X": X, "Y": Y, "Z": Z}, int(S), int(D), float(zs))
    got = torch.stack(proj.bev_grid_to_camera(pts)[0], 0).cpu().numpy()
    want = torch.stack(O.bev_grid_to_camera(pts, T, K, iw, ih, ow, oh, gray_ref=refs), 0).numpy()
    plain = z["points_2d"]
    newly = ((want == -1).all(1) & ~(plain == -1).all(1)).mean()
    assert 0.02 < newly < 0.9                                    # the rectangle really removes points
    _check_projection(got, want, pts, T, K, iw, ih, ow, oh, "grey mask")


@pytest.mark.parametrize("normalize", [False, True])
def test_pairwise_corr_and_recall(normalize):
    gen = torch.Generator().manual_seed(9)
    n, m, E = 8, 8, 64 * 28 * 28
    cam = torch.randn(n, E, generator=gen)
    mp = cam + 0.8 * torch.randn(m, E, generator=gen)
    cc, mc = cam.clone().double().requires_grad_(True), mp.clone().double().requires_grad_(True)
    a, b = (F.normalize(cc, dim=1), F.normalize(mc, dim=1)) if normalize else (cc, mc)
    want = O.pairwise_corr(a, b)
    cot = torch.randn(n, m, generator=gen)
    want.backward(cot.double())
    cg, mg = cam.clone().to(DEV).requires_grad_(True), mp.clone().to(DEV).requires_grad_(True)
    got = ops.pairwise_corr(cg, mg, normalize)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    scale = want.abs().max().item()
    assert (got.detach().cpu().double() - want.detach()).abs().max().item() < 2e-5 * max(scale, 1.0)
    assert rel_err(cg.grad.cpu().double(), cc.grad) < 1e-4
    assert rel_err(mg.grad.cpu().double(), mc.grad) < 1e-4


def test_recall_rank_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "recall.npz"))
    for tag in ("a", "b"):
        cam, mp = torch.tensor(z[f"cam_{tag}"]).float().to(DEV), torch.tensor(z[f"map_{tag}"]).float().to(DEV)
        D = ops.pairwise_corr(cam, mp, False)
        rank = ops.recall_rank(D).cpu().numpy()
        got = tuple(float((rank < i).mean() * 100.0) for i in (1, 5, 10))
        np.testing.assert_allclose(np.array(got), z[f"recall_{tag}"], atol=1e-9)


def test_device_recall_matches_reference_golden_and_numpy_at_validation_size():
    """f3: `retrieval.get_recall` / `RecallAccumulator` (embeddings stay in HBM, Gram as a GEMM, rank count on the
    device) against the reference's own `get_recall` goldens, and against the oracle's NumPy restatement on a
    validation-sized set (N = 1 500 rows of E = 4 096) in both the float64 and the float32 form."""
    from bevrender_amd.retrieval import RecallAccumulator, get_recall
    z = np.load(os.path.join(GOLDEN, "recall.npz"))
    for tag in ("a", "b"):
        cam, mp = torch.tensor(z[f"cam_{tag}"]).to(DEV), torch.tensor(z[f"map_{tag}"]).to(DEV)
        np.testing.assert_allclose(np.array(get_recall(cam, mp, exact=True)), z[f"recall_{tag}"], atol=1e-9)
        np.testing.assert_allclose(np.array(get_recall(cam.float(), mp.float(), exact=False)), z[f"recall_{tag}"], atol=1e-9)
    rng = np.random.default_rng(7)
    N, E, B = 1500, 4096, 100
    cam = rng.standard_normal((N, E)).astype(np.float32)
    mp = (cam + 20.0 * rng.standard_normal((N, E))).astype(np.float32)      # recall@1 ~ 45 %: a non-trivial ranking
    cam /= np.linalg.norm(cam, axis=1, keepdims=True)
    mp /= np.linalg.norm(mp, axis=1, keepdims=True)
    want = O.get_recall(cam.astype(np.float64), mp.astype(np.float64))
    assert 5 < want[0] < 95
    acc = RecallAccumulator(N, E, DEV)
    for i in range(N // B):
        acc.add(i, torch.tensor(cam[i * B:(i + 1) * B]).to(DEV), torch.tensor(mp[i * B:(i + 1) * B]).to(DEV))
    np.testing.assert_allclose(np.array(acc.recall(exact=True)), np.array(want), atol=1e-9)
    # float32 GEMM: a rank can flip only where two distances agree to ~1e-6: allow 3 of the 1 500 columns
    np.testing.assert_allclose(np.array(acc.recall(exact=False)), np.array(want), atol=100.0 * 3 / N)


def test_key_prep_workspace_matches_numpy():
    """bevr_attn_key_prep through the C ABI: per-key (row offset, fraction, clamped column, row relative to the half's
    first row) and the tap box of every 32-key half of a step, against a numpy restatement of the header's contract
    (clamp a to [-(Sp+1), Ht+1], b to [-(Wt/2+2), Wt+1]; padded keys excluded from the boxes)."""
    import ctypes as C
    S, Wt, N, P = 10, 99, 150, 3
    geom = ops.AttnGeom(n_prob=P, q_div=1, heads=2, groups=1, S=S, N=N, Wt=Wt, precision=_lib.PREC_BF16)
    d = geom.desc()
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    a = (torch.rand(P, geom.Np, generator=g) * 3 - 1) * 2 * S          # exercises both clamps
    b = (torch.rand(P, geom.Np, generator=g) * 3 - 1) * Wt
    nbytes = L.bevr_attn_key_ws_bytes(C.byref(d))
    assert nbytes == P * geom.Np * 16 + P * (geom.Np // 32) * 16 * (1 + 8)    # KeyW, half boxes, 8 group boxes per half
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    ad, bd = a.to(DEV).contiguous(), b.to(DEV).contiguous()
    rc = L.bevr_attn_key_prep(C.byref(d), C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()),
                              C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    raw = ws.cpu().numpy()
    kw = raw[:P * geom.Np * 16].view(np.int32).reshape(P, geom.Np, 4)
    box = raw[P * geom.Np * 16:P * geom.Np * 16 + P * (geom.Np // 32) * 16].view(np.int32).reshape(P, geom.Np // 32, 4)
    gbox = raw[P * geom.Np * 16 + P * (geom.Np // 32) * 16:].view(np.int32).reshape(P, geom.Np // 32, 8, 4)
    an = np.clip(a.numpy(), -(geom.Sp + 1), geom.Ht + 1)
    bn = np.clip(b.numpy(), -(Wt // 2 + 2), Wt + 1)
    A = np.floor(an).astype(np.int64)
    live = np.arange(geom.Np)[None, :] < N
    np.testing.assert_array_equal(kw[..., 0], ((A + geom.y_off) + geom.x_off * geom.Hp) * 8)
    np.testing.assert_allclose(kw[..., 1].view(np.float32), (an - A).astype(np.float32), atol=1e-6)
    for p in range(P):
        for h in range(geom.Np // 32):
            sl = slice(32 * h, 32 * h + 32)
            lv = live[0, sl]
            if not lv.any():
                assert box[p, h, 1] < box[p, h, 0]            # empty half: amax < amin
                continue
            amin, amax = A[p, sl][lv].min(), A[p, sl][lv].max()
            assert (box[p, h, 0], box[p, h, 1]) == (amin, amax)
            np.testing.assert_allclose(box[p, h, 2:].view(np.float32), [bn[p, sl][lv].min(), bn[p, sl][lv].max()], rtol=1e-6)
            np.testing.assert_array_equal(kw[p, sl, 3][lv] >> 3, A[p, sl][lv] - amin)
            gid = kw[p, sl, 3][lv] & 7
            if gbox[p, h, 0, 0] == 0x7ffffffe:                # no groups: the half is spread too far
                assert (gid == 0).all()
                continue
            for g in range(8):                                 # every group's box is the box of its keys
                m = gid == g
                if not m.any():
                    assert gbox[p, h, g, 1] < gbox[p, h, g, 0]
                    continue
                assert (gbox[p, h, g, 0], gbox[p, h, g, 1]) == (A[p, sl][lv][m].min(), A[p, sl][lv][m].max())
                assert A[p, sl][lv][m].max() - A[p, sl][lv][m].min() <= 31
                np.testing.assert_allclose(gbox[p, h, g, 2:].view(np.float32),
                                           [bn[p, sl][lv][m].min(), bn[p, sl][lv][m].max()], rtol=1e-6)
            np.testing.assert_allclose(kw[p, sl, 2].view(np.float32)[lv], bn[p, sl][lv].astype(np.float32), rtol=1e-6)


@pytest.mark.parametrize("cfg", [(2, 9, 7, 64, 1, 5, 5), (1, 6, 6, 16, 2, 3, 3), (2, 5, 8, 32, 1, 1, 2), (1, 4, 4, 48, 3, 1, 2),
                                 (3, 20, 20, 64, 1, 1, 2)])
def test_fused_offset_head_matches_the_stock_op_chain(cfg):
    """bevr_offset_head_fwd/bwd (depthwise 1x1 with multiplier -> LayerNorm -> GELU -> 1x1, fused per pixel) against
    the reference's op chain in float64: SCA form (Mx = Dout = D), TSA form (no expansion, 2 outputs), channel groups
    that share the head, fewer than 64 channels per group (masked lanes)."""
    B, H, W, Cc, g, mx, dout = cfg
    cg = Cc // g
    K = cg * mx
    gen = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, H, W, Cc, generator=gen)
    w0 = (torch.randn(K, generator=gen) * 0.7) if (mx > 1 or dout == mx) else None
    b0 = torch.randn(K, generator=gen) * 0.3 if w0 is not None else None
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=gen), 0.1 * torch.randn(K, generator=gen)
    W3 = torch.randn(dout, K, generator=gen) / K ** 0.5
    cot = torch.randn(B * g, H, W, dout, generator=gen)

    def ref(x, w0, b0, gamma, beta, W3):
        xg = x.reshape(B, H, W, g, cg).permute(0, 3, 1, 2, 4).reshape(B * g, H, W, cg)
        z = xg if w0 is None else (xg.unsqueeze(-1) * w0.view(cg, mx) + b0.view(cg, mx)).flatten(-2)
        y = F.gelu(F.layer_norm(z, (K,), gamma, beta, 1e-5))
        return y @ W3.t()
    pc = [None if t is None else t.clone().double().requires_grad_(True) for t in (x, w0, b0, gamma, beta, W3)]
    want = ref(*pc)
    want.backward(cot.double())
    pg = [None if t is None else t.clone().to(DEV).requires_grad_(True) for t in (x, w0, b0, gamma, beta, W3)]
    got = ops.offset_head(pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], groups=g)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
    for name, a, b in zip(("x", "w0", "b0", "gamma", "beta", "W3"), pg, pc):
        if a is None:
            continue
        assert rel_err(a.grad.cpu().double(), b.grad) < 2e-5, name


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 9, 7, 3), (1, 64, 40, 40, 3), (2, 5, 6, 11, 5), (1, 3, 4, 4, 1), (2, 256, 10, 13, 3),
                                   (1, 16, 7, 21, 5), (3, 320, 5, 9, 3), (2, 8, 203, 200, 3), (1, 4, 30, 256, 3), (1, 4, 9, 300, 3)])
def test_depthwise_conv_matches_torch(shape, nhwc):
    """bevr_dwconv_fwd / bwd_w (the EncoderLayer glue's depthwise convolutions) against F.conv2d(groups=C):
    forward, input gradient, weight and bias gradients; both layouts; ragged sizes; k = 1, 3, 5."""
    B, Cc, H, W, k = shape
    g = torch.Generator().manual_seed(B * 100 + Cc + k)
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Cc, 1, k, k, generator=g) * 0.3
    b = torch.randn(Cc, generator=g)
    cot = torch.randn(B, Cc, H, W, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    want = F.conv2d(xr, wr, br, padding=k // 2, groups=Cc)
    want.backward(cot)
    xg, wg, bg = (t.clone().to(DEV).requires_grad_(True) for t in (x, w, b))
    if nhwc:
        got = ops.depthwise_conv(xg.permute(0, 2, 3, 1).contiguous(), wg, bg, nhwc=True).permute(0, 3, 1, 2)
    else:
        got = ops.depthwise_conv(xg, wg, bg, nhwc=False)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(wg.grad.cpu().numpy(), wr.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), br.grad.numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 300, 2, 32), (2, 129, 2, 8), (1, 65, 5, 24)])
def test_pack_kv_split_format(shape):
    """BEVR_PREC_BF16X3: bevr_pack_kv writes the hi / lo bf16 planes in the fragment order (csrc/bevr_common.h) -- bit
    for bit what ops._split_rows / ops._split_perm_t make of the f32 packing, and hi + lo returns the values to 2^-16."""
    import ctypes as C
    Bp, N, h, c = shape
    Cc = h * c
    Np = 64 * ((N + 63) // 64)
    kv = torch.randn(Bp, N, 2 * Cc, generator=torch.Generator().manual_seed(N)).to(DEV)
    Kr = torch.full((Bp, h, Np, 32), 7.0, device=DEV)
    Vr, Kt = torch.full_like(Kr, 7.0), torch.full((Bp, h, 32, Np), 7.0, device=DEV)
    Vt = torch.full_like(Kt, 7.0)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert _lib.lib().bevr_pack_kv(p(kv), C.c_void_p(kv.data_ptr() + 4 * Cc), 2 * Cc, N, Bp, N, Np, h, c, _lib.PREC_BF16X3,
                                   p(Kr), p(Vr), p(Kt), p(Vt), st) == 0
    Kw, Vw = ops.pack_keys(kv[..., :Cc], h).contiguous(), ops.pack_keys(kv[..., Cc:], h).contiguous()
    as_bits = lambda t: t.view(torch.int32)
    assert torch.equal(as_bits(Kr), as_bits(ops._split_rows(Kw))) and torch.equal(as_bits(Vr), as_bits(ops._split_rows(Vw)))
    assert torch.equal(as_bits(Kt), as_bits(ops._split_perm_t(Kw))) and torch.equal(as_bits(Vt), as_bits(ops._split_perm_t(Vw)))
    assert (ops._unsplit_rows(Kr) - Kw).abs().max().item() <= 2.0 ** -16 * Kw.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
@pytest.mark.parametrize("shape", [(3, 300, 2, 32), (2, 129, 2, 8), (1, 64, 4, 16), (2, 70, 16, 32), (1, 65, 5, 24)])
def test_pack_kv_equals_the_stock_op_chain_and_unpack_is_its_adjoint(shape, prec):
    """csrc/pack.hip against the chain it replaces (ops.pack_keys -> dtype cast -> ops._perm_t): bit-identical in both
    element types, zero padding included; bevr_unpack_dkv returns exactly the rows the packing read."""
    import ctypes as C
    Bp, N, h, c = shape
    Cc = h * c
    Np = 64 * ((N + 63) // 64)
    g = torch.Generator().manual_seed(N)
    kv = torch.randn(Bp, N, 2 * Cc, generator=g).to(DEV)
    ed = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32
    Kr = torch.full((Bp, h, Np, 32), 7.0, device=DEV, dtype=ed)
    Vr, Kt, Vt = torch.full_like(Kr, 7.0), torch.full((Bp, h, 32, Np), 7.0, device=DEV, dtype=ed), None
    Vt = torch.full_like(Kt, 7.0)
    L = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.bevr_pack_kv(p(kv), C.c_void_p(kv.data_ptr() + 4 * Cc), 2 * Cc, N, Bp, N, Np, h, c, prec, p(Kr), p(Vr), p(Kt),
                        p(Vt), st)
    assert rc == 0
    k, v = kv[..., :Cc], kv[..., Cc:]
    Kw, Vw = ops.pack_keys(k, h).to(ed).contiguous(), ops.pack_keys(v, h).to(ed).contiguous()
    assert torch.equal(Kr, Kw) and torch.equal(Vr, Vw)
    assert torch.equal(Kt, ops._perm_t(Kw)) and torch.equal(Vt, ops._perm_t(Vw))
    # no transposed outputs requested
    Kr2, Vr2 = torch.empty_like(Kr), torch.empty_like(Vr)
    assert L.bevr_pack_kv(p(kv), C.c_void_p(kv.data_ptr() + 4 * Cc), 2 * Cc, N, Bp, N, Np, h, c, prec, p(Kr2), p(Vr2),
                          None, None, st) == 0
    assert torch.equal(Kr2, Kw) and torch.equal(Vr2, Vw)
    # adjoint on float gradients
    dK = torch.randn(Bp, h, Np, 32, generator=g).to(DEV)
    dV = torch.randn(Bp, h, Np, 32, generator=g).to(DEV)
    dkv = torch.full((Bp, N, 2 * Cc), 7.0, device=DEV)
    assert L.bevr_unpack_dkv(p(dK), p(dV), p(dkv), C.c_void_p(dkv.data_ptr() + 4 * Cc), 2 * Cc, N, Bp, N, Np, h, c, st) == 0
    want_k = dK[:, :, :N, :c].permute(0, 2, 1, 3).reshape(Bp, N, Cc)
    want_v = dV[:, :, :N, :c].permute(0, 2, 1, 3).reshape(Bp, N, Cc)
    assert torch.equal(dkv[..., :Cc], want_k) and torch.equal(dkv[..., Cc:], want_v)
    # a key SEGMENT of the rows (problem stride N rows, first row n0): what ops._AttnCore packs for each of its two
    # key segments without copying them out
    n0 = N // 3
    Ns = N - n0
    Nsp = 64 * ((Ns + 63) // 64)
    Ks = torch.full((Bp, h, Nsp, 32), 7.0, device=DEV, dtype=ed)
    Vs = torch.full_like(Ks, 7.0)
    base = kv.data_ptr() + n0 * 2 * Cc * 4
    assert L.bevr_pack_kv(C.c_void_p(base), C.c_void_p(base + 4 * Cc), 2 * Cc, N, Bp, Ns, Nsp, h, c, prec, p(Ks), p(Vs),
                          None, None, st) == 0
    seg = kv[:, n0:].contiguous()
    assert torch.equal(Ks, ops.pack_keys(seg[..., :Cc], h).to(ed)) and torch.equal(Vs, ops.pack_keys(seg[..., Cc:], h).to(ed))
    dks = torch.full((Bp, N, 2 * Cc), 7.0, device=DEV)
    dKs, dVs = torch.randn(Bp, h, Nsp, 32, generator=g).to(DEV), torch.randn(Bp, h, Nsp, 32, generator=g).to(DEV)
    sb = dks.data_ptr() + n0 * 2 * Cc * 4
    assert L.bevr_unpack_dkv(p(dKs), p(dVs), C.c_void_p(sb), C.c_void_p(sb + 4 * Cc), 2 * Cc, N, Bp, Ns, Nsp, h, c, st) == 0
    assert torch.equal(dks[:, n0:, :Cc], dKs[:, :, :Ns, :c].permute(0, 2, 1, 3).reshape(Bp, Ns, Cc))
    assert (dks[:, :n0] == 7.0).all()                       # rows outside the segment untouched
    # argument contract
    assert L.bevr_pack_kv(p(kv), p(kv), 2 * Cc, N, Bp, N, Np + 1, h, c, prec, p(Kr), p(Vr), None, None, st) == -2
    assert L.bevr_pack_kv(p(kv), p(kv), 2 * Cc, N - 1, Bp, N, Np, h, c, prec, p(Kr), p(Vr), None, None, st) == -2   # problem stride < N
    assert L.bevr_pack_kv(None, p(kv), 2 * Cc, N, Bp, N, Np, h, c, prec, p(Kr), p(Vr), None, None, st) == -1
    assert L.bevr_pack_kv(p(kv), p(kv), 2 * Cc, N, Bp, N, Np, h, c, 5, p(Kr), p(Vr), None, None, st) == -3


@pytest.mark.gpu
def test_sample_features_takes_channels_last_features_without_a_copy():
    """Backbone features staged channels-last (model/encoder.py backbone_features) are already the (B, Hi, Wi, C) rows
    the sampling kernel reads: same result as from an NCHW tensor, and the layout change is a view."""
    B, Cc, Hi, Wi, N = 2, 16, 6, 10, 50
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(B, Cc, Hi, Wi, generator=g).to(DEV)
    pos = (torch.rand(B, N, 2, generator=g) * 2.4 - 1.2).to(DEV)
    cl = feat.contiguous(memory_format=torch.channels_last)
    f = cl.reshape(B, 1, Cc, Hi, Wi).permute(0, 1, 3, 4, 2).reshape(B, Hi, Wi, Cc)
    assert f.is_contiguous() and f.data_ptr() == cl.data_ptr()
    assert torch.equal(ops.sample_features(cl, pos, 1), ops.sample_features(feat, pos, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 16, 6, 10, 50), (3, 64, 64, 176, 3000), (2, 24, 9, 7, 300)])
def test_sample_features_reads_bf16_features_as_they_are(shape):
    """bf16 backbone features (the bf16 configurations) are sampled without a float copy of the map: a bf16 value is an
    exact float, so the samples and the position gradient equal those of the float map bit for bit, and the map's
    gradient is the float one rounded to bf16.  Most keys pinned to the corner, as the projector leaves them."""
    B, Cc, Hi, Wi, N = shape
    g = torch.Generator().manual_seed(N)
    feat = torch.randn(B, Cc, Hi, Wi, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    pos = (torch.rand(B, N, 2, generator=g) * 2.4 - 1.2)
    pos[:, : N // 2] = -1.0 + 0.01 * torch.randn(B, N // 2, 2, generator=g)
    pos = pos.to(DEV)
    cot = torch.randn(B, N, Cc, generator=g).to(DEV)
    res = []
    for f in (feat, feat.float()):
        f = f.detach().requires_grad_(True)
        p = pos.clone().requires_grad_(True)
        out = ops.sample_features(f, p, 1)
        out.backward(cot)
        res.append((out.detach(), p.grad, f.grad))
    (ob, pb, fb), (of, pf, ff) = res
    assert ob.dtype == torch.float32 and fb.dtype == torch.bfloat16
    assert torch.equal(ob, of)
    # C / 4 a power of two: the position gradient is reduced in a fixed order; otherwise with float atomics
    c4 = Cc // 4
    assert torch.equal(pb, pf) if c4 & (c4 - 1) == 0 else torch.allclose(pb, pf, rtol=1e-4, atol=1e-4 * pf.abs().max().item())
    # float atomics in a different order on the two runs: compare at the rounding of the bf16 result
    assert (fb.float() - ff).abs().max().item() <= 2.0 ** -7 * ff.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("use_tanh", [True, False])
@pytest.mark.parametrize("cfg", [(2, 3, 1, 8, 3), (1, 6, 2, 12, 5), (3, 1, 1, 6, 1)])
def test_key_positions_sca_equals_the_stock_op_chain(cfg, use_tanh):
    """csrc/keypos.hip against the chain it replaces (model/SCA_deform_attn.py:248-277 restated with stock ops in
    float64: "(b g) d (h n) w -> (b g) n h (w d)", tanh * range or clamp, + reference, gather into the key order),
    forward and the gradient of the offsets."""
    B, V, g, S, D = cfg
    Hk, Wk = S // 2, S * D
    N = Hk * Wk
    gen = torch.Generator().manual_seed(S * D + V)
    off = (torch.randn(V, B * g, S, S, D, generator=gen) * 1.5)
    ref = torch.rand(V, N, 2, generator=gen) * 2.2 - 1.1
    order = torch.stack([torch.randperm(N, generator=gen) for _ in range(V)])
    sy, sx = 0.7 / (Hk - 1.0), 1.3 / (Wk - 1.0)
    cot = torch.randn(B, V, g, N, 2, generator=gen)

    o64 = off.double().requires_grad_(True)
    outs = []
    for v in range(V):
        o = o64[v].reshape(B * g, Hk, 2, S, D).permute(0, 2, 1, 3, 4).reshape(B * g, 2, Hk, Wk)
        if use_tanh:
            o = o.tanh() * torch.tensor([sy, sx], dtype=torch.float64).reshape(1, 2, 1, 1)
        p = o.permute(0, 2, 3, 1).reshape(B, g, N, 2) + ref[v].double()[None, None]
        if not use_tanh:
            p = p.clamp(-1.0, 1.0)
        outs.append(p)
    want = torch.stack(outs, 1).gather(3, order[None, :, None, :, None].expand(B, V, g, N, 2))
    (want * cot.double()).sum().backward()

    og = off.to(DEV).requires_grad_(True)
    got = ops.key_positions(og, ref.to(DEV), order.to(DEV), B, g, sca_SD=(S, D), use_tanh=use_tanh, sy=sy, sx=sx)
    (got * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(og.grad.cpu().numpy(), o64.grad.numpy(), rtol=1e-4, atol=1e-6)
    # without a key order
    got2 = ops.key_positions(og.detach(), ref.to(DEV), None, B, g, sca_SD=(S, D), use_tanh=use_tanh, sy=sy, sx=sx)
    np.testing.assert_allclose(got2.cpu().numpy(), torch.stack(outs, 1).detach().numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("use_tanh", [True, False])
def test_key_positions_tsa_equals_the_stock_op_chain(use_tanh):
    """The TSA form (model/TSA_deform_attn.py:170-196): (y, x) offsets per key-grid pixel, + the regular grid."""
    B, g, Hk, Wk = 2, 2, 7, 9
    N = Hk * Wk
    gen = torch.Generator().manual_seed(5)
    off = torch.randn(1, B * g, N, 2, generator=gen)
    grid = O.normalized_grid(Hk, Wk, torch.float32).reshape(1, N, 2)
    order = torch.randperm(N, generator=gen)[None]
    sy, sx = 2.0 / (Hk - 1.0), 2.0 / (Wk - 1.0)
    cot = torch.randn(B, 1, g, N, 2, generator=gen)
    o64 = off.double().requires_grad_(True)
    p = o64[0].tanh() * torch.tensor([sy, sx], dtype=torch.float64) if use_tanh else o64[0]
    p = p + grid.double()
    if not use_tanh:
        p = p.clamp(-1.0, 1.0)
    want = p.index_select(1, order[0]).reshape(B, 1, g, N, 2)
    (want * cot.double()).sum().backward()
    og = off.to(DEV).requires_grad_(True)
    got = ops.key_positions(og, grid.to(DEV), order.to(DEV), B, g, sca_SD=None, use_tanh=use_tanh, sy=sy, sx=sx)
    (got * cot.to(DEV)).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(og.grad.cpu().numpy(), o64.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [_lib.PREC_BF16, _lib.PREC_F16])
@pytest.mark.parametrize("feat_bf16", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 2, 12, 20, 300), (1, 64, 4, 9, 7, 129), (2, 32, 1, 6, 10, 64), (1, 48, 2, 8, 8, 70)])
def test_kv_project_equals_sample_gemm_pack(shape, feat_bf16, prec):
    """csrc/kvproj.hip (sampling -> proj_k | proj_v -> packed operands in one pass) against the chain it replaces:
    bevr_sample_fwd -> the GEMM in float64 on the E-rounded samples and weights -> rounding to E -> the layouts of
    bevr_pack_kv.  The kernel accumulates the same products in f32: agreement to the last place or two of E; padded keys
    and padded head channels exactly zero; a key segment (pointer offset + problem stride) as ops._AttnCore passes it."""
    import ctypes as C
    Bp, Cc, h, Hi, Wi, N = shape
    c = Cc // h
    ed = torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float16
    gen = torch.Generator().manual_seed(N + Cc)
    feat = torch.randn(Bp, Hi, Wi, Cc, generator=gen)
    feat = (feat.to(torch.bfloat16) if feat_bf16 else feat).to(DEV)
    pos = (torch.rand(Bp, N, 2, generator=gen) * 2.4 - 1.2).to(DEV)
    W = (torch.randn(2 * Cc, Cc, generator=gen) / Cc ** 0.5).to(DEV)
    bias = torch.randn(2 * Cc, generator=gen).to(DEV)
    n0 = 5                                                # a segment: keys [n0, N)
    Ns = N - n0
    Np = 64 * ((Ns + 63) // 64)
    mk = lambda *s: torch.full(s, 7.0, device=DEV, dtype=ed)
    Kr, Vr, Kt, Vt = mk(Bp, h, Np, 32), mk(Bp, h, Np, 32), mk(Bp, h, 32, Np), mk(Bp, h, 32, Np)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    We = W.to(ed).contiguous()
    vn2 = torch.zeros(1, device=DEV)
    kn2 = torch.zeros(Bp, h, device=DEV)
    rc = _lib.lib().bevr_kv_project(p(feat), int(feat_bf16), C.c_void_p(pos.data_ptr() + n0 * 8), N, p(We), p(bias), Bp, Hi,
                                    Wi, Cc, Ns, Np, h, c, prec, p(Kr), p(Vr), p(Kt), p(Vt), p(vn2), p(kn2), 1, st)
    assert rc == 0
    xs = ops._Sample.sample(feat, pos)[:, n0:]                              # float samples, the unfused kernel
    kv = (xs.to(ed).double() @ We.double().t() + bias.double()).float()     # products of E values, exact accumulation
    Kw = ops.pack_keys(kv[..., :Cc].contiguous(), h).to(ed)
    Vw = ops.pack_keys(kv[..., Cc:].contiguous(), h).to(ed)
    ulp = 2.0 ** -8 if prec == _lib.PREC_BF16 else 2.0 ** -11
    for got, want in ((Kr, Kw), (Vr, Vw), (Kt, ops._perm_t(Kw)), (Vt, ops._perm_t(Vw))):
        g, w = got.float(), want.float()
        assert (g - w).abs().max().item() <= 2.0 * ulp * max(w.abs().max().item(), 1.0)
        assert torch.equal(g == 0, w == 0) or ((g - w).abs()[(g == 0) != (w == 0)] < 1e-3).all()
    # padded keys: zeros, no bias (transposed layout: whole 32-blocks past the last key; inside a block the order is permuted
    # and the comparison with _perm_t above covers it)
    assert (Kr[:, :, Ns:] == 0).all() and (Vt[..., 32 * ((Ns + 31) // 32):] == 0).all()
    # the largest squared V row norm, for the backward's scale bound
    want_n2 = Vw.float().pow(2).sum(-1).max().item()
    assert abs(vn2.item() - want_n2) <= 0.02 * want_n2
    # the largest squared K row norm per (problem, head), for the forward's static softmax reference
    want_k2 = Kw.float().pow(2).sum(-1).amax(-1)
    assert ((kn2 - want_k2).abs() <= 0.02 * want_k2).all()
    if c < 32:
        assert (Kr[..., c:] == 0).all() and (Vt[:, :, c:] == 0).all()       # padded head channels
    # no transposed K requested (forward-only call)
    Kr2, Vr2, Vt2 = mk(Bp, h, Np, 32), mk(Bp, h, Np, 32), mk(Bp, h, 32, Np)
    assert _lib.lib().bevr_kv_project(p(feat), int(feat_bf16), C.c_void_p(pos.data_ptr() + n0 * 8), N, p(We), p(bias), Bp,
                                      Hi, Wi, Cc, Ns, Np, h, c, prec, p(Kr2), p(Vr2), None, p(Vt2), None, None, 1, st) == 0
    assert torch.equal(Kr2, Kr) and torch.equal(Vt2, Vt)
    # argument contract: f32-layout modes are refused
    assert _lib.lib().bevr_kv_project(p(feat), int(feat_bf16), p(pos), N, p(We), p(bias), Bp, Hi, Wi, Cc, Ns, Np, h, c,
                                      _lib.PREC_F32, p(Kr), p(Vr), None, p(Vt), None, None, 1, st) == -3


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [2, 4])
def test_fused_kv_source_with_channel_groups_matches_the_unfused_chain(groups):
    """n_groups > 1 on the fused path (reference model/SCA_deform_attn.py:219-255, 290-301 with g): group
    gi's channels are sampled at group gi's positions inside bevr_kv_project; forward and every gradient against
    sample_features -> F.linear -> attention_core(kv=...), bf16 operands."""
    B, V, Cc, h, S, D, Hi, Wi = 1, 2, 64, 4, 10, 3, 12, 20
    N = (S // 2) * S * D
    gen = torch.Generator().manual_seed(70 + groups)
    query = torch.randn(B, Cc, S, S, generator=gen)
    feat = torch.randn(B * V, Hi, Wi, Cc, generator=gen)
    pos = torch.rand(B * V * groups, N, 2, generator=gen) * 2.2 - 1.1
    W = torch.randn(2 * Cc, Cc, generator=gen) / Cc ** 0.5
    bias = torch.randn(2 * Cc, generator=gen) * 0.1
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    cot = torch.randn(B * V, S * S, Cc, generator=gen).to(DEV)
    assert ops.kv_source_supported(Cc, h, groups, _lib.PREC_BF16)
    res = []
    for fused in (True, False):
        ins = [t.clone().to(DEV).requires_grad_(True) for t in (query, feat, pos, W, bias, table)]
        q, f, p_, w, b_, t = ins
        if fused:
            out = ops.attention_core(q, None, None, p_, t, heads=h, groups=groups, views=V, precision=_lib.PREC_BF16,
                                     kv_source=(f, w, b_))
        else:
            xs = ops.sample_features(f.permute(0, 3, 1, 2), p_, groups)
            out = ops.attention_core(q, None, None, p_, t, heads=h, groups=groups, views=V, precision=_lib.PREC_BF16,
                                     kv=F.linear(xs, w, b_))
        out.backward(cot)
        res.append((out.detach(), [x.grad for x in ins]))
    (of, gf), (ou, gu) = res
    assert rel_err(of.cpu(), ou.cpu()) < 2e-2
    for n, a, b in zip(("query", "feat", "pos", "W", "bias", "table"), gf, gu):
        e = rel_err(a.cpu(), b.cpu())
        print(f"[fused kv groups={groups}] grad {n} rel diff {e:.2e}")
        assert e < 5e-2, f"grad {n}: {e:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [_lib.PREC_BF16, _lib.PREC_F16])
def test_attention_core_with_fused_kv_source_matches_the_unfused_chain(prec):
    """ops.attention_core(kv_source=...) against sample_features -> F.linear -> attention_core(kv=...): the same
    forward to the operands' rounding, and every gradient (feature map, sampling + bias positions, projection weights
    and bias, query, table) -- the fused path's adjoint is the unfused chain on recomputed samples."""
    B, V, Cc, h, S, D, Hi, Wi = 1, 2, 64, 2, 10, 3, 12, 20
    N = (S // 2) * S * D
    gen = torch.Generator().manual_seed(77)
    query = torch.randn(B, Cc, S, S, generator=gen)
    feat = torch.randn(B * V, Hi, Wi, Cc, generator=gen)
    pos = torch.rand(B * V, N, 2, generator=gen) * 2.2 - 1.1
    W = torch.randn(2 * Cc, Cc, generator=gen) / Cc ** 0.5
    bias = torch.randn(2 * Cc, generator=gen) * 0.1
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    cot = torch.randn(B * V, S * S, Cc, generator=gen).to(DEV)
    res = []
    for fused in (True, False):
        ins = [t.clone().to(DEV).requires_grad_(True) for t in (query, feat, pos, W, bias, table)]
        q, f, p_, w, b_, t = ins
        if fused:
            out = ops.attention_core(q, None, None, p_, t, heads=h, groups=1, views=V, precision=prec,
                                     kv_source=(f, w, b_), cell_split=N // 2)
        else:
            xs = ops._Sample.apply(f, p_)
            out = ops.attention_core(q, None, None, p_, t, heads=h, groups=1, views=V, precision=prec,
                                     kv=F.linear(xs, w, b_), cell_split=N // 2)
        out.backward(cot)
        res.append((out.detach(), [x.grad for x in ins]))
    (of, gf), (ou, gu) = res
    lim = 2e-2 if prec == _lib.PREC_BF16 else 3e-3
    assert rel_err(of.cpu(), ou.cpu()) < lim
    for n, a, b in zip(("query", "feat", "pos", "W", "bias", "table"), gf, gu):
        e = rel_err(a.cpu(), b.cpu())
        print(f"[fused kv prec={prec}] grad {n} rel diff {e:.2e}")
        assert e < 2.5 * lim, f"grad {n}: {e:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 7, 9, 64), (1, 40, 40, 64), (3, 5, 5, 8), (1, 13, 1, 256), (2, 3, 4, 4), (5, 1, 1, 128)])
def test_layer_norm_matches_torch(shape):
    """csrc/layernorm.hip against F.layer_norm in float64: forward, input gradient, d(gamma), d(beta); row counts that
    do not fill the last lane group / workgroup."""
    Cc = shape[-1]
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen) * 2.0 + 0.5
    gamma, beta = torch.randn(Cc, generator=gen), torch.randn(Cc, generator=gen)
    cot = torch.randn(*shape, generator=gen)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    want = F.layer_norm(xr, (Cc,), gr, br, 1e-5)
    want.backward(cot.double())
    xg, gg, bg = (t.clone().to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    got = ops.layer_norm(xg, gg, bg, 1e-5)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
    assert rel_err(xg.grad.cpu().double(), xr.grad) < 2e-5
    assert rel_err(gg.grad.cpu().double(), gr.grad) < 2e-5
    assert rel_err(bg.grad.cpu().double(), br.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("sign", [-1.0, 1.0])
@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
def test_attention_with_every_logit_far_from_zero(prec, sign):
    """Softmax is shift invariant, the kernels' first softmax reference must be too: every logit of every row near
    -170 (-245 in the kernels' log2 units: 2^245 overflows float) or near +170.  Found by the config-5 run (6 temporal
    frames: the history BEV grows, and with it K): the region forward scaled its still-zero accumulators by
    exp2(-first tile max) = inf and returned NaN rows.  Region kernels, cell kernels and the two-segment chain."""
    B, V, C, h, S, D, N = 1, 2, 64, 2, 12, 2, 200
    gen = torch.Generator().manual_seed(3)
    query = 2.0 + 0.1 * torch.randn(B, C, S, S, generator=gen)
    k = sign * (15.0 + 0.1 * torch.randn(B * V, N, C, generator=gen))
    v = torch.randn(B * V, N, C, generator=gen)
    pos = torch.rand(B * V, N, 2, generator=gen) * 1.6 - 0.8
    table = torch.randn(h, 2 * S - 1, 2 * S * D - 1, generator=gen) * 0.3
    ins_cpu = [t.clone().double().requires_grad_(True) for t in (query, k, v, pos, table)]
    want = _oracle_core(*ins_cpu, h, 1, V)
    cot = torch.randn(want.shape, generator=gen)
    want.backward(cot.double())
    lim = {_lib.PREC_F32: 2e-4, _lib.PREC_BF16X3: 2e-4, _lib.PREC_BF16: 6e-2, _lib.PREC_F16: 1e-2}[prec]
    for split in (None, 0, N // 2):
        ins = [t.clone().to(DEV).requires_grad_(True) for t in (query, k, v, pos, table)]
        got = ops.attention_core(*ins, heads=h, groups=1, views=V, precision=prec, cell_split=split)
        assert torch.isfinite(got).all(), f"split {split}: non-finite rows"
        e = rel_err(got.detach().cpu().double(), want.detach())
        assert e < lim, f"split {split}: rel err {e:.3e}"
        got.backward(cot.to(DEV))
        for n, a_, b_ in zip(("query", "k", "v", "table"), ins[:3] + ins[4:], ins_cpu[:3] + ins_cpu[4:]):
            assert torch.isfinite(a_.grad).all(), f"split {split}: grad {n} non-finite"
            eg = rel_err(a_.grad.cpu().double(), b_.grad)
            # logits of magnitude 245 carry 245 x the float rounding of logits of magnitude 1, and dQ multiplies by |K| = 15
            assert eg < 10 * lim, f"split {split}: grad {n} rel err {eg:.3e}"
