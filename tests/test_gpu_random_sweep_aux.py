"""Seeded random sweeps of the other entry points against the stock PyTorch ops they replace (float64 on the CPU):
feature sampling (F.grid_sample), depthwise convolution (F.conv2d, groups = C), the ego-motion warp (oracle.tv_affine
twice) and the correlation head (pairwise correlation, margin losses, recall) against the oracle.  Small cases;
BEVR_SWEEP=n widens them."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevrender_amd import ops
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
NSEED = int(os.environ.get("BEVR_SWEEP", "24"))


def rel(got, want, floor=1e-3):
    return (got.double().cpu() - want.double()).abs().max().item() / max(want.abs().max().item(), floor)


@pytest.mark.parametrize("seed", list(range(NSEED)))
def test_sample_features_random(seed):
    r = np.random.RandomState(seed)
    g = int(r.choice([1, 1, 2, 4]))
    C = 4 * g * int(r.choice([1, 2, 3, 8, 16]))
    B = int(r.choice([1, 2, 5]))
    Hi, Wi = int(r.randint(2, 40)), int(r.randint(2, 70))
    N = int(r.choice([1, 3, 64, 257, 1000, 3001]))
    gen = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, C, Hi, Wi, generator=gen)
    pos = (torch.rand(B * g, N, 2, generator=gen) * 2 - 1) * float(r.choice([0.9, 1.0, 1.3]))
    share = int(r.choice([0, 0, N // 2]))
    pos[:, :share] = -1.0                                    # the projector's pin: pixel (0, 0)
    if N > 3:
        pos[0, -1] = torch.tensor([1.0, 1.0])                # the last pixel exactly
        pos[0, -2] = torch.tensor([-1.0, 1.0])
    cot = torch.randn(B, N, C, generator=gen)
    fr, pr = feat.double().requires_grad_(True), pos.double().requires_grad_(True)
    # reference: model/SCA_deform_attn.py:290-301 -- group gi's channels sampled at group gi's positions, grid in (x, y)
    want = F.grid_sample(fr.reshape(B * g, C // g, Hi, Wi), pr[:, None, :, [1, 0]], mode="bilinear",
                         padding_mode="zeros", align_corners=True)           # (B g, C/g, 1, N)
    want = want.reshape(B, g, C // g, N).permute(0, 3, 1, 2).reshape(B, N, C)
    want.backward(cot.double())
    fg, pg = feat.clone().to(DEV).requires_grad_(True), pos.clone().to(DEV).requires_grad_(True)
    got = ops.sample_features(fg, pg, g)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    assert rel(got.detach(), want.detach()) < 1e-5
    assert rel(fg.grad, fr.grad) < 1e-5
    # d(pos) is discontinuous where a coordinate sits exactly on a pixel centre (the pins, the corners): skip those keys
    px = (pos[..., 1] + 1) * (Wi - 1) / 2
    py = (pos[..., 0] + 1) * (Hi - 1) / 2
    smooth = ((px - px.round()).abs() > 1e-4) & ((py - py.round()).abs() > 1e-4)
    if smooth.any():
        assert rel(pg.grad.cpu()[smooth], pr.grad[smooth]) < 1e-4


@pytest.mark.parametrize("seed", list(range(NSEED)))
def test_depthwise_conv_random(seed):
    r = np.random.RandomState(100 + seed)
    B, Cc = int(r.choice([1, 2, 3])), int(r.choice([1, 3, 4, 8, 20, 64, 256]))
    H, W, k = int(r.randint(1, 30)), int(r.randint(1, 45)), int(r.choice([1, 3, 3, 5]))
    nhwc = bool(r.randint(0, 2))
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, H, W, generator=gen)
    w = torch.randn(Cc, 1, k, k, generator=gen) * 0.3
    b = torch.randn(Cc, generator=gen) if r.randint(0, 2) else None
    cot = torch.randn(B, Cc, H, W, generator=gen)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if b is not None else None
    want = F.conv2d(xr, wr, br, padding=k // 2, groups=Cc)
    want.backward(cot.double())
    xg, wg = x.clone().to(DEV).requires_grad_(True), w.clone().to(DEV).requires_grad_(True)
    bg = b.clone().to(DEV).requires_grad_(True) if b is not None else None
    if nhwc:
        got = ops.depthwise_conv(xg.permute(0, 2, 3, 1).contiguous(), wg, bg, nhwc=True).permute(0, 3, 1, 2)
    else:
        got = ops.depthwise_conv(xg, wg, bg, nhwc=False)
    got.backward(cot.to(DEV))
    torch.cuda.synchronize()
    assert rel(got.detach(), want.detach()) < 1e-5
    assert rel(xg.grad, xr.grad) < 1e-5
    assert rel(wg.grad, wr.grad, floor=1e-2) < 2e-4
    if b is not None:
        assert rel(bg.grad, br.grad, floor=1e-2) < 2e-4


@pytest.mark.parametrize("seed", list(range(max(NSEED // 2, 8))))
def test_history_warp_random(seed):
    r = np.random.RandomState(200 + seed)
    B, Cc, S = int(r.choice([1, 2, 3])), int(r.choice([1, 4, 7, 16])), int(r.choice([4, 7, 14, 28, 33]))
    gen = torch.Generator().manual_seed(seed)
    img = torch.randn(B, Cc, S, S, generator=gen)
    ang = (torch.rand(B, generator=gen) - 0.5) * float(r.choice([0.0, 20.0, 170.0]))       # degrees
    tr = (torch.rand(B, 2, generator=gen) - 0.5) * float(r.choice([0.0, 3.0, 2.0 * S]))    # pixels
    want = torch.stack([O.tv_affine(img[i].double(), float(ang[i]), (float(tr[i, 0]), float(tr[i, 1])), 0.0)
                        for i in range(B)], 0)
    got = ops.affine_warp(img.to(DEV), torch.deg2rad(ang).to(DEV), tr.to(DEV))
    torch.cuda.synchronize()
    assert rel(got, want) < 1e-4


@pytest.mark.parametrize("seed", list(range(max(NSEED // 2, 8))))
def test_retrieval_losses_and_corr_random(seed):
    """The correlation head at random batch sizes, embedding widths and noise levels: pairwise correlation (both
    normalisations) and the three margin losses, values and both gradients, against the oracle; recall against the
    oracle's NumPy form."""
    from bevrender_amd.loss.contrastive_loss import ContrastiveLoss
    from bevrender_amd.loss.lift_loss import LiftedStructureLoss
    from bevrender_amd.loss.triplet_loss_metric import TripletLossMetricLearning
    from bevrender_amd.retrieval import get_recall
    r = np.random.RandomState(300 + seed)
    B = int(r.choice([2, 3, 5, 8, 13, 16, 32]))
    E = int(r.choice([3, 7, 64, 1000, 4097, 50176]))   # (E = 1: a normalised scalar has no gradient at all)
    noise = float(r.choice([0.1, 1.0, 5.0, 30.0]))
    gen = torch.Generator().manual_seed(seed)
    cam = torch.randn(B, E, generator=gen)
    mp = cam + noise * torch.linspace(0.2, 1.5, B)[:, None] * torch.randn(B, E, generator=gen)
    for normalize in (False, True):
        cc, mc = cam.double().requires_grad_(True), mp.double().requires_grad_(True)
        a, b = (F.normalize(cc, dim=1), F.normalize(mc, dim=1)) if normalize else (cc, mc)
        want = O.pairwise_corr(a, b)
        cot = torch.randn(B, B, generator=gen)
        want.backward(cot.double())
        cg, mg = cam.clone().to(DEV).requires_grad_(True), mp.clone().to(DEV).requires_grad_(True)
        got = ops.pairwise_corr(cg, mg, normalize)
        got.backward(cot.to(DEV))
        torch.cuda.synchronize()
        assert (got.detach().cpu().double() - want.detach()).abs().max().item() < 3e-5 * max(want.abs().max().item(), 1.0)
        assert rel(cg.grad, cc.grad, floor=1e-6) < 2e-4 and rel(mg.grad, mc.grad, floor=1e-6) < 2e-4
    for mod, fn in ((ContrastiveLoss(), O.contrastive_loss), (LiftedStructureLoss(), O.lifted_structure_loss),
                    (TripletLossMetricLearning(), O.triplet_margin_loss)):
        cc, mc = cam.double().requires_grad_(True), mp.double().requires_grad_(True)
        want = fn(cc, mc)
        want.backward()
        cg, mg = cam.clone().to(DEV).requires_grad_(True), mp.clone().to(DEV).requires_grad_(True)
        got = mod.get_loss(cg, mg)
        got.backward()
        torch.cuda.synchronize()
        name = f"{type(mod).__name__} B{B} E{E} noise{noise}"
        assert abs(got.item() - want.item()) < 5e-5 * max(1.0, abs(want.item())), f"{name}: {got.item()} vs {want.item()}"
        for x, y in ((cg, cc), (mg, mc)):
            gy = y.grad if y.grad is not None else torch.zeros_like(y)
            gx = x.grad if x.grad is not None else torch.zeros_like(x)
            assert rel(gx, gy, floor=1e-4 * max(gy.abs().max().item(), 1e-6) + 1e-9) < 2e-3 or \
                (gx.double().cpu() - gy).abs().max().item() < 1e-7, name
    cn, mn = F.normalize(cam, dim=1), F.normalize(mp, dim=1)
    want = O.get_recall(cn.double().numpy(), mn.double().numpy())
    got = get_recall(cn.to(DEV), mn.to(DEV), exact=True)
    np.testing.assert_allclose(np.array(got), np.array(want), atol=1e-9)


@pytest.mark.parametrize("seed", list(range(max(NSEED // 2, 8))))
def test_projector_random_rig(seed):
    """BEV pillar grid -> camera pixels (model/bev_cmr_proj.py:61-124) for random rigs: 1-6 cameras at random yaw, pitch,
    roll, height and lateral offset, random focal lengths and principal points, image sizes and rescale ratios, BEV
    bounds, S, D, z shift -- against the oracle, with the boundary-flip accounting of test_gpu_ops._check_projection."""
    import math
    from bevrender_amd.model.SCA import pillar_grid
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    from test_gpu_ops import _check_projection
    r = np.random.RandomState(800 + seed)
    V = int(r.randint(1, 7))
    S, D = int(r.choice([4, 8, 14, 28, 50])), int(r.choice([1, 2, 3, 5]))
    ow, oh = int(r.choice([128, 704, 1408])), int(r.choice([128, 256, 512]))
    iw, ih = int(ow // r.choice([1, 2, 4])), int(oh // r.choice([1, 2]))
    bound = {"X": float(r.choice([10, 20, 50])), "Y": float(r.choice([10, 50])), "Z": float(r.choice([2, 4]))}
    zs = float(r.choice([-1.0, 0.0, 0.5]))
    R0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    T, K = [], []
    for v in range(V):
        yaw, pitch, roll = r.uniform(-math.pi, math.pi), r.uniform(-0.3, 0.3), r.uniform(-0.1, 0.1)
        cz, sz, cy, sy, cx, sx = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        M = np.eye(4)
        M[:3, :3] = Rz @ Ry @ Rx @ R0
        M[:3, 3] = (r.uniform(-1, 1), r.uniform(-1, 1), r.uniform(0.5, 3.0))
        T.append(M)
        f = r.uniform(0.4, 1.5) * ow
        K.append(np.array([[f, 0, ow * r.uniform(0.4, 0.6), 0], [0, f * r.uniform(0.9, 1.1), oh * r.uniform(0.4, 0.6), 0],
                           [0, 0, 1, 0]]))
    proj = BEV2CameraProjector(imu_to_rgb={0: [t.copy() for t in T]}, K={0: [k.copy() for k in K]}, vehicle_type_code=0,
                               img_width=iw, img_height=ih, ori_img_width=ow, ori_img_height=oh, device=DEV)
    pts = pillar_grid(bound, S, D, zs)
    got = torch.stack(proj.bev_grid_to_camera(pts)[0], 0).cpu().numpy()
    want = torch.stack(O.bev_grid_to_camera(O.sample_3d_points(bound, S, D, zs), T, K, iw, ih, ow, oh), 0).numpy()
    _check_projection(got, want, pts, T, K, iw, ih, ow, oh, f"rig seed {seed} V{V} S{S} D{D} {iw}x{ih}")
