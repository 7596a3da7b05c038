"""Pin the CPU oracle against golden vectors produced by the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import bevrender_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def params_of(z, dtype=torch.float32):
    return {k[len("param."):]: torch.tensor(v, dtype=dtype).requires_grad_(True)
            for k, v in z.items() if k.startswith("param.")}


def check_grads(z, p, inputs, rtol, atol):
    for k, v in z.items():
        if k.startswith("grad_param."):
            g = p[k[len("grad_param."):]].grad
            assert g is not None, k
            np.testing.assert_allclose(g.numpy(), v, rtol=rtol, atol=atol, err_msg=k)
        if k.startswith("grad_in."):
            g = inputs[k[len("grad_in."):]].grad
            np.testing.assert_allclose(g.numpy(), v, rtol=rtol, atol=atol, err_msg=k)


TSA = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "tsa_*.npz")))
SCA = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "sca_*.npz")))
PROJ = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "proj_*.npz")))


@pytest.mark.parametrize("name", TSA)
def test_tsa_matches_reference(name):
    z = load(name)
    B, C, h, g, S, k, s, sor, xnone = [int(v) for v in z["cfg"]]
    p = params_of(z)
    query = torch.tensor(z["query"]).requires_grad_(True)
    prev = None if xnone else torch.tensor(z["prev_bev"]).requires_grad_(True)
    out = O.tsa_forward(p, query, prev, n_heads=h, n_groups=g, kernel_size=k, stride=s,
                        scale_offset_range=bool(sor))
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=2e-5, atol=2e-6)
    out.backward(torch.tensor(z["cot"]))
    check_grads(z, p, {"query": query, "prev_bev": prev}, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("name", SCA)
def test_sca_matches_reference(name):
    z = load(name)
    B, C, h, g, S, D, Hi, Wi, sor = [int(v) for v in z["cfg"]]
    p = params_of(z)
    query = torch.tensor(z["query"]).requires_grad_(True)
    x = torch.tensor(z["x"]).requires_grad_(True)
    ref = torch.tensor(z["reference_points"])
    out = O.sca_forward(p, x, query, ref, n_heads=h, n_groups=g, depth_dim=D, scale_offset_range=bool(sor))
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=2e-5, atol=2e-6)
    out.backward(torch.tensor(z["cot"]))
    check_grads(z, p, {"query": query, "x": x}, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("name", PROJ)
def test_projector_matches_reference_bit_exact(name):
    z = load(name)
    S, D, X, Y, Z, zs, iw, ih, ow, oh = z["cfg"]
    S, D, iw, ih, ow, oh = int(S), int(D), int(iw), int(ih), int(ow), int(oh)
    pts = O.sample_3d_points({"X": X, "Y": Y, "Z": Z}, S, D, float(zs))
    assert pts.shape == (4, S // 2, S, D)
    np.testing.assert_array_equal(pts.numpy(), z["points_3d"])
    p2 = O.bev_grid_to_camera(pts, list(z["imu_to_rgb"]), list(z["K"]), iw, ih, ow, oh)
    got = torch.stack(p2, 0).numpy()
    np.testing.assert_array_equal(got, z["points_2d"])
    # the fixture exercises the mask: some points are pinned to (-1, -1)
    masked = np.all(z["points_2d"] == -1.0, axis=1).mean()
    assert 0.0 < masked < 1.0


def test_encoder_layer_matches_reference():
    z = load("enclayer.npz")
    B, C, S, D, h, X, Y, Z = [int(v) for v in z["cfg"]]
    p = params_of(z)
    bev_query = torch.tensor(z["bev_query"]).requires_grad_(True)
    prev_bev = torch.tensor(z["prev_bev"]).requires_grad_(True)
    img_feat = torch.tensor(z["img_feat"]).requires_grad_(True)
    pts = O.sample_3d_points({"X": X, "Y": Y, "Z": Z}, S, D, -1.0)
    p2 = O.bev_grid_to_camera(pts, list(z["imu_to_rgb"]), list(z["K"]), 128, 128, 128, 128)
    ref = O.sca_reference_points(p2, B)
    out = O.encoder_layer_forward(p, bev_query, img_feat, prev_bev, ref, n_heads=h, n_groups=1, depth_dim=D,
                                  n_views=1, kernel_size=3, stride=1)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=5e-5, atol=5e-6)
    out.backward(torch.tensor(z["cot"]))
    check_grads(z, p, {"bev_query": bev_query, "prev_bev": prev_bev, "img_feat": img_feat}, rtol=5e-4, atol=5e-5)


def test_recall_matches_reference():
    z = load("recall.npz")
    for tag in ("a", "b"):
        got = O.get_recall(z[f"cam_{tag}"], z[f"map_{tag}"])
        np.testing.assert_allclose(np.array(got), z[f"recall_{tag}"], rtol=0, atol=1e-12)
    assert 0 < z["recall_b"][0] < 100  # a non-trivial case


def test_retrieval_losses_basic_properties():
    """PARITY UNPINNED (pytorch_metric_learning absent): only sanity properties are asserted."""
    torch.manual_seed(0)
    cam = torch.randn(4, 32, dtype=torch.float64, requires_grad=True)
    mp = torch.randn(4, 32, dtype=torch.float64, requires_grad=True)
    for fn in (O.contrastive_loss, O.lifted_structure_loss):
        l = fn(cam, mp)
        assert l.ndim == 0 and l.item() >= 0
        l.backward()
        # scale invariance (embeddings are L2-normalised inside)
        assert abs(fn(cam * 3.0, mp * 0.5).item() - l.item()) < 1e-9
    # identical pairs -> zero positive distance
    l0 = O.contrastive_loss(cam.detach(), cam.detach())
    assert l0.item() >= 0
