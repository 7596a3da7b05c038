"""The drop-in modules (reference class names / parameter names / forward signatures) on the HIP kernels
against the golden vectors produced by the reference's own modules.  Needs a real MI355X."""
import glob
import os

import numpy as np
import pytest
import torch

from bevrender_amd import _lib
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"
TSA = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "tsa_*.npz")))
SCA = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "sca_*.npz")))


LIM_BF16 = 3e-2   # gradients, bf16 operand mode, relative to the tensor's largest entry (observed worst on MI355X: 1.2e-2;
                  # f32 mode observed 8.2e-6 against its 1e-3)
WORST = {}


def teardown_module(module):
    print("\n[modules] worst gradient error / max |want| (gradients above the reference's noise floor):",
          {("f32" if k == _lib.PREC_F32 else "bf16"): f"{v:.3e}" for k, v in WORST.items()})


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def load_params(mod, z):
    sd = {k[len("param."):]: torch.tensor(v) for k, v in z.items() if k.startswith("param.")}
    missing, unexpected = mod.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def check(mod, z, out, inputs, prec):
    f32 = prec == _lib.PREC_F32
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=3e-4 if f32 else 3e-2,
                               atol=3e-5 if f32 else 2e-2)
    out.backward(torch.tensor(z["cot"]).to(DEV))
    torch.cuda.synchronize()
    lim = 2e-4 if f32 else LIM_BF16
    # absolute floor: some reference gradients are pure rounding noise (proj_k.bias shifts every logit of a
    # query equally, so its true gradient is 0)
    floor = (2e-5 if f32 else 2e-3) * max(np.abs(v).max() for k, v in z.items() if k.startswith("grad_"))
    for k, v in z.items():
        if k.startswith("grad_param."):
            g = dict(mod.named_parameters())[k[len("grad_param."):]].grad
            assert g is not None, k
            g = g.cpu().numpy()
        elif k.startswith("grad_in."):
            g = inputs[k[len("grad_in."):]].grad.cpu().numpy()
        else:
            continue
        err = np.abs(g - v).max() / (np.abs(v).max() + 1e-12)
        if np.abs(v).max() > 20 * floor:   # gradients that are not rounding noise in the reference itself
            WORST[prec] = max(WORST.get(prec, 0.0), err)
        assert np.abs(g - v).max() <= lim * np.abs(v).max() + floor, f"{k}: rel err {err:.3e}"


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
@pytest.mark.parametrize("name", TSA)
def test_tsa_module_matches_reference(name, prec):
    from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
    z = load(name)
    B, C, h, g, S, k, s, sor, xnone = [int(v) for v in z["cfg"]]
    m = TSADeformableAttention(bev_feat_shape=S, dim_embed=C, n_heads=h, n_groups=g, stride=s, kernel_size=k,
                               scale_offset_range=bool(sor), batch_size=B, n_views=1, precision=prec).to(DEV)
    load_params(m, z)
    query = torch.tensor(z["query"]).to(DEV).requires_grad_(True)
    prev = None if xnone else torch.tensor(z["prev_bev"]).to(DEV).requires_grad_(True)
    out, d = m(prev, query, {"k": 1}, False)
    assert d == {"k": 1}
    check(m, z, out, {"query": query, "prev_bev": prev}, prec)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
@pytest.mark.parametrize("name", SCA)
def test_sca_module_matches_reference(name, prec):
    from bevrender_amd.model.SCA_deform_attn import SCADeformableAttention
    z = load(name)
    B, C, h, g, S, D, Hi, Wi, sor = [int(v) for v in z["cfg"]]
    m = SCADeformableAttention(bev_feat_shape=S, bev_depth_dim=D, dim_embed=C, n_heads=h, n_groups=g, stride=1,
                               kernel_size=3, scale_offset_range=bool(sor), batch_size=B, n_views=1,
                               precision=prec).to(DEV)
    load_params(m, z)                       # the reference's own state_dict, unused m1/m2 heads included
    query = torch.tensor(z["query"]).to(DEV).requires_grad_(True)
    x = torch.tensor(z["x"]).to(DEV).requires_grad_(True)
    ref = torch.tensor(z["reference_points"]).to(DEV)
    out, _ = m(x, query, ref, None, False)
    check(m, z, out, {"query": query, "x": x}, prec)


def test_sca_multi_view_is_the_composition_of_single_views():
    """n_views > 1 has no runnable reference; pin it by composition: each view's
    contribution equals a single-view module fed that view, and proj_out mixes the concatenation."""
    from bevrender_amd.model.SCA_deform_attn import SCADeformableAttention
    from oracle import bevrender_oracle as O
    torch.manual_seed(4)
    B, V, C, h, S, D, Hi, Wi = 2, 3, 16, 2, 8, 3, 6, 10
    m = SCADeformableAttention(S, D, C, h, 1, 1, 3, True, B, n_views=V, precision=_lib.PREC_F32)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.randn_like(p) * (0.2 if p.ndim > 1 else 0.1))
    p_cpu = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, query = torch.randn(B, V, C, Hi, Wi), torch.randn(B, C, S, S)
    ref = (torch.rand(1, V, S // 2, S * D, 2) * 2.2 - 1.1).expand(B, -1, -1, -1, -1).contiguous()
    want = O.sca_forward(p_cpu, x, query, ref, n_heads=h, n_groups=1, depth_dim=D)
    m = m.to(DEV)
    got, _ = m(x.to(DEV), query.to(DEV), ref.to(DEV), {}, False)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.numpy(), rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("prec", [_lib.PREC_F32])
def test_encoder_layer_matches_reference(prec):
    """One full EncoderLayer (LPU, shared LN, TSA, MLP, LPU, SCA with the projected pillar grid, MLP), train
    mode, forward + all gradients, against the reference's own EncoderLayer (tests/golden/enclayer.npz)."""
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    from bevrender_amd.model.encoder import EncoderLayer
    z = load("enclayer.npz")
    B, C, S, D, h, X, Y, Z = [int(v) for v in z["cfg"]]
    proj = BEV2CameraProjector(imu_to_rgb={0: list(z["imu_to_rgb"])}, K={0: [k.copy() for k in z["K"]]},
                               vehicle_type_code=0, img_width=128, img_height=128, ori_img_width=128,
                               ori_img_height=128, device=DEV)
    layer = EncoderLayer(bev_bound={"X": X, "Y": Y, "Z": Z}, bev2cmr_projector=proj, n_views=1, bev_feat_shape=S,
                         bev_depth_dim=D, z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1, n_heads=h,
                         stride=1, kernel_size=3, batch_size=B, scale_offset_range=True, drop_path_rate=0.0,
                         precision=prec).to(DEV)
    load_params(layer, z)
    layer.train()
    bev_query = torch.tensor(z["bev_query"]).to(DEV).requires_grad_(True)
    prev_bev = torch.tensor(z["prev_bev"]).to(DEV).requires_grad_(True)
    img_feat = torch.tensor(z["img_feat"]).to(DEV).requires_grad_(True)
    out, _ = layer(bev_query, img_feat, prev_bev, torch.zeros(B, 2, 3, device=DEV), torch.tensor(0), {}, False)
    check(layer, z, out, {"bev_query": bev_query, "prev_bev": prev_bev, "img_feat": img_feat}, prec)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16, _lib.PREC_F16])
def test_encoder_layer_six_frame_recurrence_matches_the_oracle(prec):
    """BASELINE config 5 runs 6 temporal frames: the layer's output of frame t is TSA's history of frame t + 1
    (model/bevrender.py:203-219, model/encoder.py:366-379).  The golden layer chained over 6 frames in train mode (no
    ego-motion warp), the history amplified x2 per frame so that the sampled K / V grow as they do at the benchmark size
    with random-init weights (max |bev| 13 -> 1 125 over 6 frames; that growth drove the region forward's first softmax
    tile into NaN before its fix), against the oracle's encoder_layer_forward chained the same way in float64."""
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    from bevrender_amd.model.encoder import EncoderLayer
    z = load("enclayer.npz")
    B, C, S, D, h, X, Y, Z = [int(v) for v in z["cfg"]]
    proj = BEV2CameraProjector(imu_to_rgb={0: list(z["imu_to_rgb"])}, K={0: [k.copy() for k in z["K"]]},
                               vehicle_type_code=0, img_width=128, img_height=128, ori_img_width=128,
                               ori_img_height=128, device=DEV)
    layer = EncoderLayer(bev_bound={"X": X, "Y": Y, "Z": Z}, bev2cmr_projector=proj, n_views=1, bev_feat_shape=S,
                         bev_depth_dim=D, z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1, n_heads=h,
                         stride=1, kernel_size=3, batch_size=B, scale_offset_range=True, drop_path_rate=0.0,
                         precision=prec).to(DEV)
    load_params(layer, z)
    layer.train()
    p64 = {k[len("param."):]: torch.tensor(v).double() for k, v in z.items() if k.startswith("param.")}
    pts = O.sample_3d_points({"X": X, "Y": Y, "Z": Z}, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, list(z["imu_to_rgb"]), [k.copy() for k in z["K"]], 128, 128,
                                                      128, 128), B).double()
    q = torch.tensor(z["bev_query"])
    feat = torch.tensor(z["img_feat"])
    prev_g, prev_c = None, None
    gain = 2.0
    with torch.no_grad():
        for t in range(6):
            out_g, _ = layer(q.to(DEV), feat.to(DEV), prev_g, torch.zeros(B, 2, 3, device=DEV), torch.tensor(0), {}, False)
            out_c = O.encoder_layer_forward(p64, q.double(), feat.double(), prev_c, ref, n_heads=h, n_groups=1,
                                            depth_dim=D, n_views=1, kernel_size=3, stride=1)
            assert torch.isfinite(out_g).all(), f"frame {t}: non-finite"
            e = (out_g.cpu().double() - out_c).abs().max().item() / out_c.abs().max().item()
            # logits grow with the history (hundreds at |history| 1e3): a 16-bit operand's rounding is then a large ABSOLUTE
            # error in a logit and a sharp softmax turns it into O(0.1) of the output -- in the reference's own fp16 / bf16
            # too; observed 5e-2 (bf16) / 2e-2 (fp16) at |history| 1.6e3, 2e-5 (f32) / 4e-4 (split bf16)
            lim = {_lib.PREC_F32: 3e-4, _lib.PREC_BF16X3: 1.5e-3, _lib.PREC_BF16: 0.15, _lib.PREC_F16: 0.06}[prec]   # ~3x the observed
            print(f"[six frames prec={prec}] frame {t}: max |history| {0.0 if prev_c is None else prev_c.abs().max().item():.3e} max |out| {out_c.abs().max().item():.3e} rel err {e:.3e}")
            assert e < lim, f"frame {t}: rel err {e:.3e}"
            # both sides continue from the ORACLE's output: the limit is a per-frame limit at a growing magnitude (a free
            # running comparison measures how the amplified recurrence magnifies rounding, 2e-3 after 6 frames in f32)
            prev_c = out_c * gain
            prev_g = prev_c.float().to(DEV)
    assert prev_c.abs().max().item() > 5e2      # the history did reach the magnitude this test is about
