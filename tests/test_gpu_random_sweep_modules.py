"""Seeded random sweep of the TSA / SCA modules (offset heads, sampling, K|V projection and packing, attention,
output projection -- the whole drop-in forward and its autograd) against the oracle's restatement of the reference
modules, with the module's own randomly initialised state_dict: random BEV sizes, widths, heads, groups, strides,
kernel sizes, depth bins, views, feature-map sizes.  Both precision modes; BEVR_SWEEP=n widens it."""
import os

import numpy as np
import pytest
import torch

from bevrender_amd import _lib
from oracle import bevrender_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
NSEED = int(os.environ.get("BEVR_SWEEP", "12"))


def randomize_(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in sorted(m.named_parameters()):
            scale = 0.3 if p.dim() > 1 else 0.1
            p.copy_((torch.randn(p.shape, generator=g) * scale + (1.0 if n.endswith("norm.weight") or ".1.norm.weight" in n else 0.0)).to(p.device))


def compare(m, out, want, ins_gpu, ins_cpu, params_cpu, tag, loose=False, bf16=False):
    def rel(a, b, floor=1e-3):
        return (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), floor)
    assert rel(out.detach(), want.detach()) < (3e-2 if bf16 else 3e-4), f"{tag}: out {rel(out.detach(), want.detach()):.3e}"
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(5)).double()
    want.backward(cot)
    out.backward(cot.float().to(DEV))
    torch.cuda.synchronize()
    gmax = max(t.grad.abs().max().item() for t in ins_cpu if t is not None and t.grad is not None)
    for name, a, b in zip(("query", "x"), ins_gpu, ins_cpu):
        if b is None or b.grad is None:
            continue
        # (a case whose keys all fall outside the image has identical keys and analytically zero input gradients)
        e = rel(a.grad, b.grad, floor=max(1e-2 * gmax, 1e-1 if bf16 else 1e-3))
        assert e < (6e-2 if bf16 else 1e-2 if loose else 4e-3), f"{tag}: grad {name} {e:.3e}"   # offsets clamp / tanh-saturate: kinks in d(query) too
    pmax = max(v.grad.abs().max().item() for v in params_cpu.values() if v.grad is not None)
    for n, p in m.named_parameters():
        b = params_cpu[n]
        if b.grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, f"{tag}: {n} has a gradient, the reference none"
            continue
        if n == "proj_k.bias":   # a constant added to every key moves no softmax: analytically zero, numerically noise
            assert p.grad.abs().max().item() <= 0.1 * pmax + 1e-3, f"{tag}: {n}"
            continue
        e = rel(p.grad, b.grad, floor=max(2e-2 * pmax, 1e-2 if bf16 else 1e-4))
        # loose: without the tanh range the random offsets throw most keys onto the clamp at +-1 (pixel centres and the
        # table's edge: derivative jumps), and a few keys on the other side of one move the offset head's gradients
        assert e < (6e-2 if bf16 else 1e-2 if loose else 3e-3), f"{tag}: grad {n} {e:.3e}"


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
@pytest.mark.parametrize("seed", list(range(NSEED)))
def test_tsa_module_random(seed, prec):
    from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
    r = np.random.RandomState(seed)
    h = int(r.choice([1, 2, 4]))
    C = h * int(r.choice([8, 16, 32]))
    S = int(r.choice([6, 8, 12, 17, 24]))
    k, s = [(3, 1), (3, 1), (3, 2), (2, 2), (5, 1)][r.randint(0, 5)]
    if (S - (k if k != s else s)) // s + 1 < 2:
        k, s = 3, 1
    B = int(r.choice([1, 2]))
    xnone = bool(r.randint(0, 3) == 0)
    sor = bool(r.randint(0, 4) != 0)
    m = TSADeformableAttention(bev_feat_shape=S, dim_embed=C, n_heads=h, n_groups=1, stride=s, kernel_size=k,
                               scale_offset_range=sor, batch_size=B, n_views=1, precision=prec).to(DEV)
    randomize_(m, 100 + seed)
    g = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=g)
    prev = None if xnone else torch.randn(B, C, S, S, generator=g)
    p_cpu = {n: v.detach().cpu().double().requires_grad_(True) for n, v in m.state_dict().items()}
    qc = query.double().requires_grad_(True)
    xc = None if prev is None else prev.double().requires_grad_(True)
    want = O.tsa_forward(p_cpu, qc, xc, n_heads=h, n_groups=1, kernel_size=k, stride=s, scale_offset_range=sor)
    qg = query.clone().to(DEV).requires_grad_(True)
    xg = None if prev is None else prev.clone().to(DEV).requires_grad_(True)
    out, _ = m(xg, qg, None, False)
    compare(m, out, want, (qg, xg), (qc, xc), p_cpu, f"tsa seed {seed} C{C} h{h} S{S} k{k}s{s} xnone{xnone} sor{sor} prec{prec}", loose=not sor, bf16=prec == _lib.PREC_BF16)


@pytest.mark.parametrize("prec", [_lib.PREC_F32, _lib.PREC_BF16])
@pytest.mark.parametrize("seed", list(range(NSEED)))
def test_sca_module_random(seed, prec):
    from bevrender_amd.model.SCA_deform_attn import SCADeformableAttention
    r = np.random.RandomState(500 + seed)
    h = int(r.choice([2, 4]))
    g = int(r.choice([1, 1, 2]))
    C = h * int(r.choice([8, 16, 32]))
    S = int(r.choice([4, 6, 8, 12, 16]))
    D = int(r.choice([1, 2, 3, 5]))
    V = int(r.choice([1, 1, 2, 3]))
    B = int(r.choice([1, 2]))
    Hi, Wi = int(r.randint(3, 14)), int(r.randint(3, 20))
    sor = bool(r.randint(0, 4) != 0)
    m = SCADeformableAttention(bev_feat_shape=S, bev_depth_dim=D, dim_embed=C, n_heads=h, n_groups=g, stride=1,
                               kernel_size=3, scale_offset_range=sor, batch_size=B, n_views=V,
                               precision=prec).to(DEV)
    randomize_(m, 900 + seed)
    gen = torch.Generator().manual_seed(seed)
    query = torch.randn(B, C, S, S, generator=gen)
    x = torch.randn(B, V, C, Hi, Wi, generator=gen)
    ref = torch.rand(B, V, S // 2, S * D, 2, generator=gen) * 2.4 - 1.2
    ref[:, :, :, : (S * D) // 3] = -1.0                      # a third of the pillar points pinned to pixel (0, 0)
    p_cpu = {n: v.detach().cpu().double().requires_grad_(True) for n, v in m.state_dict().items()}
    qc, xc = query.double().requires_grad_(True), x.double().requires_grad_(True)
    want = O.sca_forward(p_cpu, xc, qc, ref.double(), n_heads=h, n_groups=g, depth_dim=D, scale_offset_range=sor)
    qg, xg = query.clone().to(DEV).requires_grad_(True), x.clone().to(DEV).requires_grad_(True)
    out, _ = m(xg, qg, ref.to(DEV), None, False)
    compare(m, out, want, (qg, xg), (qc, xc), p_cpu, f"sca seed {seed} C{C} h{h} g{g} S{S} D{D} V{V} {Hi}x{Wi} sor{sor} prec{prec}", loose=not sor, bf16=prec == _lib.PREC_BF16)


@pytest.mark.parametrize("seed", list(range(max(NSEED // 2, 6))))
def test_encoder_layer_random(seed):
    """One EncoderLayer (LPU depthwise, shared LayerNorm, TSA, conv-MLP, LPU, SCA over a random camera ring with the
    projected pillar grid, conv-MLP), train mode, F32: forward, input and every parameter gradient against the oracle's
    layer restatement with the layer's own random state_dict."""
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    from bevrender_amd.model.encoder import EncoderLayer
    from test_gpu_fullsize import ring_rig
    r = np.random.RandomState(700 + seed)
    h = int(r.choice([2, 4]))
    C = h * int(r.choice([8, 16]))
    S = int(r.choice([6, 8, 12, 14]))
    D = int(r.choice([2, 3, 5]))
    V = int(r.choice([1, 2, 3]))
    B = int(r.choice([1, 2]))
    img_w, img_h = 64 * int(r.choice([1, 2])), 32 * int(r.choice([1, 2]))
    Hi, Wi = img_h // 4, img_w // 4
    bound = {"X": float(r.choice([10, 20, 50])), "Y": float(r.choice([10, 20, 50])), "Z": 2.0}
    T, K = ring_rig(V, img_w, img_h)
    proj = BEV2CameraProjector(imu_to_rgb={0: [t.copy() for t in T]}, K={0: [k.copy() for k in K]},
                               vehicle_type_code=0, img_width=img_w, img_height=img_h, ori_img_width=img_w,
                               ori_img_height=img_h, device=DEV)
    layer = EncoderLayer(bev_bound=bound, bev2cmr_projector=proj, n_views=V, bev_feat_shape=S, bev_depth_dim=D,
                         z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1, n_heads=h, stride=1,
                         kernel_size=3, batch_size=B, scale_offset_range=True, drop_path_rate=0.0,
                         precision=_lib.PREC_F32).to(DEV)
    randomize_(layer, 1300 + seed)
    layer.train()
    gen = torch.Generator().manual_seed(seed)
    bev = torch.randn(B, C, S, S, generator=gen)
    prev = torch.randn(B, C, S, S, generator=gen)
    feat = torch.randn(B * V, C, Hi, Wi, generator=gen)
    p_cpu = {n: v.detach().cpu().double().requires_grad_(True) for n, v in layer.state_dict().items()
             if v.dtype.is_floating_point}
    pts = O.sample_3d_points(bound, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, img_w, img_h, img_w, img_h), B).double()
    bc, pc, fc = (t.double().requires_grad_(True) for t in (bev, prev, feat))
    want = O.encoder_layer_forward(p_cpu, bc, fc, pc, ref, n_heads=h, n_groups=1, depth_dim=D, n_views=V, kernel_size=3,
                                   stride=1, scale_offset_range=True)
    bg, pg, fg = (t.clone().to(DEV).requires_grad_(True) for t in (bev, prev, feat))
    out, _ = layer(bg, fg, pg, torch.zeros(B, 2, 3, device=DEV), torch.tensor(0), {}, False)
    tag = f"layer seed {seed} C{C} h{h} S{S} D{D} V{V} B{B} {img_w}x{img_h}"
    rel = lambda a, b, floor: (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), floor)
    assert rel(out.detach(), want.detach(), 1e-3) < 5e-4, f"{tag}: out"
    cot = torch.randn(want.shape, generator=torch.Generator().manual_seed(5)).double()
    want.backward(cot)
    out.backward(cot.float().to(DEV))
    torch.cuda.synchronize()
    for name, a, b in (("bev_query", bg, bc), ("prev_bev", pg, pc), ("img_feat", fg, fc)):
        e = rel(a.grad, b.grad, 1e-3)
        assert e < 5e-3, f"{tag}: grad {name} {e:.3e}"
    pmax = max(v.grad.abs().max().item() for v in p_cpu.values() if v.grad is not None)
    for n, p in layer.named_parameters():
        b = p_cpu[n]
        if b.grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, f"{tag}: {n} has a gradient, the reference none"
            continue
        if n.endswith("proj_k.bias"):
            continue
        e = rel(p.grad, b.grad, max(2e-2 * pmax, 1e-4))
        assert e < 5e-3, f"{tag}: grad {n} {e:.3e}"
