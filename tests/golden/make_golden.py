#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own PyTorch modules.

Run ONLY in the build container (needs /root/reference); the produced .npz files
are committed and are the only thing that travels.  Usage:

    python tests/golden/make_golden.py [--ref /root/reference]

The reference imports two packages that are absent from this image (ordinary
ModuleNotFoundError): `timm` (trunc_normal_, DropPath) and `torchvision`
(transforms.functional.affine / pil_to_tensor, InterpolationMode).  Neither is on
the arithmetic path pinned here (trunc_normal_ only initialises rpe_table and the
fixtures store explicit weights; DropPath is identity at rate 0; torchvision is
only reached by the eval-mode ego-warp and the optional grey mask, neither used).
They are registered as in-memory import stubs below, for this script only.

Fixtures (fp32, seeded):
  G1 tsa_*.npz      TSADeformableAttention   model/TSA_deform_attn.py:128-337
  G2 sca_*.npz      SCADeformableAttention   model/SCA_deform_attn.py:180-421 (n_views=1)
  G3 proj_*.npz     sample_3d_points + bev_grid_to_camera  model/SCA.py:112-162, model/bev_cmr_proj.py:61-124
  G4 enclayer.npz   EncoderLayer fwd+bwd (train mode, drop 0)  model/encoder.py:339-411
  G4b enclayer_s56.npz  the same at BEV side 56, C=64, D=5 (M=3136, N_sca=7840: the largest side the reference's
                    materialised tensors fit in this container); weights/inputs are seeded, only sampled values stored
  G5 full_bevrender.npz  full BEVRender fwd+bwd, T=2, S=28 (separate script: make_golden_full.py)  model/bevrender.py:14-221
  G7 recall.npz     Trainer.get_recall        train.py:551-572
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def install_import_stubs():
    import torch.nn as nn

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm_layers.DropPath = DropPath
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode:
        BILINEAR = "bilinear"

    tvt.InterpolationMode = InterpolationMode
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **to_np(d))
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def randomize_(module, seed):
    """Give every parameter a non-degenerate value (reference init zeros many biases)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("rpe_table"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.ndim <= 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)


def randomize_by_name_(module, seed):
    """randomize_ with one generator per parameter, seeded from the parameter's NAME: independent of the order in
    which a module registers its parameters, so this repo's module and the reference's get identical weights."""
    import zlib
    with torch.no_grad():
        for n, p in module.named_parameters():
            g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(n.encode()))
            if n.endswith("rpe_table"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.ndim <= 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)


def run_module(mod, call, inputs, cot_seed):
    """forward + backward with a random cotangent; returns dict of out/grads."""
    for v in inputs.values():
        if torch.is_tensor(v) and v.is_floating_point():
            v.requires_grad_(True)
    out = call()
    g = torch.Generator().manual_seed(cot_seed)
    cot = torch.randn(out.shape, generator=g)
    mod.zero_grad()
    out.backward(cot)
    rec = {"out": out, "cot": cot}
    for k, v in inputs.items():
        if torch.is_tensor(v) and v.grad is not None:
            rec["grad_in." + k] = v.grad
    for n, p in mod.named_parameters():
        rec["param." + n] = p
        if p.grad is not None:
            rec["grad_param." + n] = p.grad
    return rec


def gen_tsa():
    from model.TSA_deform_attn import TSADeformableAttention

    cases = {
        # name: (B, C, h, g, S, k, s, scale_offset_range, x_none)
        "tsa_k3s1": (2, 16, 2, 1, 8, 3, 1, True, False),
        "tsa_k3s2": (2, 16, 2, 1, 8, 3, 2, True, False),
        "tsa_k9s8": (1, 16, 2, 1, 16, 9, 8, True, False),
        "tsa_k2s2": (2, 16, 4, 1, 8, 2, 2, True, False),
        "tsa_xnone": (2, 16, 2, 1, 8, 3, 1, True, True),
        "tsa_clamp": (2, 16, 2, 1, 8, 3, 1, False, False),
        "tsa_c64": (1, 64, 2, 1, 12, 3, 1, True, False),
    }
    for i, (name, (B, C, h, g, S, k, s, sor, xnone)) in enumerate(cases.items()):
        torch.manual_seed(15213 + i)
        m = TSADeformableAttention(bev_feat_shape=S, dim_embed=C, n_heads=h, n_groups=g, stride=s,
                                   kernel_size=k, scale_offset_range=sor, batch_size=B, n_views=1)
        randomize_(m, 100 + i)
        if not sor:  # make the clamp bite: larger raw offsets
            with torch.no_grad():
                m.conv_offset[3].weight.mul_(6.0)
        query = torch.randn(B, C, S, S)
        prev = None if xnone else torch.randn(B, C, S, S)
        inputs = {"query": query, "prev_bev": prev}
        rec = run_module(m, lambda: m(prev, query, {}, False)[0], inputs, 7 + i)
        rec.update({"query": query, "cfg": np.array([B, C, h, g, S, k, s, int(sor), int(xnone)])})
        if prev is not None:
            rec["prev_bev"] = prev
        save(name + ".npz", rec)


def gen_sca():
    from model.SCA_deform_attn import SCADeformableAttention

    cases = {
        # name: (B, C, h, g, S, D, Hi, Wi, scale_offset_range, ref_mode)
        "sca_rand": (2, 16, 2, 1, 8, 3, 6, 10, True, "rand"),
        "sca_masked": (2, 16, 2, 1, 8, 3, 6, 10, True, "masked"),
        "sca_g2": (2, 16, 4, 2, 8, 3, 6, 10, True, "rand"),
        "sca_clamp": (2, 16, 2, 1, 8, 3, 6, 10, False, "rand"),
        "sca_c64": (1, 64, 2, 1, 10, 5, 8, 22, True, "masked"),
    }
    for i, (name, (B, C, h, g, S, D, Hi, Wi, sor, mode)) in enumerate(cases.items()):
        torch.manual_seed(25213 + i)
        m = SCADeformableAttention(bev_feat_shape=S, bev_depth_dim=D, dim_embed=C, n_heads=h, n_groups=g,
                                   stride=1, kernel_size=3, scale_offset_range=sor, batch_size=B, n_views=1)
        randomize_(m, 200 + i)
        if not sor:
            with torch.no_grad():
                m.conv_offset_m0[3].weight.mul_(6.0)
        query = torch.randn(B, C, S, S)
        x = torch.randn(B, 1, C, Hi, Wi)
        ref = torch.rand(B, 1, S // 2, S * D, 2) * 2.4 - 1.2
        ref[:] = ref[0:1]  # the caller repeats one static grid over the batch (SCA.py:83-85)
        if mode == "masked":
            mask = torch.rand(1, 1, S // 2, S * D, 1) < 0.4
            ref = torch.where(mask, torch.full_like(ref, -1.0), ref)
        inputs = {"x": x, "query": query}
        rec = run_module(m, lambda: m(x, query, ref, {}, False)[0], inputs, 17 + i)
        rec.update({"x": x, "query": query, "reference_points": ref,
                    "cfg": np.array([B, C, h, g, S, D, Hi, Wi, int(sor)])})
        save(name + ".npz", rec)


def make_rig(kind):
    """Two camera rigs: imu_to_rgb 4x4 (list per cam) + K 3x4 (list per cam)."""
    R0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)  # cam looks along +X of IMU

    def T(yaw_deg, t):
        a = np.deg2rad(yaw_deg)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        M = np.eye(4)
        M[:3, :3] = Rz @ R0
        M[:3, 3] = t
        return M

    if kind == "front1":
        return [T(0, (0, 0, 1.5))], [np.array([[100., 0, 64, 0], [0, 100., 64, 0], [0, 0, 1, 0]])]
    if kind == "ring3":
        return ([T(0, (0.3, 0.0, 1.5)), T(120, (-0.2, 0.4, 1.4)), T(-120, (-0.2, -0.4, 1.6))],
                [np.array([[180., 0, 128, 0], [0, 170., 96, 0], [0, 0, 1, 0]]) for _ in range(3)])
    raise ValueError(kind)


def gen_proj():
    from model.SCA import SpatialCrossAttn
    from model.bev_cmr_proj import BEV2CameraProjector

    cases = {
        # name: (rig, S, D, bound, z_shift, img_w, img_h, ori_w, ori_h)
        "proj_front1_s8": ("front1", 8, 3, {"X": 20, "Y": 10, "Z": 2}, -1.0, 128, 128, 128, 128),
        "proj_ring3_s28": ("ring3", 28, 5, {"X": 50, "Y": 50, "Z": 2}, -1.0, 128, 96, 256, 192),
        "proj_front1_s50": ("front1", 50, 5, {"X": 20, "Y": 10, "Z": 2}, -1.0, 128, 128, 128, 128),
    }
    for name, (rig, S, D, bound, zs, iw, ih, ow, oh) in cases.items():
        T, K = make_rig(rig)
        K_in = [k.copy() for k in K]
        proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=iw, img_height=ih,
                                   ori_img_width=ow, ori_img_height=oh, device="cpu")
        sca = SpatialCrossAttn.__new__(SpatialCrossAttn)  # only sample_3d_points is needed
        sca.bev_bound, sca.bev_feat_shape, sca.bev_depth_dim, sca.z_shift = bound, S, D, zs
        pts3d = SpatialCrossAttn.sample_3d_points(sca)
        pts2d = proj.bev_grid_to_camera(pts3d)[0]
        save(name + ".npz", {
            "imu_to_rgb": np.stack(T), "K": np.stack(K_in),
            "cfg": np.array([S, D, bound["X"], bound["Y"], bound["Z"], zs, iw, ih, ow, oh], dtype=np.float64),
            "points_3d": pts3d, "points_2d": torch.stack(pts2d, 0)})


def gen_enclayer():
    from model.encoder import EncoderLayer
    from model.bev_cmr_proj import BEV2CameraProjector

    torch.manual_seed(35213)
    B, C, S, D, h = 2, 16, 8, 3, 2
    T, K = make_rig("front1")
    K_in = [k.copy() for k in K]
    proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=128, img_height=128,
                               ori_img_width=128, ori_img_height=128, device="cpu")
    bound = {"X": 20, "Y": 10, "Z": 2}
    layer = EncoderLayer(bev_bound=bound, bev2cmr_projector=proj, n_views=1, bev_feat_shape=S, bev_depth_dim=D,
                         z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1, n_heads=h, stride=1,
                         kernel_size=3, batch_size=B, scale_offset_range=True, drop_path_rate=0.0)
    randomize_(layer, 300)
    layer.train()
    bev_query = torch.randn(B, C, S, S)
    prev_bev = torch.randn(B, C, S, S)
    img_feat = torch.randn(B * 1, C, 12, 12)
    pose = torch.zeros(B, 2, 3)
    vtype = torch.tensor(0)
    inputs = {"bev_query": bev_query, "prev_bev": prev_bev, "img_feat": img_feat}
    rec = run_module(layer, lambda: layer(bev_query, img_feat, prev_bev, pose, vtype, {}, False)[0], inputs, 31)
    rec.update({"bev_query": bev_query, "prev_bev": prev_bev, "img_feat": img_feat,
                "imu_to_rgb": np.stack(T), "K": np.stack(K_in),
                "cfg": np.array([B, C, S, D, h, bound["X"], bound["Y"], bound["Z"]])})
    save("enclayer.npz", rec)


ENC56 = dict(B=1, C=64, S=56, D=5, h=2, Hi=32, Wi=32, bound={"X": 20, "Y": 10, "Z": 2}, wseed=356, iseed=357,
             cseed=358)


def enclayer_s56_inputs():
    """Seeded inputs of G4b (shared with tests/test_gpu_fullsize.py: the CPU generator is deterministic for a
    given torch build, and the GPU box runs the same image)."""
    c = ENC56
    g = torch.Generator().manual_seed(c["iseed"])
    bev_query = torch.randn(c["B"], c["C"], c["S"], c["S"], generator=g)
    prev_bev = torch.randn(c["B"], c["C"], c["S"], c["S"], generator=g)
    img_feat = torch.randn(c["B"], c["C"], c["Hi"], c["Wi"], generator=g)
    cot = torch.randn(c["B"], c["C"], c["S"], c["S"], generator=torch.Generator().manual_seed(c["cseed"]))
    return bev_query, prev_bev, img_feat, cot


def sample_idx(n, k, seed):
    return torch.randperm(n, generator=torch.Generator().manual_seed(seed))[:k]


def gen_enclayer_s56():
    from model.encoder import EncoderLayer
    from model.bev_cmr_proj import BEV2CameraProjector

    c = ENC56
    T, K = make_rig("front1")
    proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: [k.copy() for k in K]}, vehicle_type_code=0, img_width=128,
                               img_height=128, ori_img_width=128, ori_img_height=128, device="cpu")
    torch.manual_seed(0)
    layer = EncoderLayer(bev_bound=c["bound"], bev2cmr_projector=proj, n_views=1, bev_feat_shape=c["S"],
                         bev_depth_dim=c["D"], z_shift=-1.0, dim_embed=c["C"], expansion=4, stage_idx=0, n_groups=1,
                         n_heads=c["h"], stride=1, kernel_size=3, batch_size=c["B"], scale_offset_range=True,
                         drop_path_rate=0.0)
    randomize_by_name_(layer, c["wseed"])
    layer.train()
    bev_query, prev_bev, img_feat, cot = enclayer_s56_inputs()
    ins = {"bev_query": bev_query, "prev_bev": prev_bev, "img_feat": img_feat}
    for v in ins.values():
        v.requires_grad_(True)
    out = layer(bev_query, img_feat, prev_bev, torch.zeros(c["B"], 2, 3), torch.tensor(0), {}, False)[0]
    layer.zero_grad()
    out.backward(cot)
    rec = {"out_sum": out.detach().double().sum(), "out_abs_sum": out.detach().double().abs().sum()}
    oi = sample_idx(out.numel(), 2048, 1)
    rec["out_idx"], rec["out_val"] = oi, out.detach().flatten()[oi]
    names = []
    for tag, items in (("grad_in.", ins.items()), ("grad_param.", layer.named_parameters())):
        for n, t in items:
            g = t.grad
            if g is None:
                continue
            gi = sample_idx(g.numel(), min(g.numel(), 256), 2)
            rec[tag + n + ".idx"], rec[tag + n + ".val"] = gi, g.flatten()[gi]
            rec[tag + n + ".absmax"] = g.abs().max()
            names.append(tag + n)
    rec["names"] = np.array(names)
    rec["n_state"] = len(layer.state_dict())
    save("enclayer_s56.npz", rec)


def gen_recall():
    """Trainer.get_recall (train.py:551-572) is a method that uses no `self` state."""
    import importlib.util
    import ast
    # train.py imports wandb etc. at module level; pull the one function out by source (executed, not copied).
    src = open(os.path.join(REF, "train.py")).read()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "get_recall":
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, "train.py:get_recall", "exec"), ns)
    rng = np.random.default_rng(15213)
    rec = {}
    for tag, (n, e, noise) in {"a": (24, 32, 0.6), "b": (40, 16, 1.5)}.items():
        cam = rng.standard_normal((n, e))
        mp = cam + noise * rng.standard_normal((n, e))
        cam /= np.linalg.norm(cam, axis=1, keepdims=True)
        mp /= np.linalg.norm(mp, axis=1, keepdims=True)
        r = ns["get_recall"](None, cam, mp)
        rec[f"cam_{tag}"], rec[f"map_{tag}"], rec[f"recall_{tag}"] = cam, mp, np.array(r)
    save("recall.npz", rec)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None, help="generate one fixture family only (e.g. enclayer_s56)")
    args = ap.parse_args()
    REF = args.ref
    sys.path.insert(0, REF)
    os.chdir(REF)  # reference modules do sys.path.append(Path.cwd())
    install_import_stubs()
    torch.set_num_threads(4)
    if args.only:
        globals()["gen_" + args.only]()
        sys.exit(0)
    gen_tsa()
    gen_sca()
    gen_proj()
    gen_enclayer()
    gen_enclayer_s56()
    gen_recall()
