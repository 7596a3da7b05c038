#!/usr/bin/env python3
"""G5: golden output of the reference's full `BEVRender` (model/bevrender.py:14-221) on a synthetic 2-frame window.

Run ONLY in the build container (needs /root/reference and this repo):  python tests/golden/make_golden_full.py

The model has 2.0 M parameters (8 MB): too large to commit.  Instead the weights are *this repo's* `BEVRender`
constructed under `torch.manual_seed(SEED)` on the CPU (deterministic for a given torch build); they are loaded
into the reference model with `strict=True` -- which also pins that both models expose the same 452 state-dict
entries with the same shapes -- and only inputs' seed and sampled outputs are stored.  The test rebuilds the same
weights from the same seed on the GPU box.  Import stubs for timm / torchvision: see make_golden.py.
"""
import importlib.util
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SEED, IN_SEED = 1234, 99


def full_config():
    R0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    T = np.eye(4)
    T[:3, :3] = R0
    T[:3, 3] = (0, 0, 1.5)
    K = np.array([[100, 0, 64, 0], [0, 100, 64, 0], [0, 0, 1, 0]], dtype=np.float64)
    return dict(
        BATCH_SIZE=2, DATA_TYPE=torch.float32, DAT_BEV_SHAPE=[28, 28], DAT_EMBED_DIMS=[64, 64], DAT_NUM_STAGES=1,
        DAT_VIT_DEPTHS=[1], DAT_NUM_HEADS=[2], DAT_NUM_GROUPS=[1], DAT_STRIDES=[1], DAT_K_SIZES=[3], DAT_EXPANSION=4,
        DAT_BEV_DEPTH_DIM=5, SAMPLE_Z_SHIFT=-1.0, BEV_BOUND={"X": 20, "Y": 10, "Z": 2}, NUM_VIEWS=1, IMG_HEIGHT=128,
        IMG_WIDTH=128, ORI_IMG_HEIGHT=128, ORI_IMG_WIDTH=128, VEHICLE_TYPE_CODE=0, IMU_TO_RGB={0: [T.copy()]},
        INTRINSIC_K={0: [K.copy()]}, DAT_SCALE_OFFSET_RANGE=True, DAT_DROP_RATE=0.0, DAT_ATTN_DROP_RATE=0.0,
        DAT_DROP_PATH_RATE=0.0, DAT_BACKBONE_TYPE="ResNet18", DECODER_HID_DIM=64, REMOVE_REF_IN_GRAY=False,
        BOUND_CHECK_IMG_PATH=None)


def full_inputs():
    g = torch.Generator().manual_seed(IN_SEED)
    img = torch.randn(2, 2, 1, 3, 128, 128, generator=g) * 0.5
    pose = torch.zeros(2, 2, 3)
    vtype = torch.zeros(2, 1, dtype=torch.long)
    return img, pose, vtype


def sample_index(n, k, seed):
    return torch.randperm(n, generator=torch.Generator().manual_seed(seed))[:k]


def main():
    spec = importlib.util.spec_from_file_location("mg", os.path.join(HERE, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_import_stubs()
    log = logging.getLogger("golden")

    sys.path.insert(0, ROOT)
    from bevrender_amd.model.bevrender import BEVRender as Mine
    torch.manual_seed(SEED)
    mine = Mine(full_config(), log, "train")
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    sys.path.remove(ROOT)

    sys.path.insert(0, "/root/reference")
    from model.bevrender import BEVRender as Ref
    torch.manual_seed(0)
    ref = Ref(full_config(), log, "train")
    ref.load_state_dict(sd, strict=True)
    img, pose, vtype = full_inputs()
    out, _ = ref(img, pose, vtype, {}, False)
    out.sum().backward()
    g_emb = ref.bev_embedding.weight.grad
    oi = sample_index(out.numel(), 256, 7)
    gi = sample_index(g_emb.numel(), 256, 8)
    np.savez_compressed(
        os.path.join(HERE, "full_bevrender.npz"), seed=SEED, in_seed=IN_SEED, out_shape=np.array(out.shape),
        out_sum=out.detach().double().sum().numpy(), out_abs_sum=out.detach().double().abs().sum().numpy(),
        out_idx=oi.numpy(), out_val=out.detach().flatten()[oi].numpy(),
        gemb_idx=gi.numpy(), gemb_val=g_emb.flatten()[gi].numpy(), gemb_abs_sum=g_emb.double().abs().sum().numpy(),
        n_state=len(sd))
    print("wrote full_bevrender.npz: out", tuple(out.shape), "sum", float(out.sum()), "n_state", len(sd))


def decoder_input(S):
    return torch.randn(2, 64, S, S, generator=torch.Generator().manual_seed(300 + S))


def gen_decoder():
    """G6: BEVImageRenderDecoder (model/decoder_img_render.py:4-93) at the three BEV sides it is defined for, train
    mode (batch statistics).  Same seeded-weights scheme as G5; plain PyTorch on both sides, so this one runs on CPU."""
    log = logging.getLogger("golden")
    rec = {}
    for S in (14, 28, 56):
        for m in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[m]
        sys.path.insert(0, ROOT)
        from bevrender_amd.model.decoder_img_render import BEVImageRenderDecoder as Mine
        torch.manual_seed(SEED + S)
        sd = {k: v.clone() for k, v in Mine(bev_spatial_dim=S, model_dim=64, hid_dim=64).state_dict().items()}
        sys.path.remove(ROOT)
        sys.path.insert(0, "/root/reference")
        from model.decoder_img_render import BEVImageRenderDecoder as Ref
        ref = Ref(bev_spatial_dim=S, model_dim=64, hid_dim=64, logger=log)
        ref.load_state_dict(sd, strict=True)
        sys.path.remove("/root/reference")
        ref.train()
        out = ref(decoder_input(S)).detach()
        oi = sample_index(out.numel(), 128, S)
        rec[f"s{S}.shape"] = np.array(out.shape)
        rec[f"s{S}.idx"] = oi.numpy()
        rec[f"s{S}.val"] = out.flatten()[oi].numpy()
        rec[f"s{S}.sum"] = out.double().sum().numpy()
        rec[f"s{S}.n_state"] = len(sd)
    np.savez_compressed(os.path.join(HERE, "decoder.npz"), seed=SEED, **rec)
    print("wrote decoder.npz")


if __name__ == "__main__":
    main()
    gen_decoder()
