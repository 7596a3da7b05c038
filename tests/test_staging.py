"""bf16 staging of the camera images and backbone features (reference: model/encoder.py:98-110, where
the images go to the backbone in whatever dtype the config says).  `BEVEncoder(stage_dtype="bf16")` stages the images once,
channels-last, in bf16 and hands the backbone's features over in bf16 -- the form the 16-bit attention modes read without
a float copy (the map's gradient with bf16 maps: tests/test_gpu_ops.py::test_sample_features_reads_bf16_features_as_they_are)."""
import importlib.util
import logging
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(HERE, "golden", "make_golden_full.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _encoders(stage):
    from bevrender_amd.model.bevrender import BEVRender
    g = _gen()
    cfg = g.full_config()
    cfg["PRECISION"] = "bf16"
    cfg["STAGE_DTYPE"] = stage
    for k in ("DAT_DROP_PATH_RATE", "DAT_DROP_RATE", "DAT_ATTN_DROP_RATE"):     # forward() always runs the current frame in
        cfg[k] = 0.0                                                            # train mode: no random draws to compare
    torch.manual_seed(11)
    return BEVRender(cfg, logging.getLogger("t"), "train"), g


def test_bf16_staging_of_images_and_features_cpu():
    """The staging step alone (the backbone is plain PyTorch: runs on the CPU): dtype, layout, closeness to the f32 features,
    and the history frames as one batch."""
    m16, g = _encoders("bf16")
    m32, _ = _encoders(None)
    m32.load_state_dict(m16.state_dict())
    m16.eval(), m32.eval()
    img, _, _ = g.full_inputs()                      # (B, T, V, 3, H, W)
    cur = img[:, -1]
    with torch.no_grad():
        f16 = m16.encoder.backbone_features(cur)
        f32 = m32.encoder.backbone_features(cur)
        h16 = m16.encoder.history_features(img[:, :-1])
    assert m16.encoder.stage_dtype is torch.bfloat16 and m32.encoder.stage_dtype is None
    assert f16.dtype is torch.bfloat16 and f32.dtype is torch.float32 and f16.shape == f32.shape
    assert f16.is_contiguous(memory_format=torch.channels_last)
    err = (f16.float() - f32).abs().max().item() / f32.abs().max().item()
    assert err < 3e-2, err
    assert all(h.dtype is torch.bfloat16 and h.shape == f16.shape for h in h16)


@pytest.mark.gpu
def test_bf16_staging_full_model_gpu():
    """Whole drop-in model, forward AND backward, with bf16 staging against the same weights with the config's (float)
    staging, bf16 kernels on both sides: the render output moves by the bf16 rounding of images and features only, and the
    backward -- bf16 feature maps that carry a gradient through the fused K | V adjoint and the sampler's scatter, the tap /
    gather / region kernels inside `BEVRender`, the `feat.to(stage_dtype)` adjoint into MIOpen's backbone backward -- gives
    finite, non-zero gradients at both ends of the model (the backward this test lost in round 4)."""
    m16, g = _encoders("bf16")
    m32, _ = _encoders(None)
    m32.load_state_dict(m16.state_dict())
    m16, m32 = m16.cuda(), m32.cuda()
    img, pose, vtype = g.full_inputs()
    out16, _ = m16(img.cuda(), pose.cuda(), vtype.cuda(), {}, False)
    with torch.no_grad():
        out32, _ = m32(img.cuda(), pose.cuda(), vtype.cuda(), {}, False)
    torch.cuda.synchronize()
    # an untrained render CNN amplifies the features' bf16 rounding: compare in the 2-norm
    a, b = out16.detach().float().flatten(), out32.float().flatten()
    err = ((a - b).norm() / b.norm()).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    assert torch.isfinite(a).all() and err < 0.25 and cos > 0.97, (err, cos)
    loss = out16.float().sum()
    if os.environ.get("BEVR_TRACE_BACKWARD") == "1":
        # localisation aid: with HIP_LAUNCH_BLOCKING=1 the last node named on stderr before a GPU fault
        # is the one whose kernel faulted
        seen, todo = set(), [loss.grad_fn]
        while todo:
            fn = todo.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            # file descriptor 2 itself: sys.stderr is pytest's capture object (--capture=sys) and dies with the process
            fn.register_prehook(lambda g, _n=fn.name(): (os.write(2, f"[bwd] {_n}\n".encode()), None)[1])
            todo.extend(f for f, _ in fn.next_functions)
    loss.backward()
    torch.cuda.synchronize()
    gq = m16.bev_embedding.weight.grad
    assert gq is not None and torch.isfinite(gq).all() and gq.abs().sum() > 0
    # the far end of the backward: the backbone's first convolution, reached only through the bf16 feature maps' gradient
    gb = next(p.grad for p in m16.encoder.img_backbone.parameters() if p.requires_grad)
    assert gb is not None and torch.isfinite(gb).all() and gb.abs().sum() > 0
    bad = [n for n, p in m16.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    assert not bad, bad
