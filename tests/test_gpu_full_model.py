"""G5: the full drop-in `BEVRender` (backbone -> history frame -> TSA/SCA encoder -> render decoder) on the HIP
kernels against the reference's own full model (tests/golden/full_bevrender.npz, made by make_golden_full.py).

The weights are not in the fixture (8 MB): both sides build them as this repo's model under the fixture's seed on
the CPU; the generator loaded them into the reference model with strict=True (452 identical state-dict entries).
"""
import importlib.util
import logging
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("mgf", os.path.join(HERE, "golden", "make_golden_full.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_full_bevrender_matches_reference_forward_and_backward():
    from bevrender_amd.model.bevrender import BEVRender
    g = _gen()
    z = np.load(os.path.join(HERE, "golden", "full_bevrender.npz"))
    cfg = g.full_config()
    cfg["PRECISION"] = "f32"
    torch.manual_seed(int(z["seed"]))
    model = BEVRender(cfg, logging.getLogger("t"), "train")
    assert len(model.state_dict()) == int(z["n_state"])
    model = model.to("cuda")
    img, pose, vtype = g.full_inputs()
    out, _ = model(img.cuda(), pose.cuda(), vtype.cuda(), {}, False)
    assert tuple(out.shape) == tuple(z["out_shape"])
    out.sum().backward()
    torch.cuda.synchronize()
    got = out.detach().flatten().cpu()[torch.tensor(z["out_idx"])].numpy()
    # f32 kernels + MIOpen convolutions vs the reference's CPU convolutions: small float noise only
    np.testing.assert_allclose(got, z["out_val"], rtol=2e-3, atol=2e-3 * np.abs(z["out_val"]).max())
    assert abs(out.double().sum().item() - float(z["out_sum"])) <= 2e-3 * float(z["out_abs_sum"])
    ge = model.bev_embedding.weight.grad.flatten().cpu()[torch.tensor(z["gemb_idx"])].numpy()
    scale = np.abs(z["gemb_val"]).max()
    np.testing.assert_allclose(ge, z["gemb_val"], rtol=2e-2, atol=2e-2 * scale)


def test_full_bevrender_bf16_product_path_forward_and_backward():
    """The same golden on the BENCHED mode: bf16 operands (tap kernels for the projector-pinned keys, gather forward and
    the backward kernels for the rest), float backbone features and, second run, bf16-staged ones.  The reference is float32;
    the limits are the bf16 mode's (an untrained render CNN follows the encoder and amplifies the operands' 2^-9 rounding:
    compared in the 2-norm over the sampled entries, as tests/test_staging.py does)."""
    from bevrender_amd.model.bevrender import BEVRender
    g = _gen()
    z = np.load(os.path.join(HERE, "golden", "full_bevrender.npz"))
    for stage in (None, "bf16"):
        cfg = g.full_config()
        cfg["PRECISION"] = "bf16"
        cfg["STAGE_DTYPE"] = stage
        torch.manual_seed(int(z["seed"]))
        model = BEVRender(cfg, logging.getLogger("t"), "train").to("cuda")
        img, pose, vtype = g.full_inputs()
        out, _ = model(img.cuda(), pose.cuda(), vtype.cuda(), {}, False)
        out.float().sum().backward()
        torch.cuda.synchronize()
        got = out.detach().float().flatten().cpu()[torch.tensor(z["out_idx"])].numpy()
        ge = model.bev_embedding.weight.grad.float().flatten().cpu()[torch.tensor(z["gemb_idx"])].numpy()
        e_out = np.linalg.norm(got - z["out_val"]) / np.linalg.norm(z["out_val"])
        e_g = np.linalg.norm(ge - z["gemb_val"]) / np.linalg.norm(z["gemb_val"])
        e_sum = abs(out.double().sum().item() - float(z["out_sum"])) / float(z["out_abs_sum"])
        print(f"full model bf16 (stage {stage}): out {e_out:.3e} sum {e_sum:.3e} grad(embedding) {e_g:.3e}")
        assert np.isfinite(got).all() and np.isfinite(ge).all()
        assert e_out < OUT_LIMIT[stage] and e_sum < 2e-2 and e_g < GRAD_LIMIT[stage], (stage, e_out, e_sum, e_g)
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        assert not bad, bad


# relative 2-norm limits of the bf16 product path against the float32 reference (written next to the test so that a
# reader sees what "bf16 limits" means here).  Output: ~2.5x the errors observed on MI355X (0.021-0.024).  Gradient of the
# embedding: this untrained model's keys have norms up to 160, its softmax rows are decided by near ties, and the error
# of the gradient -- not of the output -- depends on the ROUNDING REALISATION: scaling the static softmax reference's key
# bound (bevrender_amd/ops.py, `ub`; any value is a valid reference) over 0.8 ... 1.05 moved it between 0.105 and 0.289 with
# no trend (0.105, 0.163, 0.136, 0.244, 0.170, 0.127, 0.128), so a limit of 0.15 held only for one realisation
OUT_LIMIT = {None: 0.10, "bf16": 0.25}
GRAD_LIMIT = {None: 0.45, "bf16": 0.45}
