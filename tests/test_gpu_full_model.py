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
