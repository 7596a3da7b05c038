"""One forward + backward of the whole drop-in model on the bf16 product path (float backbone, bf16-staged features), for
a run on the bounds-checking build:  BEVRENDER_LIB=bevrender_amd/lib_debug/libbevrender_hip.so python tools/debug_full_bf16.py"""
import os, sys, logging, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_staging import _encoders
from bevrender_amd import _lib
print("library:", _lib.LIB_PATH)
for stage in ("bf16", None):
    m, g = _encoders(stage)
    m = m.cuda()
    img, pose, vtype = g.full_inputs()
    out, _ = m(img.cuda(), pose.cuda(), vtype.cuda(), {}, False)
    out.float().sum().backward()
    torch.cuda.synchronize()
    gq = m.bev_embedding.weight.grad
    print("stage", stage, "out finite", bool(torch.isfinite(out).all()), "grad finite", bool(torch.isfinite(gq).all()), float(gq.abs().sum()))
