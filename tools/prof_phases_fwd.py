"""attn_fwd phase cycle breakdown with the instrumented build (libbevrender_prof.so, -DBEVR_PROF)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bevrender_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "bevrender_amd", "lib", "libbevrender_prof.so")
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
L.bevr_debug_prof_fwd(buf, 1)
os.environ["ITERS"] = "1"
os.environ.setdefault("FWD_ONLY", "1")
exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())
torch.cuda.synchronize()
L.bevr_debug_prof_fwd(buf, 0)
names = ["region+consts", "frag+qk-mfma", "bias-loop", "softmax", "pv-mfma", "stage+barrier", "step-total", "n_step"]
for w in (0, 1):
    v = list(buf[8 * w: 8 * w + 8])
    n = max(v[7], 1)
    print("wave", 7 * w, {k: (round(x / n, 1) if i != 7 else x) for i, (k, x) in enumerate(zip(names, v))})
