"""attn_gather_fwd phase cycle breakdown with the instrumented build (make -C bevrender_amd/csrc GPROF=1 OUTDIR=../lib_prof; PROF=1 is the region bwd_q's stamps and does not export bevr_debug_prof_gather)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bevrender_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "bevrender_amd", "lib_prof", "libbevrender_hip.so")
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
os.environ["ITERS"] = "2"
L.bevr_debug_prof_gather(buf, 1)
exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())
torch.cuda.synchronize()
L.bevr_debug_prof_gather(buf, 0)
pn = ["", "advance+stage", "", "wait fills+kv", "barrier wait", "n", "issue fills", ""]
cn = ["barrier wait", "compute", "wait fills", "", "loop total", "n", "", ""]
v = list(buf[0:8]); n = max(v[5], 1)
print("producer", {k: round(x / n, 1) for k, x in zip(pn, v) if k and k != "n"}, "n", v[5])
for w, o in ((0, 8), (3, 16)):
    v = list(buf[o:o + 8]); n = max(v[5], 1)
    print("wave", w, {k: round(x / n, 1) for k, x in zip(cn, v) if k and k != "n"}, "n", v[5])
