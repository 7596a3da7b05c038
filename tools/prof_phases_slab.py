"""attn_slab_bwd_q phase clock breakdown with the instrumented build (make -C bevrender_amd/csrc SPROF=1 OUTDIR=../lib_sprof)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bevrender_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "bevrender_amd", "lib_sprof", "libbevrender_hip.so")
L = _lib.lib()
buf = (C.c_ulonglong * 48)()
os.environ["ITERS"] = "2"
exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())      # warm-up + one timed pair of iterations
L.bevr_debug_prof_slab(buf, 1)
os.environ["ITERS"] = "1"
exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())
torch.cuda.synchronize()
L.bevr_debug_prof_slab(buf, 0)
v = list(buf)
for tag, b in (("worker wave 0", 0), ("worker wave 6", 16)):
    n = max(1, v[b + 3])
    print(f"{tag}: emissions {v[b+3]} (with a live key in half 0: {v[b+2]}), items {v[b+15]}; clk per emission: "
          f"barrier wait {v[b]/n:.0f}, emission body {v[b+1]/n:.0f} (key-row loop {v[b+4]/n:.0f}, dQ product {v[b+5]/n:.0f}); "
          f"per item: slab in {v[b+12]/max(1,v[b+15]):.0f}, slab out {v[b+13]/max(1,v[b+15]):.0f}")
b = 32
n = max(1, v[b + 3])
print(f"producer: emissions {v[b+3]}; clk per emission: wait for loads {v[b]/n:.0f}, constants + stores {v[b+1]/n:.0f}, "
      f"barrier wait {v[b+2]/n:.0f}")
