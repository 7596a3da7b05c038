"""torch.profiler view of tools/prof_sca.py (one SCA block, fwd+bwd): device time of everything that is NOT an attention
kernel -- the host-side glue of the tap / gather paths (thin GEMMs, casts, norms, merges) and the offset heads."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
os.environ.setdefault("ITERS", "2")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())
    torch.cuda.synchronize()
n = int(os.environ["ITERS"])
rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows if e.self_device_time_total > 0 and "attn_" not in e.key and not e.key.startswith("_"))
print(f"non-attention device time per iteration: {tot / n / 1e3:.2f} ms")
for e in rows[:45]:
    if "attn_" in e.key or e.self_device_time_total <= 0:
        continue
    print(f"{e.self_device_time_total / n / 1e3:8.3f} ms {e.count // n:5d} calls  {e.key[:100]}")
