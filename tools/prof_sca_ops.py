"""torch.profiler view of tools/prof_sca.py (one SCA block, fwd+bwd): device time of everything that is NOT an attention
kernel -- the host-side glue of the tap / gather paths (thin GEMMs, casts, norms, merges) and the offset heads --
by operator and input shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
os.environ["ITERS"] = "1"
exec(open(os.path.join(ROOT, "tools", "prof_sca.py")).read())      # warm-up (allocator, first-use compiles)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    out, _ = sca(q, feat, torch.tensor(0), None, False)
    out.square().mean().backward()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows if "attn_" not in e.key and not e.key.startswith("_") and e.self_device_time_total > 0)
print(f"non-attention device time: {tot / 1e3:.2f} ms")
for e in rows[:90]:
    if "attn_" in e.key or e.self_device_time_total <= 0 or e.key.startswith("_"):
        continue
    print(f"{e.self_device_time_total / 1e3:8.3f} ms {e.count:4d}  {e.key[:28]:28s} {str(e.input_shapes)[:120]}")
