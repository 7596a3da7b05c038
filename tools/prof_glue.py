"""torch.profiler view of one benchmark step: device time by operator, attention kernels excluded (the PyTorch glue)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--f32-steps", "0"]
import bench
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.main()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = 0.0
for e in rows[:int(os.environ.get('ROWS', '90'))]:      # (2 steps profiled: warmup + timed)
    if "attn" in e.key.lower() or "Memcpy" in e.key or "hipEvent" in e.key:
        continue
    tot += e.self_device_time_total / 2e3
    print(f"{e.self_device_time_total/2e3:9.2f} ms/step {e.count//2:6d} calls/step  {e.key[:70]}  {str(e.input_shapes)[:100]}")
print("listed total ms/step", round(tot, 1))
