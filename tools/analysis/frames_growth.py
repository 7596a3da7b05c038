import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from bench import LiftBlock
dev = torch.device("cuda", 0)
S, C, heads, D, V, L, B = 200, 64, 2, 5, 6, 2, 2
torch.manual_seed(15213)
m = LiftBlock(S, C, heads, D, V, L, 704, 256, sys.argv[1] if len(sys.argv) > 1 else "bf16", dev).to(dev)
gen = torch.Generator(device=dev).manual_seed(15213)
f = torch.randn(B * V, C, 64, 176, device=dev, dtype=torch.bfloat16, generator=gen).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    prev = None
    for t in range(6):
        prev = m.encode(f, prev)
        print(t, "max|bev|", prev.abs().max().item(), "finite", torch.isfinite(prev).all().item(), flush=True)

# ---- where does the first non-finite value appear? (forward hooks on every module, on the first failing frame)
bad = []
def hook(mod, inp, kw, out):
    o = out[0] if isinstance(out, tuple) else out
    if torch.is_tensor(o) and not torch.isfinite(o).all() and not bad:
        ins = [i for i in list(inp) + list(kw.values()) if torch.is_tensor(i)]
        torch.save({"args": [i.cpu() if torch.is_tensor(i) else i for i in inp], "kw": {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items() if torch.is_tensor(v)},
                    "state": {k: v.cpu() for k, v in mod.state_dict().items()}}, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "tsa_nan_case.pt"))
        bad.append((mod.__class__.__name__, [(tuple(i.shape), i.abs().max().item(), torch.isfinite(i).all().item()) for i in ins]))
names = {mod: n for n, mod in m.named_modules()}
hs = [mod.register_forward_hook(hook, with_kwargs=True) for mod in m.modules()]
with torch.no_grad():
    prev = None
    for t in range(6):
        prev = m.encode(f, prev)
        if bad:
            print("frame", t, "first non-finite output:", bad[0])
            break
