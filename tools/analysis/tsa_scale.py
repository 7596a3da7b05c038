"""TSA forward with a history BEV of growing magnitude: where does the output stop being finite?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevrender_amd import ops, _lib
from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
torch.manual_seed(0)
S, C, h, B = int(os.environ.get("S", "200")), 64, 2, 1
dev = "cuda"
prec = os.environ.get("PREC", "f32")
tsa = TSADeformableAttention(S, C, h, 1, 1, 3, True, B, n_views=1, precision=prec).to(dev)
for mod in tsa.modules():
    if isinstance(mod, torch.nn.Conv2d):
        torch.nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
q = torch.nn.functional.layer_norm(torch.randn(B, S, S, C, device=dev), (C,)).permute(0, 3, 1, 2).contiguous()
x0 = torch.randn(B, C, S, S, device=dev)
for sc in (1, 8, 32, 64, 128, 256, 1024):
    with torch.no_grad():
        out, _ = tsa(x0 * sc, q, None, False)
        pos = tsa.key_positions(q)
        xs = ops.sample_features(x0 * sc, pos, 1)
    print(f"scale {sc}: out finite {torch.isfinite(out).all().item()} max {out.abs().max().item():.3e}; "
          f"pos finite {torch.isfinite(pos).all().item()}; sampled max {xs.abs().max().item():.3e}", flush=True)
