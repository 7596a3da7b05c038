import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevrender_amd import _lib
from bevrender_amd.model.TSA_deform_attn import TSADeformableAttention
name = sys.argv[1] if len(sys.argv) > 1 else "tsa_c64.npz"
z = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name)))
B, C, h, g, S, k, s, sor, xnone = [int(v) for v in z["cfg"]]
for prec in (0, 1):
    m = TSADeformableAttention(S, C, h, g, s, k, bool(sor), B, n_views=1, precision=prec).cuda()
    m.load_state_dict({kk[6:]: torch.tensor(v) for kk, v in z.items() if kk.startswith("param.")})
    q = torch.tensor(z["query"]).cuda().requires_grad_(True)
    p = torch.tensor(z["prev_bev"]).cuda().requires_grad_(True)
    out, _ = m(p, q, {}, False)
    out.backward(torch.tensor(z["cot"]).cuda())
    print("prec", prec, "out err", float((out.detach().cpu() - torch.tensor(z["out"])).abs().max()))
    for kk, v in z.items():
        if kk.startswith("grad_param."):
            gg = dict(m.named_parameters())[kk[11:]].grad.cpu().numpy()
        elif kk == "grad_in.query": gg = q.grad.cpu().numpy()
        elif kk == "grad_in.prev_bev": gg = p.grad.cpu().numpy()
        else: continue
        print(f"   {kk:40s} relerr {np.abs(gg - v).max() / (np.abs(v).max() + 1e-12):.3e}  max {np.abs(v).max():.3e}")
