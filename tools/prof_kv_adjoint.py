"""Timing of the fused K | V source's adjoint GEMMs (ops._AttnCore.backward: dW = dkv^T xs, dxs = dkv Wkv) at the benchmark
shape (48 problems x 35 936 keys, C = 64): the stock float32 calls against batched / split forms."""
import torch, time
dev = "cuda"
P, N, C = 48, 35936, 64
d3 = torch.randn(P, N, 2 * C, device=dev)
xs = torch.randn(P, N, C, device=dev)
W = torch.randn(2 * C, C, device=dev)
d2, x2 = d3.reshape(-1, 2 * C), xs.reshape(-1, C)


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


a, ra = t(lambda: d2.t() @ x2)
b, rb = t(lambda: torch.bmm(d3.transpose(1, 2), xs).sum(0))
c, rc = t(lambda: torch.einsum("pnk,pnc->kc", d3, xs))
print(f"dW: mm {a:.2f} ms | bmm+sum {b:.2f} ms | einsum {c:.2f} ms | err bmm {(ra-rb).abs().max().item()/ra.abs().max().item():.1e}")
e, re_ = t(lambda: d2 @ W)
f, rf = t(lambda: torch.bmm(d3, W.expand(P, -1, -1)))
print(f"dxs: mm {e:.2f} ms | bmm {f:.2f} ms")
g, _ = t(lambda: d2.sum(0))
print(f"dbias: sum {g:.2f} ms")
