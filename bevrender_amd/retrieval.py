"""On-device retrieval metrics and global negatives for the correlation head.

Counterpart of the reference's validation bookkeeping: `Trainer` copies every batch's `(B, E)` camera / map
embeddings to the host into two float64 NumPy arrays (train.py:326-330, 384-395) and `get_recall` (train.py:551-572)
forms `2 - 2 A B^T` with `np.matmul` and ranks the diagonal with an O(N^2 * 11) Python loop.  Here the embeddings stay
in HBM (`RecallAccumulator`), the Gram is one GEMM -- a validation set is thousands of rows of E = 50 176 columns, a
real GEMM, so it goes to the matrix cores through rocBLAS -- and the rank count is a device reduction
(`bevr_recall_rank`, csrc/corr.hip, or a float64 comparison when exact agreement with the reference's float64
arithmetic is wanted).

`all_gather_embeddings` is the optional extension the survey names for the training loss: the reference's retrieval
losses see only the rank-local batch (train.py:224); gathering the embeddings of every rank over RCCL gives each
sample world_size x more negatives.  Off by default (parity with the reference).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

from . import ops


def get_recall(cam: torch.Tensor, mp: torch.Tensor, exact: bool = True) -> Tuple[float, float, float]:
    """recall@{1,5,10} in percent of `Trainer.get_recall` (train.py:551-572) for (N, E) device tensors:
    D = 2 - 2 cam map^T; rank[k] = #{i : D[i, k] < D[k, k]}; recall@r = mean(rank < r) * 100.
    exact=True: float64 GEMM and comparison, the reference's arithmetic (np.zeros default dtype; MI355X runs f64 GEMMs
    on its matrix cores too); exact=False: the float32 HIP Gram / rocBLAS GEMM + `bevr_recall_rank`."""
    if not cam.is_cuda:
        raise ops._lib.BevrError("get_recall needs ROCm device tensors; there is no CPU fallback")
    n = cam.shape[0]
    if exact:
        D = 2.0 - 2.0 * (cam.double() @ mp.double().t())
        rank = (D < D.diagonal()[None, :]).sum(0)
    else:
        if n * mp.shape[0] <= 64 * 64:
            D = ops.pairwise_corr(cam, mp, normalize=False)          # HBM-bound register-block Gram (csrc/corr.hip)
        else:
            D = 2.0 - 2.0 * (cam.float() @ mp.float().t())           # rocBLAS f32 GEMM (MFMA)
        rank = ops.recall_rank(D)
    r = torch.stack([(rank < k).double().mean() * 100.0 for k in (1, 5, 10)])
    r = r.cpu()                                                       # the one host sync of a validation epoch
    return float(r[0]), float(r[1]), float(r[2])


class RecallAccumulator:
    """Device-resident replacement of the reference's `global_camera_tensor` / `global_map_tensor` NumPy arrays."""

    def __init__(self, n_rows: int, dim: int, device, dtype=torch.float32):
        self.cam = torch.zeros(n_rows, dim, device=device, dtype=dtype)
        self.map = torch.zeros(n_rows, dim, device=device, dtype=dtype)
        self.batch = None

    def add(self, val_idx: int, camera_tensor: torch.Tensor, map_tensor: torch.Tensor) -> None:
        """train.py:384-395: rows [val_idx * B, (val_idx + 1) * B) <- this batch's embeddings (no D2H copy)."""
        B = camera_tensor.shape[0]
        self.batch = B if self.batch is None else self.batch
        sl = slice(val_idx * self.batch, val_idx * self.batch + B)
        self.cam[sl].copy_(camera_tensor.detach().flatten(1))
        self.map[sl].copy_(map_tensor.detach().flatten(1))

    def recall(self, exact: bool = True) -> Tuple[float, float, float]:
        return get_recall(self.cam, self.map, exact)


class _AllGather(torch.autograd.Function):
    """All-gather along dim 0 with the gradient every rank needs: each rank evaluates the same global loss on the
    gathered tensor, so the gradient of a rank's own rows is the SUM over ranks of the gradients they hold for those
    rows (all-reduce, then slice) -- which the data-parallel wrapper's mean over ranks then turns into the gradient
    of the global loss."""

    @staticmethod
    def forward(ctx, x):
        world = dist.get_world_size()
        ctx.rank, ctx.rows = dist.get_rank(), x.shape[0]
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x.contiguous())
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows]


def all_gather_embeddings(x: torch.Tensor) -> torch.Tensor:
    """(B_local, E) -> (world * B_local, E), differentiable; identity without a process group.  Rank order = row
    order, so labels [0..B-1, 0..B-1] of the retrieval losses stay aligned when cam and map are gathered alike."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    return _AllGather.apply(x)


def global_negative_loss(loss_module, cam: torch.Tensor, mp: torch.Tensor) -> torch.Tensor:
    """`loss_module.get_loss` over the embeddings of ALL ranks (world_size x more negatives per sample)."""
    return loss_module.get_loss(all_gather_embeddings(cam.flatten(1)), all_gather_embeddings(mp.flatten(1)))
