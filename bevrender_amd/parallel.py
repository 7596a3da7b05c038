"""Data parallelism for the BEV-lift path: one process per GPU, batch sharded across ranks, gradients
averaged by bucketed all-reduces that start while the backward is still running (RCCL over xGMI under the
"nccl" backend; "gloo" on CPU for tests).

Counterpart of the reference's DDP use (train.py:29-32 init_process_group, :128-141 SyncBatchNorm + DDP wrap
with find_unused_parameters=True, :668 DistributedSampler).  Differences by design:
  * the parameters the reference constructs but never uses (proj_q, proj_views, down_proj, ffn_tsa, ffn_sca,
    SCA offset heads of absent views) are frozen up front, so the reducer needs no unused-parameter search
    (a device->host sync per step in the reference);
  * every op on the hot path is per-sample independent (LayerNorm only, no BatchNorm inside TSA/SCA), so
    there is no data-path collective: ~2-3 M parameters = 8-11 MB of fp32 gradients.  PyTorch's default 25 MB bucket
    would hold all of them and fire only when the LAST gradient (the backbone's first convolution) is ready -- an
    all-reduce exposed after the backward (round 4).  The bucket size is therefore derived from the gradient volume
    (N_BUCKETS buckets): the render decoder's and the last encoder layer's gradients are reduced while the attention
    kernels of the first layer still run.  DDP rebuilds its buckets in gradient-ready order after the first step
    (find_unused_parameters=False): the FIRST step still uses one bucket, every later one N_BUCKETS.
    Unmeasured on hardware: no multi-GPU node was available to any round so far.
"""
from __future__ import annotations

import os
from typing import Iterable, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

# substrings of parameter names that no forward of the reference ever touches
NEVER_USED = ("proj_q.", "proj_views.", "down_proj.", "ffn_tsa.", "ffn_sca.")


def freeze_unused_parameters(model: nn.Module, n_views: int | None = None) -> list[str]:
    """requires_grad_(False) on parameters that can never receive a gradient; returns their names."""
    frozen = []
    for name, p in model.named_parameters():
        dead = any(t in name for t in NEVER_USED)
        if n_views is not None and "conv_offset_m" in name:
            v = int(name.split("conv_offset_m")[1].split(".")[0])
            dead = dead or v >= n_views
        if dead:
            p.requires_grad_(False)
            frozen.append(name)
    return frozen


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


N_BUCKETS = 4     # gradient buckets per step: the last ones' all-reduce runs under the backward of the earlier layers


def gradient_bytes(model: nn.Module) -> int:
    return sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad)


def bucket_count(ddp: nn.Module) -> int:
    """Gradient buckets the reducer of a DistributedDataParallel module uses right now (1 before the first backward:
    the buckets are rebuilt in gradient-ready order after it)."""
    return len(ddp.reducer._get_zeros_like_grad_buckets())


def wrap_data_parallel(model: nn.Module, local_rank: int | None = None, bucket_cap_mb: float | None = None) -> nn.Module:
    """DistributedDataParallel without an unused-parameter search and with N_BUCKETS gradient buckets (bucket_cap_mb
    None: gradient volume / N_BUCKETS, so that a bucket's all-reduce overlaps the rest of the backward; the reference
    takes PyTorch's 25 MB default, train.py:133-135); identity when world == 1.

    BatchNorm layers (image backbone, render decoder -- none on the hot path) keep the reference's semantics
    (train.py:128-141 converts to SyncBatchNorm before wrapping): on GPU process groups they are converted to
    SyncBatchNorm; where SyncBatchNorm cannot run (CPU / gloo tests) the running statistics are kept identical
    across ranks by broadcasting buffers from rank 0 instead.  A module without BatchNorm has no buffers to sync."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    on_gpu = next(model.parameters()).is_cuda
    has_bn = any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in model.modules())
    sync_bn = has_bn and on_gpu and dist.get_backend() == "nccl"
    if bucket_cap_mb is None:
        bucket_cap_mb = max(gradient_bytes(model) / N_BUCKETS, 1024) / float(1 << 20)
    if sync_bn:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return nn.parallel.DistributedDataParallel(
        model, device_ids=[local_rank] if on_gpu else None, bucket_cap_mb=bucket_cap_mb,
        gradient_as_bucket_view=True, find_unused_parameters=False, broadcast_buffers=has_bn and not sync_bn)


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous batch slice of this rank (B must be divisible by world)."""
    B = t.shape[0]
    if B % world:
        raise ValueError(f"batch {B} is not divisible by world size {world}")
    per = B // world
    return t[rank * per:(rank + 1) * per]
