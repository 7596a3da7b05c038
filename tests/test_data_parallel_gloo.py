"""N>1 path on CPU: two gloo ranks, each with half the batch, must produce the gradients of one process
on the whole batch (mean-reduced loss), with the never-used parameters frozen instead of searched for."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from bevrender_amd import parallel
from bevrender_amd.model.model_utils import LayerNormProxy, TransformerMLPWithConv


class TinyGlue(nn.Module):
    """Encoder-layer glue without the GPU-only attention ops (those are per-sample independent, so the
    data-parallel contract is the same): LPU conv + shared LN + MLP, plus reference-style dead parameters."""

    def __init__(self, C=8):
        super().__init__()
        self.layer_norm = LayerNormProxy(C)
        self.lpu = nn.Conv2d(C, C, 3, 1, 1, groups=C)
        self.mlp = TransformerMLPWithConv(C, 2, 0.0)
        self.proj_q = nn.Conv2d(C, C, 1)          # never used, as in the reference
        self.ffn_tsa = nn.Linear(4, 4)            # never used

    def forward(self, x):
        x = x + self.lpu(x)
        x = self.mlp(self.layer_norm(x)) + x
        return x.square().mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, state, x, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = TinyGlue()
    model.load_state_dict(state)
    frozen = parallel.freeze_unused_parameters(model)
    assert any("proj_q" in n for n in frozen) and any("ffn_tsa" in n for n in frozen)
    net = parallel.wrap_data_parallel(model)
    assert parallel.bucket_count(net) == 1          # before the first backward: one bucket (DDP's initial assignment)
    xs = parallel.shard_batch(x, rank, world)
    for _ in range(2):                              # DDP rebuilds its buckets in gradient-ready order after step 1
        net.zero_grad(set_to_none=True)
        net(xs).backward()
    net.zero_grad(set_to_none=True)
    loss = net(xs)
    loss.backward()
    # gradient volume / N_BUCKETS per bucket: the all-reduce of the layers that finish first runs under the backward of the
    # rest (train.py:133-135 of the reference takes the 25 MB default = one bucket after the last gradient)
    assert parallel.bucket_count(net) >= 2, parallel.bucket_count(net)
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
    if rank == 0:
        out_q.put({k: v.numpy() for k, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process():
    torch.manual_seed(0)
    ref = TinyGlue()
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    x = torch.randn(4, 8, 6, 6)
    parallel.freeze_unused_parameters(ref)
    ref(x).backward()
    want = {n: p.grad.numpy() for n, p in ref.named_parameters() if p.requires_grad}

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, state, x, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert set(got) == set(want)
    for k in want:
        # mean over the full batch == average of the two half-batch means
        np.testing.assert_allclose(got[k], want[k], rtol=1e-5, atol=1e-7, err_msg=k)


class TinyBN(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 1)
        self.bn = nn.BatchNorm2d(4)

    def forward(self, x):
        return self.bn(self.conv(x)).square().mean()


def _bn_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_distributed("gloo")
    torch.manual_seed(0)
    net = parallel.wrap_data_parallel(TinyBN())
    assert net.broadcast_buffers            # CPU ranks cannot run SyncBatchNorm: buffers follow rank 0 instead
    for step in range(2):
        x = torch.randn(4, 3, 5, 5, generator=torch.Generator().manual_seed(100 * rank + step))
        net(x).backward()
    net(torch.zeros(4, 3, 5, 5))            # a forward broadcasts rank 0's running statistics
    out_q.put((rank, net.module.bn.running_mean.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_batchnorm_statistics_stay_identical_across_ranks():
    """train.py:128-141 converts to SyncBatchNorm; the wrapper does that on GPU/RCCL groups and keeps the running
    statistics identical by buffer broadcast where SyncBatchNorm cannot run (this CPU test)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got[0], got[1])
    assert np.abs(got[0]).max() > 0


def _gather_worker(rank, world, port, w0, x, y, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_distributed("gloo")
    from bevrender_amd.retrieval import all_gather_embeddings
    from oracle import bevrender_oracle as O
    lin = nn.Linear(w0.shape[1], w0.shape[0], bias=False)
    with torch.no_grad():
        lin.weight.copy_(w0)
    net = parallel.wrap_data_parallel(lin)
    cam = net(parallel.shard_batch(x, rank, world))
    mp = parallel.shard_batch(y, rank, world)
    loss = O.contrastive_loss(all_gather_embeddings(cam), all_gather_embeddings(mp))
    loss.backward()
    if rank == 0:
        out_q.put((loss.item(), lin.weight.grad.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gathered_negatives_give_the_global_batch_gradient():
    """f3: with the embeddings of every rank gathered (differentiably), each rank's retrieval loss is the loss of the
    GLOBAL batch, and the data-parallel mean of the per-rank gradients is its gradient."""
    from oracle import bevrender_oracle as O
    torch.manual_seed(3)
    w0 = torch.randn(12, 20) * 0.3
    x = torch.randn(6, 20)
    y = x @ w0.t() + 0.5 * torch.randn(6, 12)
    wr = w0.clone().requires_grad_(True)
    want = O.contrastive_loss(x @ wr.t(), y)
    want.backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, w0, x, y, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss, grad = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert abs(loss - want.item()) < 1e-6
    np.testing.assert_allclose(grad, wr.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_shard_batch_and_world1_passthrough():
    t = torch.arange(8).reshape(8, 1)
    assert parallel.shard_batch(t, 1, 4).flatten().tolist() == [2, 3]
    with pytest.raises(ValueError):
        parallel.shard_batch(t, 0, 3)
    m = TinyGlue()
    assert parallel.wrap_data_parallel(m) is m
