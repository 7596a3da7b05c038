"""N>1 path with the REAL modules: two ranks (sharing the one GPU of the test box, gloo transport) wrap an
EncoderLayer -- the module that owns the custom autograd Functions of the HIP kernels -- in the data-parallel
wrapper and must reproduce the gradients of one process on the concatenated batch; and `python bench.py --gpus 2`
must launch its own ranks.  RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _layer(S=12, C=64, h=2, D=3, V=2):
    from bevrender_amd import _lib
    from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
    from bevrender_amd.model.encoder import EncoderLayer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_fullsize import ring_rig
    T, K = ring_rig(V, 128, 96)
    proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=128, img_height=96,
                               ori_img_width=128, ori_img_height=96, device="cuda")
    torch.manual_seed(7)
    return EncoderLayer(bev_bound={"X": 20, "Y": 20, "Z": 2}, bev2cmr_projector=proj, n_views=V, bev_feat_shape=S,
                        bev_depth_dim=D, z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1, n_heads=h,
                        stride=1, kernel_size=3, batch_size=1, scale_offset_range=True, drop_path_rate=0.0,
                        precision=_lib.PREC_F32)


def _inputs(B=4, S=12, C=64, V=2):
    g = torch.Generator().manual_seed(11)
    return (torch.randn(B, C, S, S, generator=g), torch.randn(B * V, C, 8, 12, generator=g),
            torch.randn(B, C, S, S, generator=g))


def _loss(layer, q, feat, prev):
    out, _ = layer(q, feat, prev, None, torch.zeros((), dtype=torch.long), None, False)
    return out.square().mean()


def _worker(rank, world, port, state, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from bevrender_amd import parallel
    torch.cuda.set_device(0)
    r, w, _ = parallel.init_distributed("gloo")
    layer = _layer().cuda()
    layer.load_state_dict(state)
    frozen = parallel.freeze_unused_parameters(layer, n_views=2)
    assert any("proj_q" in n for n in frozen)
    q, feat, prev = _inputs()
    V = 2
    qs, ps = parallel.shard_batch(q, rank, world).cuda(), parallel.shard_batch(prev, rank, world).cuda()
    fs = parallel.shard_batch(feat.reshape(q.shape[0], V, *feat.shape[1:]), rank, world).flatten(0, 1).cuda()

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, a, b, c):
            return _loss(self.m, a, b, c)

    net = parallel.wrap_data_parallel(Wrap(layer), 0)
    net(qs, fs, ps).backward()
    torch.cuda.synchronize()
    if rank == 0:
        out_q.put({n: p.grad.cpu().numpy() for n, p in layer.named_parameters() if p.requires_grad})
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_reproduce_single_process_gradients_with_the_hip_modules():
    from bevrender_amd import parallel
    layer = _layer().cuda()
    state = {k: v.detach().cpu().clone() for k, v in layer.state_dict().items()}
    parallel.freeze_unused_parameters(layer, n_views=2)
    q, feat, prev = _inputs()
    _loss(layer, q.cuda(), feat.cuda(), prev.cuda()).backward()
    torch.cuda.synchronize()
    want = {n: p.grad.cpu().numpy() for n, p in layer.named_parameters() if p.requires_grad}
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, state, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    got = out_q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert set(got) == set(want) and len(want) > 20
    scale = max(np.abs(v).max() for v in want.values())
    for k in want:
        # mean over the whole batch == average of the two half-batch means (float atomics reorder sums: 1e-4)
        np.testing.assert_allclose(got[k], want[k], rtol=2e-3, atol=1e-4 * scale, err_msg=k)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver calls it) must start the two
    ranks itself.  Rehearsal mode: both ranks on the one GPU over gloo -- the control flow, not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BEVR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bev", "28", "--batch", "1", "--f32-steps", "0",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["value"] > 0
    assert "REHEARSAL" in rec["data"]
