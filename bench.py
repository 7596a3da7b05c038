#!/usr/bin/env python3
"""Headline benchmark: samples/s, forward+backward, of the BEV-lift + correlation hot path.

Workload (BASELINE.json configs[1] + the correlation head of configs[2]):
  per sample: T=2 frames x V=6 cameras of 256x704 backbone FEATURES (stride 4: 64 x 64 x 176, synthetic
  N(0,1), generated in bf16) through L=2 encoder layers (LPU conv, LN, TSA, MLP, LPU, SCA, MLP) on a
  200x200 BEV grid, C=64, 2 heads, D=5 height bins; frame 0 forward-only/no-grad (history BEV), frame 1
  forward+backward; loss = contrastive ground<->aerial correlation of the flattened BEV against a synthetic
  aerial embedding + a mean-square render proxy; AdamW step.  Batch 4 per GPU, data parallel over N GPUs
  (one process per GPU, gradient all-reduce over RCCL, weak scaling).  The image backbone and render CNN
  are outside the path (left to MIOpen) and are not run.

Prints ONE JSON line (rank 0).  `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either launch it under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU; RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* read from the environment) or run it bare: with no WORLD_SIZE in the environment it starts
that launcher itself as a child process -- before anything touches the GPU -- and exits with the child's code.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense, MI355X_MICROARCH.md chip-level parameters
HBM_PEAK_GBS = 8000.0


def ring_rig(V, img_w, img_h):
    """V cameras on a ring (yaw 360 v / V, pitch 0, 1.5 m up), fx = fy = 0.8 W."""
    R0 = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)
    T, K = [], []
    for v in range(V):
        a = 2 * math.pi * v / V
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        M = np.eye(4)
        M[:3, :3] = Rz @ R0
        M[:3, 3] = (0, 0, 1.5)
        T.append(M)
        K.append(np.array([[0.8 * img_w, 0, img_w / 2, 0], [0, 0.8 * img_w, img_h / 2, 0], [0, 0, 1, 0]], dtype=np.float64))
    return T, K


class LiftBlock(nn.Module):
    """L encoder layers + the correlation head: the unit the metric counts."""

    def __init__(self, S, C, heads, D, V, L, img_w, img_h, precision, device):
        super().__init__()
        from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
        from bevrender_amd.model.encoder import EncoderLayer
        from bevrender_amd.loss.contrastive_loss import ContrastiveLoss
        T, K = ring_rig(V, img_w, img_h)
        proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=img_w, img_height=img_h,
                                   ori_img_width=img_w, ori_img_height=img_h, device=device)
        self.layers = nn.ModuleList([
            EncoderLayer(bev_bound={"X": 50, "Y": 50, "Z": 2}, bev2cmr_projector=proj, n_views=V, bev_feat_shape=S,
                         bev_depth_dim=D, z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1,
                         n_heads=heads, stride=1, kernel_size=3, batch_size=1, scale_offset_range=True,
                         drop_path_rate=0.0, precision=precision) for _ in range(L)])
        self.bev_embedding = nn.Embedding(S * S, C)
        self.S, self.C = S, C
        self.loss = ContrastiveLoss()
        # reference init (model/bevrender.py:152-172)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
            elif isinstance(m, nn.Embedding):
                nn.init.uniform_(m.weight)
        # parameters the reference constructs but never uses get no gradient: keep them out of the all-reduce
        from bevrender_amd.parallel import freeze_unused_parameters
        freeze_unused_parameters(self, n_views=V)

    def encode(self, feats, prev_bev):
        B = feats.shape[0] // self.layers[0].spatial_cross_attn.num_views
        x = self.bev_embedding.weight.t().reshape(1, self.C, self.S, self.S).expand(B, -1, -1, -1)
        vt = torch.zeros((), dtype=torch.long)
        for layer in self.layers:
            x, _ = layer(x, feats, prev_bev, None, vt, None, False)
        return x

    def forward(self, feats_hist, feats_cur, map_emb):
        with torch.no_grad():
            prev = self.encode(feats_hist, None)
        bev = self.encode(feats_cur, prev)
        corr = self.loss.get_loss(bev.flatten(1), map_emb)
        return corr + bev.square().mean()


def attn_flops(kind, geom):
    """algorithmic MFMA flops of one attention launch (2 flop per MAC, head_dim 32)."""
    pairs = geom.n_prob * geom.heads * (geom.S * geom.S) * geom.N
    n_mm = {"bevr_attn_fwd": 2, "bevr_attn_bwd_q": 3, "bevr_attn_bwd_k": 4}[kind]
    return 2.0 * 32 * pairs * n_mm


def cpu_baseline(seconds_budget=20.0):
    """The oracle (CPU restatement of the reference, materialised formulation) on the host cores:
    the same unit of work (L=2 layers, T=2 frames, V=6 views, fwd+bwd on the last frame) at a BEV side the
    materialised (M x N) tensors fit in host RAM."""
    from oracle import bevrender_oracle as O
    S, C, h, D, V, L = 28, 64, 2, 5, 6, 2
    torch.manual_seed(15213)
    # the threads this process may actually use: a GPU box hands one GPU's job a 16-core share of a much
    # larger host, and os.cpu_count() would oversubscribe it by an order of magnitude
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("BEVR_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)

    def mk_params():
        p = {}
        def conv(name, co, ci, k, bias=True):
            p[name + ".weight"] = (torch.randn(co, ci, k, k) / math.sqrt(ci * k * k)).requires_grad_(True)
            if bias:
                p[name + ".bias"] = torch.zeros(co, requires_grad=True)
        def ln(name, c):
            p[name + ".norm.weight"] = torch.ones(c, requires_grad=True)
            p[name + ".norm.bias"] = torch.zeros(c, requires_grad=True)
        ln("layer_norm", C)
        for mlp in ("tsa_mlp", "sca_mlp"):
            conv(mlp + ".linear1.0", 4 * C, C, 1); conv(mlp + ".linear2.0", C, 4 * C, 1); conv(mlp + ".dwc", 4 * C, 1, 3)
        conv("tsa_local_percept_unit", C, 1, 3); conv("sca_local_percept_unit", C, 1, 3)
        t = "temporal_self_attn.temporal_deform_attn."
        conv(t + "conv_offset.0", C, 1, 3); ln(t + "conv_offset.1", C); conv(t + "conv_offset.3", 2, C, 1, False)
        s = "spatial_cross_attn.spatial_deform_attn."
        for v in range(V):
            conv(s + f"conv_offset_m{v}.0", C * D, 1, 1); ln(s + f"conv_offset_m{v}.1", C * D)
            conv(s + f"conv_offset_m{v}.3", D, C * D, 1, False)
        for pre, cin in ((t, C), (s, C)):
            conv(pre + "proj_k", C, C, 1); conv(pre + "proj_v", C, C, 1)
        conv(t + "proj_out", C, C, 1); conv(s + "proj_out", C, V * C, 1)
        p[t + "rpe_table"] = (torch.randn(h, 2 * S - 1, 2 * S - 1) * 0.01).requires_grad_(True)
        p[s + "rpe_table"] = (torch.randn(h, 2 * S - 1, 2 * S * D - 1) * 0.01).requires_grad_(True)
        return p

    params = [mk_params() for _ in range(L)]
    T, K = ring_rig(V, 704, 256)
    pts = O.sample_3d_points({"X": 50, "Y": 50, "Z": 2}, S, D, -1.0)
    ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, 704, 256, 704, 256), 1)
    q0 = torch.rand(1, C, S, S)
    feats = [torch.randn(V, C, 64, 176) for _ in range(2)]

    def run(x, f, prev):
        for p in params:
            x = O.encoder_layer_forward(p, x, f, prev, ref, n_heads=h, n_groups=1, depth_dim=D, n_views=V,
                                        kernel_size=3, stride=1)
        return x

    def one_sample():
        with torch.no_grad():
            prev = run(q0, feats[0], None)
        out = run(q0, feats[1], prev)
        out.square().mean().backward()

    t_w = time.perf_counter()
    one_sample()  # warm-up (allocator, thread pool)
    print(f"[bench] cpu_baseline warm-up sample: {time.perf_counter() - t_w:.1f} s on {cores} threads",
          file=sys.stderr, flush=True)
    n, t0 = 0, time.perf_counter()
    while True:
        one_sample()
        n += 1
        dt = time.perf_counter() - t0
        print(f"[bench] cpu_baseline: {n} samples, {dt:.1f} s", file=sys.stderr, flush=True)
        if dt > seconds_budget or n >= 50:
            break
    return {"value": n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU restatement, materialised MxN formulation, fp32) on the same unit of work at "
                      f"BEV side S={S} (M=784, N_sca=1960 per view; cfg2 is S=200 with 2600x more query-key pairs), "
                      f"V=6, L=2, T=2, fwd+bwd, {n} samples in {dt:.1f} s; NOT extrapolated"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU")
    ap.add_argument("--bev", type=int, default=200, help="BEV side (200 = BASELINE config; smaller for debugging)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (importing torch does
        # not), the ranks are fresh child processes (never an exec), rank 0's JSON line is the children's stdout.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    from bevrender_amd import parallel
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BEVR_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- exercises the multi-rank control flow (DDP wrap,
    # barriers, max-over-ranks timing, rank-0 report) on a one-GPU box.  Never a measurement.
    rehearsal = os.environ.get("BEVR_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    rank, world, local_rank = parallel.init_distributed("gloo" if rehearsal else "nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    dev = torch.device("cuda", dev_index)

    from bevrender_amd import _lib, ops
    S, C, heads, D, V, L, B = args.bev, 64, 2, 5, 6, 2, args.batch
    img_w, img_h = 704, 256
    Hi, Wi = img_h // 4, img_w // 4
    torch.manual_seed(15213 + rank)
    model = LiftBlock(S, C, heads, D, V, L, img_w, img_h, args.precision, dev).to(dev)
    # identical initial weights on every rank (DDP broadcasts rank 0's); one flat 25 MB bucket holds all grads
    net = parallel.wrap_data_parallel(model, dev_index)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)

    gen = torch.Generator(device=dev).manual_seed(15213 + rank)
    feats = [torch.randn(B * V, C, Hi, Wi, device=dev, dtype=torch.bfloat16, generator=gen).float() for _ in range(2)]
    map_emb = torch.nn.functional.normalize(torch.randn(B, C * S * S, device=dev, generator=gen), dim=1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = net(feats[0], feats[1], map_emb)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.KERNEL_TIMER.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ktimes = ops.KERNEL_TIMER.stop()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert math.isfinite(loss.item())

    if rank == 0:
        total_samples = B * world * args.steps
        # dominant kernel by accumulated device time inside the timed region (HIP events on its stream)
        attn = {k: v for k, v in ktimes.items() if k.startswith("bevr_attn")}
        dom = max(attn, key=lambda k: attn[k]["ms"]) if attn else None
        roof = None
        if dom:
            rec = attn[dom]
            avg_ms = rec["ms"] / rec["n"]
            flops = rec["flops"] / rec["n"]
            ach = flops / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            # HBM bytes per launch of that kernel: measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this
            # very command (separate passes, gfx950 correction applied; see profiles/r01_traffic.json), not live
            traffic = None
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
            if S == 200 and B == 4 and args.precision == "bf16" and os.path.exists(tpath):
                with open(tpath) as fh:
                    traffic = json.load(fh)["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
            pairs = rec["flops"] / rec["n"] / (2.0 * 32 * {"bevr_attn_fwd": 2, "bevr_attn_bwd_q": 3, "bevr_attn_bwd_k": 4}[dom])
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 5), "traffic": traffic, "avg_ms": round(avg_ms, 3),
                    "launches": rec["n"], "pair_ops_per_s": round(pairs / (avg_ms * 1e-3), 0),
                    "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 2) for k, v in sorted(ktimes.items())}}
        out = {
            "metric": "samples/sec fwd+bwd, 6-cam 256x704 BEV-lift+corr",
            "value": round(total_samples / dt, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": f"cfg2+corr: 6-cam 256x704 features (64x64x176), {S}x{S} BEV, C=64 h=2 D=5, L=2 "
                                   f"encoder layers (TSA+SCA), T=2 (1 no-grad history frame + 1 fwd+bwd), "
                                   f"contrastive correlation head, AdamW; backbone/render CNN excluded",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (bench contract)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
