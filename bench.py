#!/usr/bin/env python3
"""Headline benchmark: samples/s, forward+backward, of the BEV-lift + correlation hot path.

Workload (BASELINE.json configs[2] = configs[1] + the correlation head):
  per sample: T=2 frames x V=6 cameras of 256x704 backbone FEATURES (stride 4: 64 x 64 x 176, synthetic
  N(0,1), generated in bf16) through L=2 encoder layers (LPU conv, LN, TSA, MLP, LPU, SCA, MLP) on a
  200x200 BEV grid, C=64, 2 heads, D=5 height bins; frame 0 forward-only/no-grad (history BEV), frame 1
  forward+backward; loss = contrastive + lifted-structure ground<->aerial correlation of the flattened BEV
  against a synthetic aerial embedding + a mean-square render proxy; AdamW step.  Batch 8 per GPU (config 3;
  `--batch 4` is config 2's batch), data parallel over N GPUs (one process per GPU, gradient all-reduce over
  RCCL, weak scaling).  The image backbone and render CNN are outside the path (left to MIOpen) and are not run.

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

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "bf16x3": 2500.0 / 3}
# dense, MI355X_MICROARCH.md chip-level parameters; bf16x3: three bf16 MFMAs per product
HBM_PEAK_GBS = 8000.0
N_CU, CLOCK_HZ = 256, 2.4e9                          # for the LDS-pipe view (roofline_lds)


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

    def __init__(self, S, C, heads, D, V, L, img_w, img_h, precision, device, rig=None, bound=None):
        super().__init__()
        from bevrender_amd.model.bev_cmr_proj import BEV2CameraProjector
        from bevrender_amd.model.encoder import EncoderLayer
        from bevrender_amd.loss.contrastive_loss import ContrastiveLoss
        from bevrender_amd.loss.lift_loss import LiftedStructureLoss
        T, K = rig if rig is not None else ring_rig(V, img_w, img_h)
        proj = BEV2CameraProjector(imu_to_rgb={0: T}, K={0: K}, vehicle_type_code=0, img_width=img_w, img_height=img_h,
                                   ori_img_width=img_w, ori_img_height=img_h, device=device)
        self.layers = nn.ModuleList([
            EncoderLayer(bev_bound=bound or {"X": 50, "Y": 50, "Z": 2}, bev2cmr_projector=proj, n_views=V, bev_feat_shape=S,
                         bev_depth_dim=D, z_shift=-1.0, dim_embed=C, expansion=4, stage_idx=0, n_groups=1,
                         n_heads=heads, stride=1, kernel_size=3, batch_size=1, scale_offset_range=True,
                         drop_path_rate=0.0, precision=precision) for _ in range(L)])
        self.bev_embedding = nn.Embedding(S * S, C)
        self.S, self.C = S, C
        self.loss = ContrastiveLoss()
        self.lift = LiftedStructureLoss()
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

    def forward(self, feats_hist, feats_cur, map_emb, n_hist=1):
        """n_hist no-grad history frames chained through TSA's prev_bev (the reference's recurrence over frames), then
        the current frame forward + backward.  The default T = 2 is one history frame."""
        with torch.no_grad():
            prev = None
            for _ in range(n_hist):
                prev = self.encode(feats_hist, prev)
        bev = self.encode(feats_cur, prev)
        emb = bev.flatten(1)
        corr = self.loss.get_loss(emb, map_emb) + self.lift.get_loss(emb, map_emb)   # config 3: contrastive + lifted
        return corr + bev.square().mean()


# matrix products per (query, key) pair (2 * 32 flop each) that the flash-style formulation RUNS in each entry point:
# QK^T + PV forward; S, dP, dQ on the query side; S, dP, dV, dK on the key side -- 9 per forward + backward.  This is what
# ops.py attaches to a launch and what the per-kernel `roofline` lines price (the work a kernel of that kind does).
# SURVEY.md 8d counts 7 (backward = 2.5 x forward: S is recomputed once, not twice): ALG_MATMUL splits the backward's 5
# evenly between the two sides, and `all_attention_aggregate` is on that accounting, with the 9-product figure kept as
# `executed`.  The tap / cell kernels' bias and table-gradient products are extra work their formulation spends, and
# the tap kernels run FEWER matrix flops than these counts (contraction 12 taps instead of 32 channels): the counts
# are the algorithm's, not the instruction stream's.
N_MATMUL = {"bevr_attn_fwd": 2, "bevr_attn_bwd_q": 3, "bevr_attn_bwd_k": 4, "bevr_attn_gather_fwd": 2,
            "bevr_attn_slab_bwd_q": 3,
            "bevr_attn_cell_fwd": 2, "bevr_attn_cell_bwd_q": 3, "bevr_attn_cell_bwd_k": 4,
            "bevr_attn_tap_fwd": 2, "bevr_attn_tap_bwd_q": 3, "bevr_attn_tap_bwd_k": 4}
ALG_MATMUL = {k: (2.0 if k.endswith("fwd") else 2.5) for k in N_MATMUL}
# SURVEY.md 8d, algorithmic HBM bytes per sample of the block (L = 2, T = 2): 374 MB + the correlation head's 82 MB
BLOCK_HBM_BYTES_PER_SAMPLE = {4: 374e6, 8: 456e6}


def csrc_sha():
    """Revision of the kernels: sha1 over the HIP sources (what profiles/*_traffic.json is tied to)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "bevrender_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "bevrender_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _oracle_params(C, h, D, V, S):
    """Random parameters of one EncoderLayer under the reference's state_dict names (oracle.encoder_layer_forward)."""
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
    sp = "spatial_cross_attn.spatial_deform_attn."
    for v in range(V):
        conv(sp + f"conv_offset_m{v}.0", C * D, 1, 1); ln(sp + f"conv_offset_m{v}.1", C * D)
        conv(sp + f"conv_offset_m{v}.3", D, C * D, 1, False)
    for pre in (t, sp):
        conv(pre + "proj_k", C, C, 1); conv(pre + "proj_v", C, C, 1)
    conv(t + "proj_out", C, C, 1); conv(sp + "proj_out", C, V * C, 1)
    p[t + "rpe_table"] = (torch.randn(h, 2 * S - 1, 2 * S - 1) * 0.01).requires_grad_(True)
    p[sp + "rpe_table"] = (torch.randn(h, 2 * S - 1, 2 * S * D - 1) * 0.01).requires_grad_(True)
    return p


def cpu_baseline(seconds_budget=12.0):
    """The oracle (CPU restatement of the reference, its materialised M x N formulation, fp32) on the host cores of
    this box, on the unit of work the GPU number counts (per sample: L=2 encoder layers, T=2 frames, fwd+bwd on the
    last frame), as SURVEY.md section 8d specifies:
      value : BASELINE config 1 exactly -- 1 camera, 128x128 image (16x16 features), 50x50 BEV;
      cfg2_geometry_s56 : config 2's geometry (6-camera ring, 64x176 features) at the largest BEV side whose
          materialised tensors fit host memory in reasonable time (S=56), one layer timed forward and
          forward+backward, assembled into the per-sample time and EXTRAPOLATED to S=200 by the M*N ratio (labelled)."""
    from oracle import bevrender_oracle as O
    torch.manual_seed(15213)
    # the threads this process may actually use: a GPU box hands one GPU's job a 16-core share of a much
    # larger host, and os.cpu_count() would oversubscribe it by an order of magnitude
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("BEVR_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    C, h, D, L = 64, 2, 5, 2

    def layer_fn(S, V, T, K, img_w, img_h, Hi, Wi, bound):
        params = [_oracle_params(C, h, D, V, S) for _ in range(L)]
        pts = O.sample_3d_points(bound, S, D, -1.0)
        ref = O.sca_reference_points(O.bev_grid_to_camera(pts, T, K, img_w, img_h, img_w, img_h), 1)
        q0 = torch.rand(1, C, S, S)
        feats = [torch.randn(V, C, Hi, Wi) for _ in range(2)]

        def run(x, f, prev, n_layers=L):
            for p in params[:n_layers]:
                x = O.encoder_layer_forward(p, x, f, prev, ref, n_heads=h, n_groups=1, depth_dim=D, n_views=V,
                                            kernel_size=3, stride=1)
            return x
        return run, q0, feats

    # ---- config 1 exactly ------------------------------------------------------------------------------
    T1 = np.eye(4); T1[:3, :3] = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64); T1[:3, 3] = (0, 0, 1.5)
    K1 = np.array([[100, 0, 64, 0], [0, 100, 64, 0], [0, 0, 1, 0]], dtype=np.float64)
    run1, q1, f1 = layer_fn(50, 1, [T1], [K1], 128, 128, 16, 16, {"X": 20, "Y": 10, "Z": 2})

    def one_sample():
        with torch.no_grad():
            prev = run1(q1, f1[0], None)
        run1(q1, f1[1], prev).square().mean().backward()

    t_w = time.perf_counter()
    one_sample()  # warm-up (allocator, thread pool)
    print(f"[bench] cpu_baseline cfg1 warm-up sample: {time.perf_counter() - t_w:.1f} s on {cores} threads",
          file=sys.stderr, flush=True)
    n, t0 = 0, time.perf_counter()
    while True:
        one_sample()
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or n >= 50:
            break
    print(f"[bench] cpu_baseline cfg1: {n} samples, {dt:.1f} s", file=sys.stderr, flush=True)
    out = {"value": n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"oracle (CPU restatement of the reference, materialised MxN formulation, fp32) on BASELINE config 1 "
                     f"exactly: 1 camera, 128x128 image (16x16 features), 50x50 BEV (M=2500, N_tsa=2500, N_sca=6250), "
                     f"L=2, T=2, fwd+bwd; {n} samples in {dt:.1f} s"}

    # ---- config 2's geometry at S=56, one layer, extrapolated ---------------------------------------------------
    S2, V2 = 56, 6
    T2, K2 = ring_rig(V2, 704, 256)
    run2, q2, f2 = layer_fn(S2, V2, T2, K2, 704, 256, 64, 176, {"X": 50, "Y": 50, "Z": 2})
    ta = time.perf_counter()
    with torch.no_grad():
        prev = run2(q2, f2[0], None, 1)
    t_f = time.perf_counter() - ta
    ta = time.perf_counter()
    run2(q2, f2[1], prev, 1).square().mean().backward()
    t_fb = time.perf_counter() - ta
    per_sample = L * (t_f + t_fb)                         # T = 2: one no-grad frame + one fwd+bwd frame, L layers
    ratio = (200 ** 2 * (200 // 2) * 200 * D) / float(S2 ** 2 * (S2 // 2) * S2 * D)     # M * N_sca, S=200 over S=56
    print(f"[bench] cpu_baseline cfg2 geometry S={S2}: layer fwd {t_f:.1f} s, fwd+bwd {t_fb:.1f} s", file=sys.stderr, flush=True)
    out["cfg2_geometry_s56"] = {
        "samples_per_s_at_s56": 1.0 / per_sample, "layer_fwd_s": round(t_f, 2), "layer_fwd_bwd_s": round(t_fb, 2),
        "extrapolated_samples_per_s_at_s200": 1.0 / (per_sample * ratio), "extrapolation": f"EXTRAPOLATED by the M*N ratio "
        f"({ratio:.0f}x); the materialised formulation cannot run S=200 (16 GB per head, sample and tensor)",
        "sample": f"6-camera ring, 64x176 features, S={S2} (M={S2 * S2}, N_sca={S2 // 2 * S2 * D} per view), ONE encoder layer "
                  f"timed forward and forward+backward once each; per sample = L * (fwd + fwd+bwd)"}
    return out


def gpu_same_workloads(dev):
    """The GPU path on the two workloads the CPU baseline times (nothing put the two on the SAME workload):
    BASELINE config 1 exactly (1 camera, 16x16 features, 50x50 BEV, batch 2) and config 2's geometry at S = 56 (6-camera
    ring, 64x176 features, batch 1), whole LiftBlock steps (L = 2, T = 2, correlation head, AdamW), bf16 operands and the
    exact-f32 mode (the CPU baseline's arithmetic)."""
    C, heads, D, L = 64, 2, 5, 2
    T1 = np.eye(4); T1[:3, :3] = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64); T1[:3, 3] = (0, 0, 1.5)
    K1 = np.array([[100, 0, 64, 0], [0, 100, 64, 0], [0, 0, 1, 0]], dtype=np.float64)
    cases = {"cfg1": dict(S=50, V=1, img=(128, 128), feat=(16, 16), B=2, rig=([T1], [K1]), bound={"X": 20, "Y": 10, "Z": 2}),
             "cfg2_geometry_s56": dict(S=56, V=6, img=(704, 256), feat=(64, 176), B=1, rig=None, bound=None)}
    res = {}
    for name, c in cases.items():
        res[name] = {}
        for mode in ("bf16", "f32"):
            torch.manual_seed(15213)
            m = LiftBlock(c["S"], C, heads, D, c["V"], L, c["img"][0], c["img"][1], mode, dev, rig=c["rig"], bound=c["bound"]).to(dev)
            o = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4)
            f = [torch.randn(c["B"] * c["V"], C, *c["feat"], device=dev, dtype=torch.bfloat16)
                 .contiguous(memory_format=torch.channels_last) for _ in range(2)]
            me = torch.nn.functional.normalize(torch.randn(c["B"], C * c["S"] ** 2, device=dev), dim=1)

            def st():
                o.zero_grad(set_to_none=True)
                m(f[0], f[1], me, 1).backward()
                o.step()
            st()
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while n < 3 or (time.perf_counter() - t0 < 1.0 and n < 50):
                st()
                n += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name][mode] = {"samples_per_s": round(c["B"] * n / dt, 3), "steps": n, "batch": c["B"]}
            del m, o
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (8 = BASELINE config 3, 4 = config 2)")
    ap.add_argument("--f32-steps", type=int, default=5,
                    help="also time this many steps in the fp32-tolerance mode (split-bf16 products, bf16x3) and one in the "
                         "exact-f32 MFMA mode (rank 0, N=1; 0 = skip)")
    ap.add_argument("--bev", type=int, default=200, help="BEV side (200 = BASELINE config; smaller for debugging)")
    ap.add_argument("--frames", type=int, default=2, help="temporal frames T: T - 1 no-grad history frames + the current one")
    ap.add_argument("--img", default="704x256", help="camera image WxH (features are 1/4 of it); config 5: 1408x512")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "f16", "bf16x3"])
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
    img_w, img_h = (int(v) for v in args.img.lower().split("x"))
    Hi, Wi = img_h // 4, img_w // 4
    torch.manual_seed(15213 + rank)
    model = LiftBlock(S, C, heads, D, V, L, img_w, img_h, args.precision, dev).to(dev)
    # identical initial weights on every rank (DDP broadcasts rank 0's); gradient buckets of a quarter of the gradient
    # volume (parallel.wrap_data_parallel): the all-reduces of the late layers run under the early layers' backward
    net = parallel.wrap_data_parallel(model, dev_index)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)

    gen = torch.Generator(device=dev).manual_seed(15213 + rank)
    # backbone features as the backbone hands them over: channels-last (the sampler's layout, no copy)
    # bf16 in HBM (north_star: bf16 input); the sampler reads them as they are (csrc/sample.hip, bevr_sample_*_bf16)
    feats = [torch.randn(B * V, C, Hi, Wi, device=dev, dtype=torch.bfloat16, generator=gen)
             .contiguous(memory_format=torch.channels_last) for _ in range(2)]
    map_emb = torch.nn.functional.normalize(torch.randn(B, C * S * S, device=dev, generator=gen), dim=1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = net(feats[0], feats[1], map_emb, args.frames - 1)
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
        # HBM bytes per launch: measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this very command (separate
        # passes, gfx950 correction applied) by tools/measure_traffic.sh, which writes profiles/r05_traffic.json together
        # with the kernel sources' sha; only attached when that file was measured at this batch / BEV side / precision
        # AND on these very kernel sources (a stale file is not reported as measured traffic: ADVICE r02)
        traffic_db, traffic_note = {}, "no profiles/r05_traffic.json for this batch / BEV side / precision"
        tpath = os.path.join(ROOT, "profiles", "r05_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("batch") == B and tj.get("bev") == S and tj.get("precision") == args.precision:
                if tj.get("csrc_sha") == csrc_sha():
                    traffic_db, traffic_note = tj["kernels"], f"measured at csrc_sha {tj.get('csrc_sha')}"
                else:
                    traffic_note = (f"profiles/r05_traffic.json was measured at csrc_sha {tj.get('csrc_sha')}, the kernels "
                                    f"are at {csrc_sha()}: not attached")
        if dom:
            rec = attn[dom]
            avg_ms = rec["ms"] / rec["n"]
            flops = rec["flops"] / rec["n"]
            ach = flops / (avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            pairs = rec["flops"] / rec["n"] / (2.0 * 32 * N_MATMUL[dom])
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 5), "traffic": traffic_db.get(dom, {}).get("hbm_bytes_per_launch"),
                    "traffic_note": traffic_note, "avg_ms": round(avg_ms, 3), "launches": rec["n"],
                    "pair_ops_per_s": round(pairs / (avg_ms * 1e-3), 0),
                    "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 2) for k, v in sorted(ktimes.items())},
                    "all_attention": {k: {"avg_ms": round(v["ms"] / v["n"], 3),
                                          "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak, 5)}
                                      for k, v in sorted(attn.items())}}
            # every attention launch of the timed region together, on SURVEY 8d's accounting (7 products per pair forward +
            # backward); `executed`: the 9 products the two-sided backward runs (S and dP on both sides)
            tot_ms = sum(v["ms"] for v in attn.values())
            tot_exec = sum(v["flops"] for v in attn.values())
            tot_alg = sum(v["flops"] / N_MATMUL[k] * ALG_MATMUL[k] for k, v in attn.items())
            roof["all_attention_aggregate"] = {"achieved": round(tot_alg / (tot_ms * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                                               "frac": round(tot_alg / (tot_ms * 1e-3) / 1e12 / peak, 5),
                                               "accounting": "SURVEY 8d: 2 products per pair forward, 5 backward",
                                               "executed": {"achieved": round(tot_exec / (tot_ms * 1e-3) / 1e12, 3),
                                                            "frac": round(tot_exec / (tot_ms * 1e-3) / 1e12 / peak, 5)},
                                               "ms_per_step": round(tot_ms / args.steps, 2)}
            # per call: TSA (square table) and SCA (the scattered keys on the region kernels, the pinned ones on the tap
            # kernels) separately -- kernel_ms_per_step lumps them
            roof["kernel_ms_per_step_by_call"] = {k: {"ms_per_step": round(v["ms"] / args.steps, 2), "launches": v["n"],
                                                      "avg_ms": round(v["ms"] / v["n"], 3),
                                                      "Tpairs_per_s": round(v["flops"] / (2.0 * 32 * N_MATMUL[k.split("[")[0]]) / (v["ms"] * 1e-3) / 1e12, 3)}
                                                  for k, v in sorted(ops.KERNEL_TIMER.by_tag.items()) if k.startswith("bevr_attn")}
        # What actually paces the attention kernels is per-pair work on the LDS pipe (bias taps, per-key constants, the
        # table-gradient atomics), not the matrix cores.  Secondary, clearly separate from
        # `roofline`: the pair rate against the rate at which one CU's LDS pipe could issue each kernel's LDS
        # instructions back to back (clk per wave and key row = per 64 pairs, from the measured instruction costs in
        # profiles/r02_lds_*.txt; bf16 mode).
        LDS_CLK_PER_KEYROW = {"bevr_attn_fwd": 10.0,     # taps 4.8 + constants (b64 + b32 broadcast) 5.2
                              "bevr_attn_bwd_q": 22.6,   # + two ds_add_u64 12.6
                              "bevr_attn_slab_bwd_q": 22.6,   # the same five instructions per key row and half (no window moves)
                              "bevr_attn_bwd_k": 6.0}    # per-lane tap gathers: 3 reads per 4 rows and table column
        roof_lds = []
        if args.precision == "bf16":
            for k, v in sorted(attn.items()):
                if k not in LDS_CLK_PER_KEYROW:      # the cell kernels are not paced by per-pair LDS work
                    continue
                prs = v["flops"] / (2.0 * 32 * N_MATMUL[k]) / (v["ms"] * 1e-3)
                peak_prs = N_CU * CLOCK_HZ * 64.0 / LDS_CLK_PER_KEYROW[k]
                roof_lds.append({"bound": "lds", "kernel": k, "achieved": round(prs, 0), "peak": round(peak_prs, 0),
                                 "unit": "pairs/s", "frac": round(prs / peak_prs, 4),
                                 "lds_clk_per_wave_keyrow": LDS_CLK_PER_KEYROW[k]})
        # the HBM-bound kernels of the path: bilinear feature sampling, forward and backward scatter.
        # achieved = algorithmic (compulsory) bytes / launch time: feature map once + one row and one position per key
        # (forward); + the map's gradient once (backward).  `traffic` = what the PMC counters saw (the backward's
        # scatter is memory-side float atomics, 4 taps per key)
        roof_hbm = []
        # bevr_kv_project: sampling + K | V projection + operand packing in one pass (feature map in, packed operands out)
        # bevr_corr_fwd / _bwd: the ground <-> aerial correlation of the retrieval losses (train.py:551-572,
        # loss/contrastive_loss.py:10-19): at B = 8 a (16 x 2.56 M) embedding matrix against itself -- the Gram on the f32
        # matrix cores, HBM-bound (82 MB per operand set read once; the backward reads it again and writes the
        # gradient once)
        # bevr_merge_views_fwd / _bwd: the attention output into proj_out's layout (+ the two segments' softmax merge)
        for k in ("bevr_kv_project", "bevr_sample_fwd", "bevr_sample_bwd", "bevr_corr_fwd", "bevr_corr_bwd",
                  "bevr_merge_views_fwd", "bevr_merge_views_bwd"):
            if k in ktimes and ktimes[k]["ms"] > 0:
                v = ktimes[k]
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                roof_hbm.append({"bound": "hbm", "kernel": k, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic_db.get(k, {}).get("hbm_bytes_per_launch"),
                                 "avg_ms": round(v["ms"] / v["n"], 3), "max_ms": round(v["max_ms"], 3), "launches": v["n"],
                                 "algorithmic_bytes_per_launch": round(v["bytes"] / v["n"])})
        out = {
            "metric": "samples/sec fwd+bwd, 6-cam 256x704 BEV-lift+corr",
            "value": round(total_samples / dt, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": (f"cfg{3 if B == 8 else 2}{'' if B == 8 else '+corr'}"
                                    if (S, args.frames, args.img) == (200, 2, "704x256") else "non-default shape") +
                                   f": 6-cam {img_h}x{img_w} features (64x{Hi}x{Wi}), "
                                   f"{S}x{S} BEV, C=64 h=2 D=5, L=2 encoder layers (TSA+SCA), T={args.frames} "
                                   f"({args.frames - 1} no-grad history frame(s) + 1 fwd+bwd), correlation head with "
                                   f"contrastive + lifted-structure losses, AdamW; "
                                   f"batch {B} per GPU; backbone/render CNN excluded",
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            "roofline": roof, "roofline_hbm": roof_hbm, "roofline_lds": roof_lds,
        }
        if (S, args.frames, args.img) == (200, 2, "704x256") and B in BLOCK_HBM_BYTES_PER_SAMPLE:
            # the block as a whole against HBM (north_star / SURVEY 8d): algorithmic bytes per sample x samples/s.  Small by
            # construction: the block is compute-bound (arithmetic intensity ~1.6e5 flop/B)
            gbs = BLOCK_HBM_BYTES_PER_SAMPLE[B] * (total_samples / dt) / world / 1e9
            out["roofline_hbm_block"] = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBS, 6),
                                         "bytes_per_sample": BLOCK_HBM_BYTES_PER_SAMPLE[B],
                                         "note": "whole block per GPU, SURVEY 8d's algorithmic bytes; compute-bound, so small"}
        if world == 1 and args.f32_steps > 0 and args.precision == "bf16":
            # the reference's arithmetic is fp32: the same workload in the kernels' exact-f32 MFMA mode, as a secondary
            # figure (never `value`)
            del net, opt, model

            def timed_mode(mode, n_steps):
                torch.cuda.empty_cache()
                torch.manual_seed(15213 + rank)
                m32 = LiftBlock(S, C, heads, D, V, L, img_w, img_h, mode, dev).to(dev)
                o32 = torch.optim.AdamW([p for p in m32.parameters() if p.requires_grad], lr=1e-4)

                def step32():
                    o32.zero_grad(set_to_none=True)
                    m32(feats[0], feats[1], map_emb, args.frames - 1).backward()
                    o32.step()
                step32()
                torch.cuda.synchronize()
                ops.KERNEL_TIMER.start()
                t1 = time.perf_counter()
                for _ in range(n_steps):
                    step32()
                torch.cuda.synchronize()
                d32 = time.perf_counter() - t1
                kt = {k: v for k, v in ops.KERNEL_TIMER.stop().items() if k.startswith("bevr_attn")}
                res = {"value": round(B * n_steps / d32, 4), "unit": "samples/s", "steps": n_steps,
                       "warmup": 1, "ms_per_step": round(d32 / n_steps * 1e3, 2)}
                if kt:   # the dominant kernel of this mode against this mode's matrix peak
                    dk = max(kt, key=lambda k: kt[k]["ms"])
                    pk = MFMA_PEAK_TFLOPS[mode]
                    ach = kt[dk]["flops"] / (kt[dk]["ms"] * 1e-3) / 1e12
                    tms = sum(v["ms"] for v in kt.values())
                    talg = sum(v["flops"] / N_MATMUL[k] * ALG_MATMUL[k] for k, v in kt.items())
                    res["roofline"] = {"bound": "mfma", "kernel": dk, "achieved": round(ach, 3), "peak": round(pk, 1),
                                       "unit": "TFLOP/s", "frac": round(ach / pk, 5), "avg_ms": round(kt[dk]["ms"] / kt[dk]["n"], 3),
                                       "launches": kt[dk]["n"],
                                       "all_attention_aggregate_frac": round(talg / (tms * 1e-3) / 1e12 / pk, 5)}
                return res
            # `f32_mode` stays the exact-f32 figure (comparable with rounds 1-2); the split mode has its own key
            out["bf16x3_mode"] = dict(timed_mode("bf16x3", args.f32_steps), note="same workload at fp32 TOLERANCE: f32 "
                                      "storage and per-pair arithmetic, matrix products as three split-bf16 MFMAs "
                                      "(BEVR_PREC_BF16X3; results within ~1e-5 of the exact mode, tests hold it to the f32 "
                                      "limits); peak = bf16 dense / 3")
            out["f32_mode"] = dict(timed_mode("f32", 1), note="exact-f32 MFMA operands (v_mfma_f32_32x32x2_f32), the "
                                   "tests' reference mode and the reference's arithmetic; peak = the f32 matrix rate")
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only (bench contract)
            torch.cuda.empty_cache()
            same = gpu_same_workloads(dev)
            out["cpu_baseline"] = cpu_baseline()
            # the GPU path on the SAME two workloads (whole steps incl. correlation head and AdamW; the CPU figures are
            # the encoder layers alone), f32 = the kernels' exact-f32 mode, the CPU baseline's arithmetic
            out["cpu_baseline"]["gpu_same_workload"] = {
                "cfg1": dict(same["cfg1"], cpu_samples_per_s=round(out["cpu_baseline"]["value"], 4)),
                "cfg2_geometry_s56": dict(same["cfg2_geometry_s56"], cpu_samples_per_s=round(
                    out["cpu_baseline"]["cfg2_geometry_s56"]["samples_per_s_at_s56"], 5))}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
