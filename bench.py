#!/usr/bin/env python
"""bench.py -- frames/sec of the vid2vid pose->RGB generator on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one output frame of the hot path: pack the 3-pose-map window (uint8 maps already
resident in HBM), run CompositeGenerator (SURVEY config 2: ngf 128, 3 down-samplings, 9 ResNet
blocks, 512x512; --openpose_only => no flow branch, `--flow` turns the flow-warp compositor on),
shift the 2-frame FIFO and convert the frame to uint8 (tensor2im) in HBM.  N>1: each rank runs its
own 64-frame-style chunk (sequence-chunk sharding; weak scaling) and the chunk outputs
are all-gathered over RCCL inside the timed region.  Weights are random-init (seeded), data is
synthetic: there is no network for checkpoints.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (1024->1024 3x3 ResnetBlock conv, 77.3 GFLOP per launch) timed with
                  HIP events on the launch stream vs the fp32 MFMA peak (157.3 TFLOP/s)
  "cpu_baseline": the CPU oracle (stock torch fp32) timed on this box's host cores on a bounded
                  sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def gflop_per_frame(H, W, flow, ngf=128, n_down=3, n_blocks=9):
    """Algorithmic FLOPs: sum 2*k*k*Cin*Cout*Hout*Wout, transposed convs on the input grid."""
    f = 0.0
    for cin in (9, 6):
        f += 2 * 49 * cin * ngf * H * W
        for i in range(n_down):
            f += 2 * 9 * (ngf << i) * (ngf << (i + 1)) * (H >> (i + 1)) * (W >> (i + 1))
        f += (n_blocks - n_blocks // 2) * 2 * 2 * 9 * (ngf << n_down) ** 2 * (H >> n_down) * (W >> n_down)
    branch = (n_blocks // 2) * 2 * 2 * 9 * (ngf << n_down) ** 2 * (H >> n_down) * (W >> n_down)
    for i in range(n_down):
        l = n_down - i
        branch += 2 * 9 * (ngf << l) * (ngf << (l - 1)) * (H >> l) * (W >> l)
    f += branch + 2 * 49 * ngf * 3 * H * W
    if flow:
        f += branch + 2 * 49 * ngf * 3 * H * W
    return f / 1e9


def local_gflop_per_frame(H, W, flow, ngf=64, n_blocks=3):
    """CompositeLocalGenerator add-on at full resolution."""
    f = 0.0
    for cin in (9, 6):
        f += 2 * 49 * cin * ngf * H * W + 2 * 9 * ngf * 2 * ngf * (H // 2) * (W // 2)
    branch = n_blocks * 2 * 2 * 9 * (2 * ngf) ** 2 * (H // 2) * (W // 2) + 2 * 9 * 2 * ngf * ngf * (H // 2) * (W // 2)
    f += branch + 2 * 49 * ngf * 3 * H * W
    if flow:
        f += branch + 2 * 49 * ngf * 3 * H * W
    return f / 1e9


def synthetic_pose_u8(n, H, W, seed):
    """pose maps as the dataset would hand them over: uint8 HWC, black background, ~1 % coloured pixels"""
    rng = np.random.default_rng(seed)
    a = np.zeros((n, H, W, 3), np.uint8)
    m = rng.random((n, H, W)) < 0.01
    a[m] = rng.integers(0, 256, size=(int(m.sum()), 3), dtype=np.uint8)
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=62)   # a 64-pose-map sequence yields 62 frames
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--flow", action="store_true", help="enable the flow-warp compositor branch")
    ap.add_argument("--scales", type=int, default=1, choices=[1, 2],
                    help="2 = config 4's two-scale generator: G0 at H/2 x W/2 + local enhancer G1 at H x W")
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--kernel-iters", type=int, default=40)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)"
              % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("T2V_BENCH_FORCE_DIST") == "1":   # the env var exercises the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from text2video_amd import ops
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict

    H, W, K, Wm = args.height, args.width, args.steps, args.warmup
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=not args.flow, norm="batch")
    sd = synthetic_state_dict(spec, seed=1, flow_gain=0.1)
    nets = [HipGenerator(spec, dev).load_state_dict(sd)]
    if args.scales == 2:
        spec1 = GeneratorSpec(ngf=64, n_blocks=3, no_flow=not args.flow, norm="batch", is_local=True, scale=1)
        nets.append(HipGenerator(spec1, dev).load_state_dict(synthetic_state_dict(spec1, seed=2, flow_gain=0.1)))
    model = Vid2VidModelG(nets)

    nposes = K + Wm + 2
    poses = torch.from_numpy(synthetic_pose_u8(nposes, H, W, seed=rank)).to(dev)   # resident in HBM
    window = torch.zeros(H, W, 12, dtype=torch.float32, device=dev)
    frames = torch.empty(K, H, W, 4, dtype=torch.uint8, device=dev)
    gathered = torch.empty(world * K, H, W, 4, dtype=torch.uint8, device=dev) if dist else None

    def step(t, out_slot):
        for f in range(3):                       # sliding window of tG = 3 pose maps, oldest first
            ops.pose_u8_to_f32(poses[t + f], window, 3 * f)
        out = model.inference_nhwc(window)
        u8 = ops.tensor2im_u8(out)
        if out_slot is not None:
            frames[out_slot].copy_(u8)

    for t in range(Wm):
        step(t, None)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        step(Wm + t, t)
    if dist:
        dist.all_gather_into_tensor(gathered, frames)   # RCCL over xGMI: chunk outputs to every rank
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        fps = world * K / elapsed
        if args.scales == 2:   # G0 on the half-resolution pyramid level + the local enhancer
            gf = gflop_per_frame(H // 2, W // 2, args.flow) + local_gflop_per_frame(H, W, args.flow)
        else:
            gf = gflop_per_frame(H, W, args.flow)
        # ---- dominant kernel, timed live with HIP events on the stream it is launched on ----
        C, hb, wb = 1024, H // 8, W // 8
        desc = ops.conv_desc(hb, wb, C, C, 3, 1, 1, ops.PAD_REFLECT)
        # inputs as the kernel sees them in a frame: conv1 of a ResnetBlock reads the residual stream
        # (dense, signed), conv2 reads a ReLU output (half zeros) -- alternate the two
        xs = [torch.randn(hb, wb, C, device=dev), torch.relu(torch.randn(hb, wb, C, device=dev))]
        wt = ops.pack_conv_weight(sd["model_res_img.0.conv_block.1.weight"].to(dev), desc, C)
        bias = sd["model_res_img.0.conv_block.1.bias"].to(dev)
        stats = ops.conv_stats_buffer(desc, dev)
        y = torch.empty(hb, wb, C, device=dev)
        for i in range(64):   # ~35 ms: the clocks ramp back up for tens of ms after the host-side gap above
            ops.conv2d(xs[i & 1], wt, bias, desc, y_cs=C, stats=stats, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.kernel_iters):
            ops.conv2d(xs[i & 1], wt, bias, desc, y_cs=C, stats=stats, out=y)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / args.kernel_iters
        k_flop = 2.0 * 9 * C * C * hb * wb
        achieved = k_flop / (k_ms * 1e-3) / 1e12
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("conv_igemm_rb_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": "conv_igemm_kernel<128x128,fp32 32x32x2,reflect,stats> 1024->1024 3x3 @%dx%d" % (hb, wb),
                    "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "ms_per_launch": round(k_ms, 4), "gflop_per_launch": round(k_flop / 1e9, 2)}
        # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
        cpu = None
        if args.cpu_frames > 0 and world == 1:   # reported at N=1 only (rank 0)
            from oracle.generator_ref import CompositeGenerator, Vid2VidInferenceRef
            cores = torch.get_num_threads()
            ref_net = CompositeGenerator(9, 3, 6, 128, 3, 9, spec.no_flow, "batch")
            ref_net.load_state_dict(sd, strict=False)
            ref_nets = [ref_net]
            if args.scales == 2:
                from oracle.generator_ref import CompositeLocalGenerator
                loc = CompositeLocalGenerator(9, 3, 6, 128, 3, 1, spec.no_flow, "batch")
                loc.load_state_dict(synthetic_state_dict(spec1, seed=2, flow_gain=0.1), strict=False)
                ref_nets.append(loc)
            ref = Vid2VidInferenceRef(ref_nets)
            pf = ((poses[:args.cpu_frames + 3].cpu().float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)
            ref.inference(pf[0:3].unsqueeze(0))          # warm-up frame (thread pool, first-frame path)
            c0 = time.perf_counter()
            for t in range(1, 1 + args.cpu_frames):
                ref.inference(pf[t:t + 3].unsqueeze(0))
            csec = time.perf_counter() - c0
            cpu = {"value": round(args.cpu_frames / csec, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                   "sample": "%d frames %dx%d after 1 warm-up frame, torch %s CPU fp32, %d threads"
                             % (args.cpu_frames, H, W, torch.__version__, cores)}
        result = {
            "metric": "frames/sec 512x512 pose->RGB (vid2vid generator)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: fadg0 openpose_only %dx%d, %d-frame synthetic pose seq per GPU, "
                                   "generator-only inference, ngf128 n_down3 n_blocks9%s, %s"
                                   % ("configs[1]" if (H, W, args.scales) == (512, 512, 1) else "configs[3]-style", H, W, K,
                                      " + local enhancer (n_scales_spatial 2)" if args.scales == 2 else "",
                                      "flow-warp compositor ON" if args.flow else "no flow branch (--openpose_only)"),
                       "frames_per_gpu": K, "parallelism": "sequence-chunk dp%d" % world,
                       "gflop_per_frame": round(gf, 1), "model_tflops": round(fps * gf / 1e3, 2),
                       "model_frac_of_fp32_mfma_peak": round(fps * gf / 1e3 / (PEAK_FP32_MFMA_TFLOPS * world), 4)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:   # RCCL writes banner lines through C stdio: flush them so the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
