"""Micro-benchmark of single conv launches at the generator's real layer shapes (for rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import ops

from kernel_bench_shapes import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="rb1024")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--warmup", type=int, default=80)
ap.add_argument("--winograd", type=int, nargs="?", const=1, default=0, help="1 = F(2x2,3x3), 2 = F(4x4,3x3)")
ap.add_argument("--no_stats", action="store_true", help="without the norm-statistics epilogue (what it costs)")
ap.add_argument("--stages", type=int, default=7, help="Winograd stage mask: 1 input transform, 2 GEMM, 4 output transform")
ap.add_argument("--batch", type=int, default=1, help="images per launch (t2v_conv2d_forward_batch; direct algorithm)")
args = ap.parse_args()
dev = torch.device("cuda:0")
for name in args.shapes.split(","):
    H, W, Cin, Cout, k, st, pad, pm, tr, stats = SHAPES[name]
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr, ops.ACT_TANH if Cout == 3 else ops.ACT_NONE,
                         algo=args.winograd)
    xcs = ops.round_up(Cin, 4)
    x = torch.randn(H, W, xcs, device=dev)
    w = torch.randn(*((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)), device=dev) * 0.02
    pw = ops.pack_conv_weight(w, desc, xcs)
    b = torch.randn(Cout, device=dev)
    sb = ops.conv_stats_buffer(desc, dev) if (stats and not args.no_stats) else None
    ho, wo = ops.conv_out_dims(desc)
    ycs = ops.round_up(Cout, 4)
    y = torch.empty(ho, wo, ycs, device=dev)
    flop = 2.0 * k * k * Cin * Cout * (H * W if tr else ho * wo)
    wws = ops.winograd_workspace(desc, xcs, dev) if args.winograd else None
    if args.winograd:
        ops.conv2d_winograd(x, pw, b, desc, stats=sb, out=y, workspace=wws)
    if args.batch > 1:
        B = args.batch
        xb = torch.randn(B, H, W, xcs, device=dev)
        yb = torch.empty(B, ho, wo, ycs, device=dev)
        sbb = torch.empty(B * sb.numel(), device=dev) if sb is not None else None
        flop *= B
    run = (lambda: ops.conv2d_batch(xb, pw, b, desc, y_cs=ycs, stats=sbb, out=yb)) if args.batch > 1 else \
        (lambda: ops.conv2d_winograd(x, pw, b, desc, stats=sb, out=y, workspace=wws, stages=args.stages)) if args.winograd else \
        (lambda: ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y))
    for _ in range(args.warmup):  # clocks ramp over tens of ms: warm up long enough
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print("%-8s %8.4f ms  %7.2f GFLOP  %7.2f TFLOP/s" % (name, ms, flop / 1e9, flop / ms / 1e9), flush=True)
