"""Weight-gradient kernel micro-benchmark on the generator's layer shapes (kernel_bench_shapes.SHAPES)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
from kernel_bench_shapes import SHAPES
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="rb1024,down512,down256,down128,up1024,up512,up256")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--warmup", type=int, default=40)
ap.add_argument("--batch", type=int, default=1, help="images reduced over in one launch (a train step reduces over its 2 frames)")
args = ap.parse_args()
dev = torch.device("cuda:0")
for name in args.shapes.split(","):
    H, W, Cin, Cout, k, st, pad, pm, tr, stats = SHAPES[name]
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr)
    xcs = ops.round_up(Cin, 4)
    ho, wo = ops.conv_out_dims(desc)
    x = torch.randn(args.batch, H, W, xcs, device=dev)
    dy = torch.randn(args.batch, ho, wo, ops.round_up(Cout, 4), device=dev)
    run = lambda: ops.conv2d_backward_weight(x, dy, desc)
    for _ in range(args.warmup): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    flop = 2.0 * args.batch * k * k * Cin * Cout * (H * W if tr else ho * wo)
    print("%-8s wgrad %8.4f ms  %7.2f GFLOP  %7.2f TFLOP/s" % (name, ms, flop / 1e9, flop / ms / 1e9), flush=True)
