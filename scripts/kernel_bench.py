"""Micro-benchmark of single conv launches at the generator's real layer shapes (for rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import ops

SHAPES = {
    # name: (H, W, Cin, Cout, k, stride, pad, pad_mode, transposed, stats)
    "rb1024": (64, 64, 1024, 1024, 3, 1, 1, 1, False, True),
    "down512": (128, 128, 512, 1024, 3, 2, 1, 0, False, True),
    "down256": (256, 256, 256, 512, 3, 2, 1, 0, False, True),
    "down128": (512, 512, 128, 256, 3, 2, 1, 0, False, True),
    "up1024": (64, 64, 1024, 512, 3, 2, 1, 0, True, True),
    "up512": (128, 128, 512, 256, 3, 2, 1, 0, True, True),
    "up256": (256, 256, 256, 128, 3, 2, 1, 0, True, True),
    "stem9": (512, 512, 9, 128, 7, 1, 3, 1, False, True),
    "head3": (512, 512, 128, 3, 7, 1, 3, 1, False, False),
}

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="rb1024")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--warmup", type=int, default=80)
args = ap.parse_args()
dev = torch.device("cuda:0")
for name in args.shapes.split(","):
    H, W, Cin, Cout, k, st, pad, pm, tr, stats = SHAPES[name]
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr, ops.ACT_TANH if Cout == 3 else ops.ACT_NONE)
    xcs = ops.round_up(Cin, 4)
    x = torch.randn(H, W, xcs, device=dev)
    w = torch.randn(*((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)), device=dev) * 0.02
    pw = ops.pack_conv_weight(w, desc, xcs)
    b = torch.randn(Cout, device=dev)
    sb = ops.conv_stats_buffer(desc, dev) if stats else None
    ho, wo = ops.conv_out_dims(desc)
    ycs = Cout if Cout % 4 == 0 else 4
    y = torch.empty(ho, wo, ycs, device=dev)
    flop = 2.0 * k * k * Cin * Cout * (H * W if tr else ho * wo)
    for _ in range(args.warmup):  # clocks ramp over tens of ms: warm up long enough
        ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print("%-8s %8.4f ms  %7.2f GFLOP  %7.2f TFLOP/s" % (name, ms, flop / 1e9, flop / ms / 1e9), flush=True)
