"""Does the direct weight-gradient kernel's time depend on WHERE its operand buffers lie?  (bench.py's `up 1024->512` row moved
0.664 -> 0.778 ms between rounds with no kernel change.)  Two frames from buffers of their own (conv2d_backward_weight_pair), the
second frame's buffers `gap` bytes further away each time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import ops

dev = "cuda:0"


def timed(fn, iters=30):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, (h, w, ci, co, tr) in (("down 512->1024", (128, 128, 512, 1024, False)), ("up 1024->512", (64, 64, 1024, 512, True))):
    d = ops.conv_desc(h, w, ci, co, 3, 2, 1, ops.PAD_ZERO, tr)
    ho, wo = ops.conv_out_dims(d)
    nx, ny = h * w * ci, ho * wo * co
    gf = 2 * 2.0 * 9 * ci * co * (h * w if tr else ho * wo) / 1e9
    for gap_kb in (0, 4, 64, 256, 1024, 2048, 4096 + 64, 8192, 16384 + 512, 65536):
        pool = torch.randn(2 * (nx + ny) + 4 * (gap_kb * 256 + 1024), device=dev)
        g = gap_kb * 256
        o = 0
        x0 = pool[o:o + nx].view(h, w, ci); o += nx + g
        x1 = pool[o:o + nx].view(h, w, ci); o += nx + g
        y0 = pool[o:o + ny].view(ho, wo, co); o += ny + g
        y1 = pool[o:o + ny].view(ho, wo, co)
        ms = timed(lambda: ops.conv2d_backward_weight_pair(x0, y0, x1, y1, d))
        print("%-15s gap %6d KiB: %.4f ms  %.1f TF" % (name, gap_kb, ms, gf / ms), flush=True)
    # separate allocations, as bench.py makes them
    for rep in range(3):
        junk = [torch.empty(1 + 37 * 1024 * rep, device=dev) for _ in range(rep)]
        xs = [torch.randn(h, w, ci, device=dev) for _ in range(2)]
        dys = [torch.randn(ho, wo, co, device=dev) for _ in range(2)]
        ms = timed(lambda: ops.conv2d_backward_weight_pair(xs[0], dys[0], xs[1], dys[1], d))
        print("%-15s separate allocations #%d (x %x %x dy %x %x): %.4f ms  %.1f TF"
              % (name, rep, xs[0].data_ptr(), xs[1].data_ptr(), dys[0].data_ptr(), dys[1].data_ptr(), ms, gf / ms), flush=True)
