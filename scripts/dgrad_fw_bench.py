"""A/B of the ResnetBlock conv's data gradient (1024 -> 1024 at 64x64): the fixed-grid GEMM reading the transposed copy of the
transformed weights ([N][K]) against the forward packing in place ([K][N]).  Usage: python scripts/dgrad_fw_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from text2video_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H = W = 64
C = 1024
desc = ops.with_algo(ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT), ops.ALGO_WINOGRAD_F4)
g = torch.Generator().manual_seed(1)
xs = torch.randn(1, H, W, C, generator=g).cuda()
dys = torch.randn(1, H, W, C, generator=g).cuda()
wd = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
ws = ops.backward_weight_winograd_workspace(desc, C, 1, "cuda:0")
ops.conv2d_backward_weight_winograd_stages(xs, dys, desc, ws, 1, 0, False)
ut = ops.pack_conv_weight_transposed(wd, desc, C)
u = ops.pack_conv_weight(wd, desc, C)
out = torch.empty(H, W, C, device="cuda")
for name, wt, fw in (("transposed copy [N][K]", ut, False), ("forward packing [K][N]", u, True)) * 2:
    for _ in range(5):
        ops.conv2d_backward_data_winograd(desc, 1, 0, ws, C, wt, out=out, forward_weights=fw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_backward_data_winograd(desc, 1, 0, ws, C, wt, out=out, forward_weights=fw)
    e1.record()
    torch.cuda.synchronize()
    print("%-26s %.1f us per data gradient (GEMM + output transform + fold)" % (name, e0.elapsed_time(e1) / iters * 1e3))
