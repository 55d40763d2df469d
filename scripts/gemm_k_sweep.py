"""Efficiency of the implicit-GEMM kernel as a plain GEMM [M x K] x [K x N] vs K (1x1 conv over a 36 x 256 image)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
N = 1024
for (P, T) in [(36, 256), (16, 1024)]:
    for K in [256, 512, 1024, 2048, 4096, 8192]:
        desc = ops.conv_desc(P, T, K, N, 1, 1, 0, ops.PAD_ZERO)
        x = torch.randn(P, T, K, device=dev)
        w = torch.randn(N, K, 1, 1, device=dev) * 0.02
        pw = ops.pack_conv_weight(w, desc, K)
        y = torch.empty(P, T, N, device=dev)
        run = lambda: ops.conv2d(x, pw, None, desc, y_cs=N, out=y)
        for _ in range(60):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        fl = 2.0 * P * T * K * N
        print("M=%5d K=%5d  %.4f ms  %6.1f TF  (%.1f %%)" % (P * T, K, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573), flush=True)
