"""Proxy for a polyphase-Winograd (F(4,2) x F(4,2) on the four sub-pixel phases: 81 positions) form of the stride-2 / transposed
layers: the GEMM stage of an F(4x4,3x3) conv with the same [tiles x Cin] x [Cin x Cout] per position (36 positions), scaled by
81/36, against the direct kernel's time for the layer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=60):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# second proxy: the same total number of tile rows in 36 positions (a 96x96 / 192x192 map: 576 / 2304 tiles = 256 / 1024 x 81/36)
for name, (HT, C, N) in {"down3 512->1024 (64x64 out)": (64, 512, 1024), "up1 1024->512 (64x64 in)": (64, 1024, 512),
                         "down2 256->512 (128x128 out)": (128, 256, 512), "up2 512->256 (128x128 in)": (128, 512, 256),
                         "down3 proxy, 96x96 map (576 tiles x 36)": (96, 512, 1024), "up1 proxy, 96x96 map": (96, 1024, 512),
                         "down2 proxy, 192x192 map (2304 tiles x 36)": (192, 256, 512), "up2 proxy, 192x192 map": (192, 512, 256)}.items():
    desc = ops.conv_desc(HT, HT, C, N, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    x = torch.randn(HT, HT, C, device=dev)
    w = torch.randn(N, C, 3, 3, device=dev) * 0.02
    b = torch.randn(N, device=dev)
    pu = ops.pack_conv_weight(w, desc, C)
    ws = ops.winograd_workspace(desc, C, dev)
    ops.conv2d_winograd(x, pu, b, desc, workspace=ws)
    t_gemm = timed(lambda: ops.conv2d_winograd(x, pu, b, desc, workspace=ws, stages=2))
    t_all = timed(lambda: ops.conv2d_winograd(x, pu, b, desc, workspace=ws))
    gf = 2.0 * 36 * (HT // 4) ** 2 * C * N / 1e9
    print("%-30s F(4,3) GEMM stage %.1f us (%.1f TF, form %s), whole conv %.1f us -> 81 positions: GEMM ~%.1f us + transforms ~%.1f us"
          % (name, t_gemm, gf / t_gemm * 1e3, ops.winograd_gemm_form(desc), t_all, t_gemm * 81 / 36, (t_all - t_gemm) * 81 / 36))
