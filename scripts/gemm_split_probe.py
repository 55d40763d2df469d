"""What would "two whole rounds of 128x128 tiles + a K-split tail" buy the Winograd F(4x4) GEMM stage ([36 x 256] x 1024 x 1024)?
Times, with the existing implicit-GEMM kernel run as a plain GEMM (1x1 conv over a P x 256 image, one shared weight matrix):
  full   : P = 36, the tile / ring the generator uses today (64x64 tiles)            -- env from the caller
  rounds : P = 32 -> 512 tiles of 128x128, K = 1024 (two per CU)
  tail   : P = 16, K = 256 -> 256 blocks of 8 stages (= 4 positions split 4-way along K)
Run once per (T2V_CONV_TILE, T2V_CONV_RING) setting: the variables are read once per process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
N = 1024
which = sys.argv[1:] or ["full", "rounds", "tail"]
cfg = {"full": (36, 1024), "rounds": (32, 1024), "tail": (16, 256), "full18": (18, 1024)}
for name in which:
    P, K = cfg[name]
    desc = ops.conv_desc(P, 256, K, N, 1, 1, 0, ops.PAD_ZERO)
    x = torch.randn(P, 256, K, device=dev)
    w = torch.randn(N, K, 1, 1, device=dev) * 0.02
    pw = ops.pack_conv_weight(w, desc, K)
    y = torch.empty(P, 256, N, device=dev)
    run = lambda: ops.conv2d(x, pw, None, desc, y_cs=N, out=y)
    for _ in range(80):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    fl = 2.0 * P * 256 * K * N
    print("%-7s tile=%s ring=%s  M=%5d K=%5d  %.4f ms  %6.1f TF" % (name, os.environ.get("T2V_CONV_TILE", "auto"),
          os.environ.get("T2V_CONV_RING", "auto"), P * 256, K, ms, fl / ms / 1e9), flush=True)
