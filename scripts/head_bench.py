"""7x7 head kernel (csrc/conv_head.hip): time per launch (HIP events, median) and max|delta| vs an fp64 torch reference, at the
generator's sizes.  (Round 5 used it as the same-box A/B of the strip form against the thread-per-pixel form it replaced:
profiles/r05_head_ab.txt.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (H, W, C, act) in ((512, 512, 128, ops.ACT_TANH), (512, 512, 128, ops.ACT_FLOW_W), (512, 320, 128, ops.ACT_TANH), (1024, 1024, 64, ops.ACT_TANH), (67, 45, 32, ops.ACT_TANH)):
    desc = ops.conv_desc(H, W, C, 3, 7, 1, 3, ops.PAD_REFLECT, act=act, act_scale=20.0 if act == ops.ACT_FLOW_W else 0.2)
    w = torch.randn(3, C, 7, 7, device=dev) * 0.02
    b = torch.randn(3, device=dev) * 0.1
    pw = ops.pack_conv_weight(w, desc, C)
    x = torch.relu(torch.randn(H, W, C, device=dev))
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x.permute(2, 0, 1).unsqueeze(0).double(), (3, 3, 3, 3), mode="reflect"), w.double(), b.double())[0]
    if act == ops.ACT_TANH:
        ref = torch.tanh(ref)
    else:
        ref = torch.stack([ref[0] * 20.0, ref[1] * 20.0, torch.sigmoid(ref[2])])
    y = torch.empty(H, W, 4, device=dev)
    for _ in range(5):
        ops.conv2d(x, pw, b, desc, y_cs=4, out=y)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d(x, pw, b, desc, y_cs=4, out=y); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    err = (y[..., :3].permute(2, 0, 1).double() - ref).abs().max().item()
    flop = 2.0 * H * W * C * 49 * 3
    print("%4dx%-4d C=%3d act=%d: %.1f us = %.1f TFLOP/s (fp32 VALU peak 157.3), max|delta| vs fp64 %.2e, pad channel zero: %s"
          % (H, W, C, act, ts[len(ts) // 2], flop / ts[len(ts) // 2] / 1e6, err, bool((y[..., 3] == 0).all())))
