"""Polyphase Winograd (ALGO_POLYPHASE) against the direct implicit-GEMM kernel and an fp64 torch reference on the generator's
stride-2 / transposed layer shapes: max error, statistics partials through the finalize, time per conv (all stages) and per stage."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, n=40):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("down3", 128, 128, 512, 1024, False), ("down2", 256, 256, 256, 512, False), ("up1", 64, 64, 1024, 512, True),
          ("up2", 128, 128, 512, 256, True), ("down3 512x320", 128, 80, 512, 1024, False), ("up1 512x320", 64, 40, 1024, 512, True),
          ("down3 512x680", 128, 170, 512, 1024, False), ("down2 512x680", 256, 340, 256, 512, False),
          ("up1 512x680", 64, 85, 1024, 512, True), ("up2 512x680", 128, 170, 512, 256, True), ("ragged small", 10, 14, 32, 128, False),
          ("ragged small up", 7, 9, 64, 128, True),
          ("down1", 512, 512, 128, 256, False), ("up3", 256, 256, 256, 128, True),
          ("down3 1024^2", 256, 256, 512, 1024, False), ("up1 1024^2", 128, 128, 1024, 512, True),
          ("ngf64 down2", 128, 128, 128, 256, False), ("ngf64 up1", 64, 64, 512, 256, True),
          ("small down", 16, 24, 32, 128, False), ("small up", 8, 12, 64, 128, True)]
for name, H, W, Cin, Cout, tr in shapes:
    d0 = ops.conv_desc(H, W, Cin, Cout, 3, 2, 1, ops.PAD_ZERO, tr)
    dp = ops.with_algo(d0, ops.ALGO_POLYPHASE)
    assert ops.polyphase_supported(dp, Cin), name
    w = (torch.randn(Cin, Cout, 3, 3, device=dev) if tr else torch.randn(Cout, Cin, 3, 3, device=dev)) * 0.03
    b = torch.randn(Cout, device=dev) * 0.1
    x = torch.relu(torch.randn(H, W, Cin, device=dev))
    xr = x.permute(2, 0, 1).unsqueeze(0).double()
    ref = (torch.nn.functional.conv_transpose2d(xr, w.double(), b.double(), stride=2, padding=1, output_padding=1) if tr else
           torch.nn.functional.conv2d(xr, w.double(), b.double(), stride=2, padding=1))[0].permute(1, 2, 0)
    p0, pp = ops.pack_conv_weight(w, d0, Cin), ops.pack_conv_weight(w, dp, Cin)
    s0, sp = ops.conv_stats_buffer(d0, dev), ops.conv_stats_buffer(dp, dev)
    ws = ops.winograd_workspace(dp, Cin, dev)
    ws.fill_(float("nan"))
    y0 = ops.conv2d(x, p0, b, d0, y_cs=Cout, stats=s0)
    yp = ops.conv2d_winograd(x, pp, b, dp, stats=sp, workspace=ws)
    torch.cuda.synchronize()
    e0, ep = (y0.double() - ref).abs().max().item(), (yp.double() - ref).abs().max().item()
    m0, mp = ops.instance_norm_finalize(s0, d0), ops.instance_norm_finalize(sp, dp)
    mref = torch.stack([ref.mean((0, 1)), 1.0 / torch.sqrt(ref.var((0, 1), unbiased=False) + 1e-5)], 1).float()
    em0, emp = (m0.view(-1, 2) - mref).abs().max().item(), (mp.view(-1, 2) - mref).abs().max().item()
    t0 = timed(lambda: ops.conv2d(x, p0, b, d0, y_cs=Cout, stats=s0))
    tp = timed(lambda: ops.conv2d_winograd(x, pp, b, dp, stats=sp, workspace=ws))
    st = [timed(lambda k=k: ops.conv2d_winograd(x, pp, b, dp, stats=sp, workspace=ws, stages=k)) for k in (1, 2, 4)]
    print("%-14s %4dx%-4d %4d->%-4d  direct %.1f us (err %.1e, stats %.1e) | polyphase %.1f us = in %.1f + gemm %.1f + out %.1f (err %.1e, stats %.1e)  %+.0f%%"
          % (name, H, W, Cin, Cout, t0, e0, em0, tp, st[0], st[1], st[2], ep, emp, 100 * (tp / t0 - 1)), flush=True)
