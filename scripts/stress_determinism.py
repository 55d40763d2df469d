"""Race screen: every conv shape launched repeatedly must reproduce its first output bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import ops
from kernel_bench_shapes import SHAPES
dev = torch.device("cuda:0")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for name in names:
    H, W, Cin, Cout, k, st, pad, pm, tr, stats = SHAPES[name]
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr, ops.ACT_TANH if Cout == 3 else ops.ACT_NONE)
    xcs = ops.round_up(Cin, 4)
    torch.manual_seed(0)
    x = torch.randn(H, W, xcs, device=dev)
    w = torch.randn(*((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)), device=dev) * 0.02
    pw = ops.pack_conv_weight(w, desc, xcs)
    b = torch.randn(Cout, device=dev)
    ho, wo = ops.conv_out_dims(desc)
    ycs = Cout if Cout % 4 == 0 else 4
    sb = ops.conv_stats_buffer(desc, dev) if stats else None
    ref = ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb).clone()
    ref_s = sb.clone() if stats else None
    bad = 0
    for i in range(iters):
        y = torch.full((ho, wo, ycs), float("nan"), device=dev)
        if stats: sb.fill_(float("nan"))
        ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y)
        if not torch.equal(y, ref) or (stats and not torch.equal(sb, ref_s)):
            bad += 1
            if bad == 1:
                d = (y - ref).abs()
                idx = torch.nonzero(d > 0)
                print("   first mismatch: n=%d max=%g at %s nan=%d" % (idx.shape[0], d.max().item(), idx[:3].tolist(), torch.isnan(y).sum().item()))
    print("%-12s finite=%s mismatching launches %d / %d" % (name, bool(torch.isfinite(ref).all()), bad, iters), flush=True)
