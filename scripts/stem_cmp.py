import os, sys, torch
sys.path.insert(0, os.getcwd())
from text2video_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (Cin, cs) in ((9, 12), (6, 8)):
    H = W = 512; Cout = 128
    desc = ops.conv_desc(H, W, Cin, Cout, 7, 1, 3, ops.PAD_REFLECT)
    x = torch.zeros(H, W, cs, device=dev); x[..., :Cin] = torch.randn(H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 7, 7, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    pw = ops.pack_conv_weight(w, desc, cs)
    outs = {}
    for m in ("1", "2", "0"):
        os.environ["T2V_CONV_STEM"] = m; ops.reload_env()
        st = ops.conv_stats_buffer(desc, dev)
        y = torch.empty(H, W, Cout, device=dev)
        ops.conv2d(x, pw, b, desc, y_cs=Cout, stats=st, out=y)
        torch.cuda.synchronize()
        outs[m] = (y.clone(), st.clone())
    ref = F.conv2d(F.pad(x[..., :Cin].permute(2, 0, 1).unsqueeze(0).double(), (3, 3, 3, 3), mode="reflect"), w.double(), b.double())[0].permute(1, 2, 0)
    for m in ("1", "2", "0"):
        d = (outs[m][0].double() - ref).abs()
        print("Cin %d mode %s: max|y - fp64| %.3e  (max|y| %.2f)  rms %.3e   stats equal to mode 2: %s  max stats diff %.3e" % (Cin, m, d.max().item(), ref.abs().max().item(), d.pow(2).mean().sqrt().item(),
              torch.equal(outs[m][1], outs["2"][1]), (outs[m][1] - outs["2"][1]).abs().max().item() if outs[m][1].shape == outs["2"][1].shape else -1))
    # garbage in the padding channels must not matter
    os.environ["T2V_CONV_STEM"] = "1"; ops.reload_env()
    x2 = x.clone(); x2[..., Cin:] = 1e3 * torch.randn(H, W, cs - Cin, device=dev)
    y2 = torch.empty(H, W, Cout, device=dev); st = ops.conv_stats_buffer(desc, dev)
    ops.conv2d(x2, pw, b, desc, y_cs=Cout, stats=st, out=y2)
    print("   padding channels filled with noise: equal %s" % torch.equal(y2, outs["1"][0]))
