"""Fixed-grid ("stream-K") Winograd GEMM against the tile-per-block kernel on ResnetBlock convs, F(4x4,3x3):
bit equality of the conv output and time of the GEMM stage alone (stages=2) and of the whole conv."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops

def timed(fn, iters=40, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def run(H, W, C):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    x = torch.randn(H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    b = torch.randn(C, device=dev)
    pu = ops.pack_conv_weight(w, desc, C)
    ws = ops.winograd_workspace(desc, C, dev)
    res = {}
    for sk in ("0", "1"):
        os.environ["T2V_WINO_GEMM_SK"] = sk
        ws.fill_(float("nan"))
        y = ops.conv2d_winograd(x, pu, b, desc, workspace=ws)
        torch.cuda.synchronize()
        t_gemm = timed(lambda: ops.conv2d_winograd(x, pu, b, desc, workspace=ws, stages=2))
        t_all = timed(lambda: ops.conv2d_winograd(x, pu, b, desc, workspace=ws))
        res[sk] = (y.clone(), t_gemm, t_all)
    eq = torch.equal(res["0"][0], res["1"][0]) and bool(torch.isfinite(res["1"][0]).all())
    d = (res["0"][0] - res["1"][0]).abs()
    print("   nan %d  differing %d of %d  max|d| %.3e  max|y| %.3e" % (int(torch.isnan(res["1"][0]).sum()), int((d > 0).sum()),
          d.numel(), float(torch.nan_to_num(d).max()), float(res["0"][0].abs().max())))
    print("%3dx%3dx%4d: bit-equal %s; GEMM stage %.1f -> %.1f us, conv %.1f -> %.1f us" %
          (H, W, C, eq, res["0"][1], res["1"][1], res["0"][2], res["1"][2]), flush=True)
    return eq

if __name__ == "__main__":
    ok = True
    shapes = [(64, 64, 1024), (64, 128, 1024), (32, 32, 1024), (64, 64, 512), (128, 128, 256), (64, 40, 1024), (64, 88, 1024),
              (128, 128, 1024)]
    if len(sys.argv) > 1:
        shapes = shapes[:int(sys.argv[1])]
    for (H, W, C) in shapes:
        ok &= run(H, W, C)
    sys.exit(0 if ok else 1)
