"""Fixed-grid Winograd-domain weight gradient (wino_wgrad_sk_kernel) against one block per tile (conv_wgrad_kernel with 36
"taps"): bit equality of dW and time of the reduction stage (reduce + transform back), per layer shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops


def timed(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(H, W, Cin, Cout, batch):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT)
    x = torch.randn(batch, H, W, Cin, device=dev)
    dy = torch.randn(batch, H, W, Cout, device=dev)
    ws = ops.backward_weight_winograd_workspace(desc, Cin, batch, dev)
    ws.fill_(float("nan"))
    ops.conv2d_backward_weight_winograd_stages(x, dy, desc, ws, batch, 0, False)
    res = {}
    for sk in ("0", "1"):
        os.environ["T2V_WGRAD_SK"] = sk
        dw = ops.conv2d_backward_weight_winograd_reduce(desc, ws, batch, Cin, Cout).clone()
        t = timed(lambda: ops.conv2d_backward_weight_winograd_reduce(desc, ws, batch, Cin, Cout))
        res[sk] = (dw, t)
    ref = torch.nn.grad.conv2d_weight(torch.nn.functional.pad(x.permute(0, 3, 1, 2).double(), (1, 1, 1, 1), mode="reflect"),
                                      (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double()) if H * W * Cin * Cout <= 64 * 64 * 256 * 256 else None
    eq = torch.equal(res["0"][0], res["1"][0]) and bool(torch.isfinite(res["1"][0]).all())
    err = "" if ref is None else "  rel err vs fp64 %.2e" % float((res["1"][0].double() - ref).abs().max() / ref.abs().max())
    print("%3dx%3d %4d->%4d x%d frames: bit-equal %s; reduction stage %.1f -> %.1f us%s" %
          (H, W, Cin, Cout, batch, eq, res["0"][1], res["1"][1], err), flush=True)
    return eq


if __name__ == "__main__":
    ok = True
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
    for g in [(64, 64, 1024, 1024, 2), (64, 64, 1024, 1024, 1), (32, 32, 256, 256, 2), (64, 88, 640, 896, 2), (128, 128, 512, 512, 1)][:n]:
        ok &= run(*g)
    sys.exit(0 if ok else 1)
