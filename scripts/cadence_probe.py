import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from text2video_amd import ops
dev = torch.device("cuda:0")
C, hb, wb = 1024, 64, 64
desc = ops.conv_desc(hb, wb, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
xs = [torch.randn(hb, wb, C, device=dev), torch.relu(torch.randn(hb, wb, C, device=dev))]
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
wt = ops.pack_conv_weight(w, desc, C); bias = torch.zeros(C, device=dev)
stats = ops.conv_stats_buffer(desc, dev); y = torch.empty(hb, wb, C, device=dev)
wss = [ops.winograd_workspace(desc, C, dev) for _ in range(2)]
for i in range(2): ops.conv2d_winograd(xs[i], wt, bias, desc, stats=stats, out=y, workspace=wss[i], stages=1)
def launch(i): ops.conv2d_winograd(xs[i & 1], wt, bias, desc, stats=stats, out=y, workspace=wss[i & 1], stages=2)
for iters in (40, 200):
    for rep in range(3):
        for i in range(64): launch(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for i in range(iters): launch(i)
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
        print("iters %d: issue loop %.1f us/launch on the host, cadence %.1f us/launch on the GPU" % (iters, (t1 - t0) / iters * 1e6, e0.elapsed_time(e1) / iters * 1e3), flush=True)
# same with one workspace only
def launch1(i): ops.conv2d_winograd(xs[0], wt, bias, desc, stats=stats, out=y, workspace=wss[0], stages=2)
for i in range(64): launch1(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200): launch1(i)
e1.record(); torch.cuda.synchronize()
print("one workspace: cadence %.1f us" % (e0.elapsed_time(e1) / 200 * 1e3))
