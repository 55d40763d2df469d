"""Race screen of the fixed-grid kernels' accumulator hand-over under load: two streams run Winograd convs (fixed-grid GEMM
stage) and Winograd-domain weight-gradient reductions (fixed grid) + the data gradient on the forward packing of the weights
(the [K][N] form of the fixed-grid GEMM) on their own workspaces at the same time, with changing
inputs, while a third stream streams memory; every result must equal the one-block-per-tile result bit for bit.
usage: stress_fixed_grid.py [iterations=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
H = W = 64
C = 1024
desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
ddesc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT)
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
b = torch.randn(C, device=dev)
pu = ops.pack_conv_weight(w, desc, C)
xs = [torch.randn(H, W, C, device=dev) for _ in range(4)]
dys = [torch.randn(2, H, W, C, device=dev) for _ in range(2)]
x2 = [torch.randn(2, H, W, C, device=dev) for _ in range(2)]
os.environ["T2V_WINO_GEMM_SK"] = "0"
os.environ["T2V_WGRAD_SK"] = "0"
ops.reload_env()
want_y = [ops.conv2d_winograd(x, pu, b, desc).clone() for x in xs]
want_dw, want_dx = [], []
ut = ops.pack_conv_weight_transposed(w, desc, C)
for i in range(2):
    ws = ops.backward_weight_winograd_workspace(ddesc, C, 2, dev)
    want_dw.append(ops.conv2d_backward_weight_winograd_stages(x2[i], dys[i], ddesc, ws, 2, 0, True).clone())
    want_dx.append(ops.conv2d_backward_data_winograd(desc, 2, 1, ws, C, ut).clone())      # one block per tile, transposed copy
os.environ["T2V_WINO_GEMM_SK"] = "1"
os.environ["T2V_WGRAD_SK"] = "1"
ops.reload_env()
assert ops.backward_data_winograd_takes_forward_weights(desc, C, C)
torch.cuda.synchronize()
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
wsA = [ops.winograd_workspace(desc, C, dev) for _ in range(2)]
wsB = ops.backward_weight_winograd_workspace(ddesc, C, 2, dev)
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)     # 1 GiB streamed by the third stream
bad = 0
for it in range(iters):
    with torch.cuda.stream(sC):
        big.add_(1.0)
    with torch.cuda.stream(sA):
        ya = ops.conv2d_winograd(xs[it % 4], pu, b, desc, workspace=wsA[it & 1])
        yb = ops.conv2d_winograd(xs[(it + 1) % 4], pu, b, desc, workspace=wsA[it & 1])     # same workspace, back to back
    with torch.cuda.stream(sB):
        dw = ops.conv2d_backward_weight_winograd_stages(x2[it & 1], dys[it & 1], ddesc, wsB, 2, 0, True)
        dx = ops.conv2d_backward_data_winograd(desc, 2, 1, wsB, C, pu, forward_weights=True)
    torch.cuda.synchronize()
    ok = torch.equal(ya, want_y[it % 4]) and torch.equal(yb, want_y[(it + 1) % 4]) and torch.equal(dw, want_dw[it & 1]) \
        and torch.equal(dx, want_dx[it & 1])
    bad += not ok
print("fixed-grid kernels under load: %d of %d iterations differ" % (bad, iters))
sys.exit(1 if bad else 0)
