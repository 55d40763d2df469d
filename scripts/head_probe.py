"""Why is the 7x7 head slower inside a frame than alone?  Time it (HIP events around single launches) after
different predecessors: itself, a norm-apply producing its input, a big MFMA conv."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
H = W = 512; C = 128
desc = ops.conv_desc(H, W, C, 3, 7, 1, 3, ops.PAD_REFLECT, act=ops.ACT_TANH)
w = torch.randn(3, C, 7, 7, device=dev) * 0.02
pw = ops.pack_conv_weight(w, desc, C); b = torch.zeros(3, device=dev)
y = torch.empty(H, W, 4, device=dev)
raw = torch.randn(H, W, C, device=dev)
mr = torch.stack([torch.zeros(C, device=dev), torch.ones(C, device=dev)], 1).contiguous().view(-1)
x = torch.empty_like(raw)
rb = ops.conv_desc(64, 64, 1024, 1024, 3, 1, 1, ops.PAD_REFLECT)
rx = torch.randn(64, 64, 1024, device=dev); rw = ops.pack_conv_weight(torch.randn(1024, 1024, 3, 3, device=dev) * 0.01, rb, 1024)
ry = torch.empty(64, 64, 1024, device=dev)
def head(inp): ops.conv2d(inp, pw, b, desc, y_cs=4, out=y)
def apply_(relu): ops.instance_norm_apply(raw, mr, relu=relu, out=x)
def big(): ops.conv2d(rx, rw, None, rb, y_cs=1024, out=ry)
def timed(pre, inp, n=20):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); head(inp); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for _ in range(30): big()
apply_(True)
print("head after head        (dense randn input)  %.1f us" % timed(lambda: head(raw), raw))
print("head after head        (relu'd input)       %.1f us" % timed(lambda: head(x), x))
print("head after apply(relu) producing its input  %.1f us" % timed(lambda: apply_(True), x))
print("head after apply(none) producing its input  %.1f us" % timed(lambda: apply_(False), x))
print("head after big MFMA conv (input cold)       %.1f us" % timed(big, x))
def both(): big(); apply_(True)
print("head after conv + apply                      %.1f us" % timed(both, x))
xz = torch.zeros_like(raw)
print("head on zeros                                 %.1f us" % timed(lambda: head(xz), xz))
