"""Emulates a merged "two rounds of 128x128 tiles + K-split tail" launch of the F(4x4) GEMM stage by issuing the two parts
on two streams at once (the dispatcher interleaves their blocks); compares with the parts back to back and with today's
64x64-tile launch.  Run with T2V_CONV_TILE=0 T2V_CONV_RING=2 (128x128 tiles) and again without (64x64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import ops
dev = torch.device("cuda:0")
N = 1024


def mk(P, K):
    desc = ops.conv_desc(P, 256, K, N, 1, 1, 0, ops.PAD_ZERO)
    x = torch.randn(P, 256, K, device=dev)
    pw = ops.pack_conv_weight(torch.randn(N, K, 1, 1, device=dev) * 0.02, desc, K)
    y = torch.empty(P, 256, N, device=dev)
    return lambda: ops.conv2d(x, pw, None, desc, y_cs=N, out=y)


full, rounds, tail = mk(36, 1024), mk(32, 1024), mk(16, 256)
s2 = torch.cuda.Stream()


def both(tail_first):
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        (tail if not tail_first else rounds)()
        ev2 = torch.cuda.Event()
        ev2.record()
    (rounds if not tail_first else tail)()
    torch.cuda.current_stream().wait_event(ev2)


def timeit(fn, n=40):
    for _ in range(60):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tag = "tile=%s ring=%s" % (os.environ.get("T2V_CONV_TILE", "auto"), os.environ.get("T2V_CONV_RING", "auto"))
print(tag, "full 36 positions          %.4f ms" % timeit(full))
print(tag, "rounds then tail (serial)  %.4f ms" % timeit(lambda: (rounds(), tail())))
print(tag, "rounds || tail (2 streams) %.4f ms" % timeit(lambda: both(False)))
print(tag, "tail || rounds (2 streams) %.4f ms" % timeit(lambda: both(True)))
