"""Whole-frame graph replay against the eager frame loop.

One steady-state step of the frame loop -- 3 x pose_u8_to_f32 -> t2v_generator_forward -> FIFO shift -> tensor2im -- is captured
into a hipGraph (torch.cuda.CUDAGraph: the library launches on torch's current stream, which is the capturing one; its side stream
joins the capture through the fork / join events).  The FIFO ping-pongs between two buffers, so the captured unit is a PAIR of
frames (the state is back in the same buffers after two); the three uint8 pose maps of each frame are copied into static staging
buffers before a replay.  Eager and replay runs alternate on the same box; frames are compared bit for bit.

    python scripts/graph_replay_probe.py [--frames 50] [--rounds 4] [--height 512 --width 512] [--noflow]
T2V_STREAMS=1 in the environment gives the single-stream form of the generator."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from text2video_amd import ops
from text2video_amd.generator import Recurrence

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=512)
ap.add_argument("--noflow", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
H, W, K = args.height, args.width, args.frames - args.frames % 2
model, _ = bench.build_models(dev, not args.noflow, 1)
poses = torch.from_numpy(bench.synthetic_pose_u8(K + 8, H, W, 0)).to(dev)
window = torch.zeros(H, W, 12, dtype=torch.float32, device=dev)
stage = [torch.zeros(H, W, 3, dtype=torch.uint8, device=dev) for _ in range(6)]     # 2 frames x 3 maps
st = [Recurrence()]
out_u8 = [torch.empty(H, W, 4, dtype=torch.uint8, device=dev) for _ in range(2)]


def frame(maps, slot):
    for f in range(3):
        ops.pose_u8_to_f32(maps[f], window, 3 * f)
    out_u8[slot].copy_(ops.tensor2im_u8(model.inference_nhwc_batch([window], st)[0]))


def warm():
    """frames 0..3 eagerly: the first frame's raw-only path, packs, kernel attributes; leaves the FIFO in a known phase"""
    st[0].reset()
    for t in range(4):
        frame([poses[t + f] for f in range(3)], t & 1)
    torch.cuda.synchronize()


def eager(collect=None):
    warm()
    t0 = time.perf_counter()
    for t in range(4, 4 + K):
        frame([poses[t + f] for f in range(3)], t & 1)
        if collect is not None:
            collect.append(out_u8[t & 1].clone())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


warm()
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for t in (4, 5):       # a dry pair on the capture stream (allocator warm-up)
        frame(stage[3 * (t & 1):3 * (t & 1) + 3], t & 1)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
warm()
fixed_prev, fixed_spare = st[0].prev[0], st[0]._spare[0]      # the FIFO buffers the graph is captured on
prev_ptrs = [p.data_ptr() for p in st[0].prev]
with torch.cuda.graph(graph):
    frame(stage[0:3], 0)
    frame(stage[3:6], 1)
assert [p.data_ptr() for p in st[0].prev] == prev_ptrs, "FIFO not back in the same buffers after a frame pair"
torch.cuda.synchronize()


def replay(collect=None):
    warm()
    fixed_prev.copy_(st[0].prev[0])          # the state after the eager warm-up frames, in the captured buffers
    st[0].prev, st[0]._spare = [fixed_prev], [fixed_spare]
    t0 = time.perf_counter()
    for t in range(4, 4 + K, 2):
        for k in range(2):
            for f in range(3):
                stage[3 * k + f].copy_(poses[t + k + f])
        graph.replay()
        if collect is not None:
            collect.append(out_u8[0].clone())
            collect.append(out_u8[1].clone())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K


a, b = [], []
eager(a)
replay(b)
same = all(torch.equal(x, y) for x, y in zip(a, b))
print("frames of the replayed graph %s the eager loop's (%d frames)" % ("EQUAL" if same else "DIFFER FROM", len(a)))
for r in range(args.rounds):
    e = eager()
    g = replay()
    print("round %d: eager %.3f ms/frame, graph replay %.3f ms/frame (%+.2f %%)" % (r, 1e3 * e, 1e3 * g, 100 * (e / g - 1)))
