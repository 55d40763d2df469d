"""ms per frame of the full-size generator (512x512, ngf 128) when N independent sequences advance in lock-step
(t2v_generator_forward_batch).  Usage: python scripts/batch_probe.py [--flow 0|1] [--batches 1,2,4] [--steps 40]
Environment knobs of interest: T2V_NORM_TICKET, T2V_WINO_GEMM_TILE, T2V_STREAMS."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_pose_u8  # noqa: E402
from text2video_amd import ops  # noqa: E402
from text2video_amd.generator import GeneratorSpec, HipGenerator, Recurrence, Vid2VidModelG, synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--flow", type=int, default=1)
ap.add_argument("--batches", default="1,2,4")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
H = W = a.size
spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=not a.flow, norm="batch")
model = Vid2VidModelG([HipGenerator(spec, dev).load_state_dict(synthetic_state_dict(spec, seed=1, flow_gain=0.1))])
K, Wm = a.steps, a.warmup
for nb in [int(v) for v in a.batches.split(",")]:
    poses = [torch.from_numpy(synthetic_pose_u8(K + Wm + 2, H, W, seed=i)).to(dev) for i in range(nb)]
    windows = [torch.zeros(H, W, 12, device=dev) for _ in range(nb)]
    states = [Recurrence() for _ in range(nb)]

    def step(t):
        for i in range(nb):
            for f in range(3):
                ops.pose_u8_to_f32(poses[i][t + f], windows[i], 3 * f)
        outs = model.inference_nhwc_batch(windows, states)
        return [ops.tensor2im_u8(o) for o in outs]

    for t in range(Wm):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        step(Wm + t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("batch %d: %.3f ms per step, %.3f ms per frame, %.2f frames/s  (flow=%d, env %s)"
          % (nb, 1e3 * el / K, 1e3 * el / K / nb, nb * K / el, a.flow,
             {k: v for k, v in os.environ.items() if k.startswith("T2V_")}), flush=True)
