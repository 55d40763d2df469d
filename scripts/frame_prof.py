"""N frames of the config-2 generator (flow on unless --noflow) for rocprofv3: nothing but the frame loop.
Usage: [T2V_STREAMS=1] frame_prof.py [--frames 40] [--noflow] [--size 512] [--width W] [--batch N] [--scales 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from text2video_amd import ops
from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--noflow", action="store_true")
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--width", type=int, default=0, help="frame width when it differs from --size (the reference's 512x320 / 512x680)")
ap.add_argument("--batch", type=int, default=1, help="independent sequences advanced in lock-step")
ap.add_argument("--scales", type=int, default=1, help="2: the coarse generator at half size + the local enhancer (configs[3] two-scale)")
a = ap.parse_args()
dev = torch.device("cuda:0")
spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=a.noflow, norm="batch")
nets = [HipGenerator(spec, dev).load_state_dict(synthetic_state_dict(spec, 1, flow_gain=0.1))]
if a.scales == 2:
    spec1 = GeneratorSpec(ngf=64, n_blocks=3, no_flow=a.noflow, norm="batch", is_local=True, scale=1)
    nets.append(HipGenerator(spec1, dev).load_state_dict(synthetic_state_dict(spec1, 2, flow_gain=0.1)))
model = Vid2VidModelG(nets)
H = a.size
W = a.width or a.size
rng = np.random.default_rng(0)
win = torch.zeros(H, W, 12, device=dev)
win[..., :9] = torch.from_numpy(np.where(rng.random((H, W, 1)) < 0.02, rng.uniform(-1, 1, (H, W, 9)), -1.0).astype(np.float32)).to(dev)
from text2video_amd.generator import Recurrence
wins = [win] + [win.clone() for _ in range(a.batch - 1)]
states = [Recurrence() for _ in range(a.batch)]
for _ in range(10):
    model.inference_nhwc_batch(wins, states)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.frames):
    model.inference_nhwc_batch(wins, states)
torch.cuda.synchronize()
# FRAMES counts generated frames (steps x batch): the per-frame table divides by it
print("FRAMES %d  %.3f ms/frame (batch %d)" % ((a.frames + 10) * a.batch, 1e3 * (time.perf_counter() - t0) / a.frames / a.batch, a.batch))
