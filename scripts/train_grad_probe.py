"""Gradient agreement of the train step's convolution algorithm choices at full size (512x512, ngf 128, 2 frames):
direct implicit GEMM vs Winograd F(2x2) / F(4x4) in forward + data gradient, with the weight gradient in the Winograd
domain or direct.  Prints per-configuration relative differences against the all-direct run."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2video_amd import train as T
from text2video_amd.options import TrainOptions

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
extra = sys.argv[2:]
opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--no_vgg",
                            "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img", "--fineSize", str(size)] + extra)
H = W = size
rng = np.random.default_rng(0)
pose = torch.zeros(2, H, W, 12, device="cuda:0")
pose[..., :9] = torch.from_numpy(np.where(rng.random((2, H, W, 1)) < 0.02, rng.uniform(-1, 1, (2, H, W, 9)), -1.0).astype(np.float32)).cuda()
real = torch.zeros(2, H, W, 4, device="cuda:0")
real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
prev0 = torch.zeros(1, H, W, 8, device="cuda:0")
prev0[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).cuda()


def run(env):
    for k in ("T2V_CONV_ALGO", "T2V_WGRAD_WINOGRAD", "T2V_WGRAD_BATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    tr = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
    losses, _ = tr.train_step(pose, real, None, prev0.clone(), real_prev=real_prev)
    g = {k: p.grad.clone() for k, p in tr.G.named_upstream_parameters().items() if p.grad is not None}
    del tr
    torch.cuda.empty_cache()
    return losses, g


base_l, base = run({"T2V_CONV_ALGO": "1"})
print("direct losses", {k: round(v, 5) for k, v in base_l.items()})
for name, env in [("F2", {"T2V_CONV_ALGO": "2"}), ("F4 all", {}), ("F4 fwd/dgrad, direct wgrad", {"T2V_WGRAD_WINOGRAD": "0"}),
                  ("F4, wgrad per frame", {"T2V_WGRAD_BATCH": "0"})]:
    l, g = run(env)
    errs = {}
    for k, r in base.items():
        sc = r.abs().max().item()
        if sc > 1e-7:
            errs[k] = (g[k] - r).abs().max().item() / sc
    v = np.array(list(errs.values()))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print("%-30s median %.1e  p90 %.1e  max %.1e  loss diff %.1e | %s" % (
        name, np.median(v), np.quantile(v, 0.9), v.max(), max(abs(l[k] - base_l[k]) for k in l), ["%s %.0e" % kv for kv in top]))
