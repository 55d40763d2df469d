"""One train step with the direct weight gradients of the clip's two frames reduced per frame (T2V_WGRAD_PAIR=0) and in one
launch per layer (=1), same seed: per-parameter difference of the gradients the step delivered (G and D buckets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from text2video_amd import train as T
from text2video_amd.options import TrainOptions

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
argv = ["--name", "b", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--max_frames_per_gpu", "2",
        "--n_scales_temporal", "0", "--no_first_img", "--fineSize", str(size), "--no_vgg", "--add_face_disc"]
dev = "cuda:0"
H = W = size
F = 2
rng = np.random.default_rng(0)
pose = torch.zeros(F, H, W, 12, device=dev)
pose[..., :9] = torch.from_numpy(np.where(rng.random((F, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F, H, W, 9)), -1.0).astype(np.float32)).to(dev)
real = torch.zeros(F, H, W, 4, device=dev)
real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F, H, W, 3)).astype(np.float32))).to(dev)
real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
side = max(8, size // 32 * 8)
boxes = [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * F
prev = torch.zeros(1, H, W, 8, device=dev)
prev[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).to(dev)
res = {}
for mode in ("0", "1"):
    os.environ["T2V_WGRAD_PAIR"] = mode
    tr = T.Vid2VidTrainer(TrainOptions().parse(argv), dev, seed=1)
    tr.optG.step = lambda: None      # keep the weights: only the gradients of this one step are compared
    tr.optD.step = lambda: None
    losses = tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
    torch.cuda.synchronize()
    res[mode] = ({n: b for n, b in (("G", tr.bucketsG), ("D", tr.bucketsD))}, losses, tr)
for net in ("G", "D"):
    b0, b1 = res["0"][0][net], res["1"][0][net]
    worst = []
    for i, (s0, s1) in enumerate(zip(b0.slots, b1.slots)):
        if s0 is None or s1 is None:
            continue
        a, b = s0.view.double(), s1.view.double()
        d = (a - b).norm().item() / max(a.norm().item(), 1e-30)
        worst.append((d, i, tuple(b0.params[i].shape)))
    worst.sort(reverse=True)
    print(net, "params", len(worst), "largest relative L2 differences:", [(("%.2e" % d), i, shp) for d, i, shp in worst[:6]])
print("losses 0:", {k: round(float(v), 5) for k, v in res["0"][1].items()})
print("losses 1:", {k: round(float(v), 5) for k, v in res["1"][1].items()})
