import os, sys, collections
sys.argv = [sys.argv[0]]
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/scripts")
import runpy
# reuse train_bench's setup by exec'ing its head (up to the first timing loop) is fragile: build directly
import numpy as np, torch
from text2video_amd import train as T
from text2video_amd.options import TrainOptions
argv = ["--name", "b", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--max_frames_per_gpu", "2",
        "--n_scales_temporal", "0", "--no_first_img", "--fineSize", "512", "--no_vgg", "--add_face_disc"]
opt = TrainOptions().parse(argv)
dev = "cuda:0"; H = W = 512; F = 2
tr = T.Vid2VidTrainer(opt, dev, seed=1)
rng = np.random.default_rng(0)
pose = torch.zeros(F, H, W, 12, device=dev)
pose[..., :9] = torch.from_numpy(np.where(rng.random((F, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F, H, W, 9)), -1.0).astype(np.float32)).to(dev)
real = torch.zeros(F, H, W, 4, device=dev)
real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F, H, W, 3)).astype(np.float32))).to(dev)
real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
side = 128
boxes = [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * F
prev = torch.zeros(1, H, W, 8, device=dev)
def step():
    return tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat", "aten::sum", "aten::ones_like", "aten::zeros"):
        chain = []
        p = ev.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name[:60]); p = p.cpu_parent
        shp = ""
        st = [f for f in (ev.stack or []) if "text2video_amd" in f][:2]
        cnt[(ev.name, " <- ".join(chain), " | ".join(s.split("/")[-1][:60] for s in st))] += 1
for (n, ch, st), k in cnt.most_common(60):
    print("%4d %-12s %s   [%s]" % (k, n, ch, st))
