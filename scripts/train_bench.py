"""Time Vid2VidTrainer.train_step at config-5 size on one GPU (1 sequence per GPU, max_frames_per_gpu 2): generator
(with the flow branch unless --no_flow) forward + backward, 2-scale discriminator (+ face discriminator), all losses,
Adam.  Usage: train_bench.py [--size 512] [--iters 3] [--no_flow] [--vgg] [--no_face]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from text2video_amd import train as T
from text2video_amd.options import TrainOptions

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--no_flow", action="store_true")
ap.add_argument("--no_face", action="store_true")
ap.add_argument("--vgg", action="store_true", help="add the VGG19 perceptual loss (seeded random weights)")
ap.add_argument("--aten_stacks", action="store_true", help="one step under torch.profiler: where the ATen fills / adds / "
                "copies of the step come from (python call sites, by count)")
ap.add_argument("--aten_kernels", action="store_true", help="one step under torch.profiler with stacks: every ATen operator that "
                "launches a fill / add / mul / copy / cat kernel, by the innermost text2video_amd frame that issued it (the autograd "
                "engine's own -- gradient accumulation, materialised zero gradients -- show up as <autograd engine>)")
ap.add_argument("--high_priority", action="store_true", help="run the steps on a high-priority stream (the weight-gradient side "
                "stream keeps the default priority)")
ap.add_argument("--host_time", action="store_true", help="also report when the host has finished ENQUEUEING a step (return of "
                "the last optimiser step, before the losses are read back): host-bound or GPU-bound?")
ap.add_argument("--cprofile", action="store_true", help="three more steps under cProfile: where the host spends its time")
ap.add_argument("--force_dist", action="store_true", help="run the gradient exchange on a 1-rank RCCL group and report its "
                "bytes, buckets and the part still running after the backward pass")
args = ap.parse_args()
if args.force_dist:
    import torch.distributed as dist
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", T2V_TRAIN_FORCE_DIST="1",
                      T2V_TRAIN_COMM_TIMING="1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
argv = ["--name", "b", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--max_frames_per_gpu",
        str(args.frames), "--n_scales_temporal", "0", "--no_first_img", "--fineSize", str(args.size)]
argv += ["--vgg_random_init"] if args.vgg else ["--no_vgg"]
argv += ["--no_flow"] if args.no_flow else []
argv += [] if args.no_face else ["--add_face_disc"]
opt = TrainOptions().parse(argv)
dev = "cuda:0"
H = W = args.size
F = args.frames
tr = T.Vid2VidTrainer(opt, dev, seed=1)
rng = np.random.default_rng(0)
pose = torch.zeros(F, H, W, 12, device=dev)
pose[..., :9] = torch.from_numpy(np.where(rng.random((F, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F, H, W, 9)), -1.0).astype(np.float32)).to(dev)
real = torch.zeros(F, H, W, 4, device=dev)
real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F, H, W, 3)).astype(np.float32))).to(dev)
real_prev = torch.cat([real[1:], real[:1]], 0).contiguous() if not args.no_flow else None
side = max(8, args.size // 32 * 8)
boxes = None if args.no_face else [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * F
prev = torch.zeros(1, H, W, 8, device=dev)
prev[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).to(dev)


_hp = torch.cuda.Stream(priority=-1) if args.high_priority else None


def step():
    if _hp is None:
        return tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
    _hp.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(_hp):
        out = tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
    torch.cuda.current_stream().wait_stream(_hp)
    return out


step(); step(); torch.cuda.synchronize()
host_marks = []
if args.host_time:
    _optD_step = tr.optD.step

    def _marked():
        _optD_step()
        host_marks.append(time.perf_counter())
    tr.optD.step = _marked
t0 = time.perf_counter()
starts = []
for _ in range(args.iters):
    starts.append(time.perf_counter())
    losses = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
if args.host_time:
    print("host has enqueued the step after %.1f ms on average (of %.1f ms per step)"
          % (1e3 * sum(m - s0 for m, s0 in zip(host_marks, starts)) / len(starts), dt * 1e3))
print("train step %dx%d, %d frames, %s%s%s: %.1f ms/step, peak mem %.1f GB | %s"
      % (H, W, F, "no flow" if args.no_flow else "flow branch on", "" if args.no_face else " + face D", " + VGG" if args.vgg else "",
         dt * 1e3, torch.cuda.max_memory_allocated() / 2**30, " ".join("%s %.3f" % kv for kv in losses.items())))
if args.cprofile:
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
    print("\n".join(l[:170] for l in out.getvalue().splitlines()[:60]))
if args.aten_stacks:
    import collections, traceback
    sites = collections.Counter()

    def spy(owner, name):
        orig = getattr(owner, name)

        def wrapped(*a, **k):
            st = [f for f in traceback.extract_stack(limit=8)[:-1] if "text2video_amd" in f.filename]
            sites[(name, " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-2:][::-1]))] += 1
            return orig(*a, **k)
        setattr(owner, name, wrapped)
    for owner, name in [(torch, "zeros"), (torch, "zeros_like"), (torch, "cat"), (torch, "ones_like"), (torch, "full"),
                        (torch.Tensor, "zero_"), (torch.Tensor, "fill_"), (torch.Tensor, "clone"), (torch.Tensor, "contiguous"),
                        (torch.Tensor, "add_"), (torch.Tensor, "__add__"), (torch.Tensor, "__mul__"), (torch.Tensor, "mul_"),
                        (torch.Tensor, "copy_"), (torch.Tensor, "sum"), (torch.Tensor, "mean")]:
        spy(owner, name)
    step()
    torch.cuda.synchronize()
    for (name, where), n in sites.most_common(70):
        print("%4d  %-12s %s" % (n, name, where))
if args.aten_kernels:
    import collections
    from torch.profiler import profile, ProfilerActivity
    step(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::copy_", "aten::cat", "aten::sum",
            "aten::div", "aten::stack", "aten::sub", "aten::neg", "aten::norm", "aten::lt", "aten::_to_copy", "aten::linalg_vector_norm")
    sites = collections.Counter()
    dev_us = collections.Counter()
    for ev in prof.events():
        if ev.name not in want or ev.device_time_total <= 0 and not ev.kernels:
            continue
        if not ev.kernels:
            continue
        fr = [f for f in (ev.stack or []) if "text2video_amd" in f or "train_bench" in f]
        if fr:
            where = fr[0].split("/")[-1]
        else:       # no Python stack in this build: the chain of enclosing profiler ranges (autograd Function names, aten parents)
            chain, q = [], ev.cpu_parent
            while q is not None and len(chain) < 4:
                chain.append(q.name.replace("aten::", ""))
                q = q.cpu_parent
            where = " < ".join(chain) if chain else "<top level>"
        shape = str(ev.input_shapes[0]) if ev.input_shapes else ""
        key = (ev.name, where[:70], shape[:28])
        sites[key] += 1
        dev_us[key] += sum(k.duration for k in ev.kernels)
    tot = sum(dev_us.values())
    print("ATen operators with device kernels in one step: %d launches, %.3f ms of kernel time" % (sum(sites.values()), tot / 1e3))
    for key, n in sorted(sites.items(), key=lambda kv: -dev_us[kv[0]])[:80]:
        print("%4d %8.1f us  %-14s %-30s %s" % (n, dev_us[key], key[0], key[2], key[1]))
if args.force_dist:
    print("gradient exchange (1-rank RCCL%s): %.1f MB per step in %d + %d buckets, %.2f ms still running after the backward pass"
          % (", reduce-scatter + all-gather" if os.environ.get("T2V_GRAD_RS_AG") == "1" else ", all-reduce", tr.comm_bytes / 2**20,
             len(tr.bucketsG.bounds), len(tr.bucketsD.bounds), tr.comm_ms))
    dist.destroy_process_group()
