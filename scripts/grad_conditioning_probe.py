"""How well-conditioned is the generator's parameter gradient at full width (ngf 128, 9 blocks)?  One frame, linear
loss sum(fake * R) / N.  Compares: CPU oracle fp32 vs CPU oracle fp64 (the yardstick), HIP direct kernels and HIP
Winograd path vs fp64.  Usage: grad_conditioning_probe.py [size] [n_blocks] [init]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.generator_ref import CompositeGenerator
from text2video_amd import train as T
from text2video_amd.generator import GeneratorSpec, synthetic_state_dict

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 9
init = sys.argv[3] if len(sys.argv) > 3 else "vid2vid"
H = W = size
spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=nb, no_flow=True, norm="batch")
sd = synthetic_state_dict(spec, 6, init)
rng = np.random.default_rng(0)
pose = torch.from_numpy(np.where(rng.random((1, 1, H, W)) < 0.02, rng.uniform(-1, 1, (1, 9, H, W)), -1.0).astype(np.float32))
prev = torch.tanh(torch.from_numpy(rng.standard_normal((1, 6, H, W)).astype(np.float32)))
R = torch.from_numpy(rng.standard_normal((1, 3, H, W)).astype(np.float32))


def oracle(dtype):
    net = CompositeGenerator(9, 3, 6, 128, 3, nb, True, "batch").train()
    net.load_state_dict(sd, strict=False)
    net = net.to(dtype)
    out = net(pose.to(dtype), prev.to(dtype), True)[0]
    loss = (out * R.to(dtype)).sum() / R.numel()
    g = torch.autograd.grad(loss, list(net.parameters()))
    return out.detach(), {k: v for (k, _), v in zip(net.named_parameters(), g)}


o64, g64 = oracle(torch.float64)
o32, g32 = oracle(torch.float32)


def hip(env):
    for k in ("T2V_CONV_ALGO", "T2V_WGRAD_WINOGRAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    G = T.TrainableGenerator(spec, sd, "cuda:0")
    p = torch.zeros(1, H, W, 12, device="cuda:0"); p[..., :9] = pose.permute(0, 2, 3, 1).cuda()
    q = torch.zeros(1, H, W, 8, device="cuda:0"); q[..., :6] = prev.permute(0, 2, 3, 1).cuda()
    r = torch.zeros(1, H, W, 4, device="cuda:0"); r[..., :3] = R.permute(0, 2, 3, 1).cuda()
    out = G(p, q)
    loss = (out * r).sum() / R.numel()
    g = torch.autograd.grad(loss, list(G.parameters()), allow_unused=True)
    return out.detach()[..., :3].permute(0, 3, 1, 2).cpu(), {k: v.cpu() for (k, _), v in zip(G.named_upstream_parameters().items(), g)}


def report(name, out, g):
    errs = {}
    for k, r in g64.items():
        sc = r.abs().max().item()
        if sc > 1e-9 and g.get(k) is not None:
            errs[k] = (g[k].double() - r).abs().max().item() / sc
    v = np.array(list(errs.values()))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:2]
    print("%-22s forward max|d| %.1e | grad rel err vs fp64: median %.1e p90 %.1e max %.1e %s" % (
        name, (out.double() - o64).abs().max().item(), np.median(v), np.quantile(v, 0.9), v.max(), ["%s %.0e" % kv for kv in top]))


print("size %d n_blocks %d init %s" % (size, nb, init))
report("CPU oracle fp32", o32, g32)
report("HIP direct", *hip({"T2V_CONV_ALGO": "1"}))
report("HIP F(2x2)", *hip({"T2V_CONV_ALGO": "2"}))
report("HIP F(4x4)", *hip({}))
report("HIP F(4x4) wgrad direct", *hip({"T2V_WGRAD_WINOGRAD": "0"}))
