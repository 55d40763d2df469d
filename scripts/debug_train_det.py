"""Find nondeterminism in the HIP train step: run generator forward + backward of a fixed loss twice, compare."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import train as T
from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
H = W = 32
spec = GeneratorSpec(ngf=32, n_downsample=2, n_blocks=2, no_flow=True, norm="batch")
sd = synthetic_state_dict(spec, 3, "vid2vid")
rng = np.random.default_rng(1)
pz = torch.zeros(1, H, W, 12, device="cuda:0"); pz[..., :9] = torch.from_numpy(rng.standard_normal((1, H, W, 9)).astype(np.float32)).cuda().clamp(-1, 1)
pv = torch.zeros(1, H, W, 8, device="cuda:0"); pv[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).cuda()
tgt = torch.from_numpy(rng.standard_normal((1, H, W, 4)).astype(np.float32)).cuda()
def run(dual=7):
    os.environ['T2V_DUAL'] = str(dual)
    Gh = T.TrainableGenerator(spec, sd, "cuda:0")
    h1 = Gh(pz, pv)
    pv2 = torch.zeros_like(pv); pv2[..., 0:3] = pv[..., 3:6]; pv2[..., 3:6] = h1.detach()[..., :3]
    h2 = Gh(pz.flip(1).contiguous(), pv2)
    out = torch.cat([h1, h2], 0)
    loss = ((out - tgt) ** 2)[..., :3].mean()
    gs = torch.autograd.grad(loss, list(Gh.parameters()), allow_unused=True)
    return out.detach().clone(), {k: g.clone() for (k, _), g in zip(Gh.named_upstream_parameters().items(), gs) if g is not None}
o0, g0 = run(0)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o1, g1 = run(7 if it % 2 == 0 else 0)
    bad = [(k, (g1[k] - g0[k]).abs().max().item() / (g0[k].abs().max().item() + 1e-30)) for k in g0 if not torch.equal(g0[k], g1[k])]
    bad.sort(key=lambda kv: -kv[1])
    print("run", it, "dual", 7 if it % 2 == 0 else 0, "forward equal:", torch.equal(o0, o1), "differing grads:", len(bad), "of", len(g0), [("%s %.1e" % kv) for kv in bad[:6]])
