"""D_f inside the 512x512 HIP train step against D_f alone on the same frames: is the step's D_f gradient an exact function of the
frames it returns?  (tests/test_gpu_device_oracle.py's set-up; T2V_CONV_STEM=1|2 changes the frames' rounding only.)"""
import importlib.util, os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
spec = importlib.util.spec_from_file_location("tdo", os.path.join(ROOT, "tests", "test_gpu_device_oracle.py"))
tdo = importlib.util.module_from_spec(spec); spec.loader.exec_module(tdo)
from text2video_amd import ops, train as T
torch.backends.cudnn.enabled = False
tr, mods, clip, boxes = tdo._step_setup(512, 128, 3, 9, 64, seed=5)
_, g64, _ = tdo._oracle_step_on(mods, clip, boxes, tdo.DEV, torch.float64)
lh, gh, fh = tdo._hip_step(tr, clip, boxes)
DEV = tdo.DEV
mse = torch.nn.MSELoss()
def crop(t): return torch.stack([t[i, :, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(boxes)])
def oracle_df(dtype, fake):
    Dfr = copy.deepcopy(mods[2]).to(device=DEV, dtype=dtype)
    pose, real = clip[0].to(DEV, dtype), clip[1].to(DEV, dtype)
    fr = Dfr(torch.cat([crop(pose[:, 6:9]), crop(real)], 1)); ff = Dfr(torch.cat([crop(pose[:, 6:9]), crop(fake.to(DEV, dtype))], 1))
    loss = 0.5 * (sum(mse(p[-1], torch.zeros_like(p[-1])) for p in ff) + sum(mse(p[-1], torch.ones_like(p[-1])) for p in fr))
    return {"Df." + k: g for (k, _), g in zip(Dfr.named_parameters(), torch.autograd.grad(loss, list(Dfr.parameters())))}
g_ref = oracle_df(torch.float64, fh)
# D_f of a FRESH trainer (same seed: same weights) alone on crops of the returned frames
tr2, _, _, _ = tdo._step_setup(512, 128, 3, 9, 64, seed=5)
def nhwc(t, cs):
    out = torch.zeros(t.shape[0], t.shape[2], t.shape[3], cs, device=DEV); out[..., :t.shape[1]] = t.permute(0, 2, 3, 1).to(DEV); return out
pose_n, real_n, fake_n = nhwc(clip[0], 12), nhwc(clip[1], 4), nhwc(fh, 4)
def cropn(t): return torch.stack([t[i, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(boxes)]).contiguous()
A3 = pose_n[..., 6:9]
params = list(tr2.Df.parameters())
frp = tr2.Df(tr2._d_input(cropn(A3), cropn(real_n)))
ffp = tr2.Df(tr2._d_input(cropn(A3), cropn(fake_n)))
loss = 0.5 * (T.gan_loss(ffp, False) + T.gan_loss(frp, True))
ga = torch.autograd.grad(loss, params, allow_unused=True)
ga = {"Df." + k: v for (k, _), v in zip(tr2.Df.named_upstream_parameters().items(), ga)}
print("T2V_CONV_STEM=%s" % os.environ.get("T2V_CONV_STEM", "1"))
for k, r in g_ref.items():
    s = r.abs().max().item()
    if g64[k].abs().max().item() <= 1e-9: continue
    print("   %-30s step vs fp64-on-its-frames %.1e | D_f alone (HIP) on those frames vs fp64 %.1e | step vs alone %.1e" % (k,
          (gh[k].double() - r).abs().max().item() / s, (ga[k].double() - r).abs().max().item() / s, (gh[k] - ga[k]).abs().max().item() / s))
