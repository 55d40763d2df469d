import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.generator_ref import CompositeGenerator, resample
from text2video_amd.generator import GeneratorSpec, HipGenerator, synthetic_state_dict
from text2video_amd import ops
spec = GeneratorSpec(ngf=32, n_downsample=3, n_blocks=3, no_flow=False, norm="batch")
sd = synthetic_state_dict(spec, 1, "vid2vid")
net = CompositeGenerator(9, 3, 6, 32, 3, 3, False, "batch"); net.load_state_dict(sd, strict=False); net.train()
hip = HipGenerator(spec, "cuda:0").load_state_dict(sd)
rng = np.random.default_rng(0)
H = W = 64
x = torch.from_numpy(np.where(rng.random((1, 9, H, W)) < 0.03, rng.uniform(-1, 1, (1, 9, H, W)), -1.0).astype(np.float32))
p = torch.tanh(torch.from_numpy(rng.standard_normal((1, 6, H, W)).astype(np.float32)))
with torch.no_grad():
    final, flow, wgt, raw, img_feat, flow_feat = net(x, p, False)
    dseg = net.model_down_seg(x); dimg = net.model_down_img(p)
xn = ops.nchw_to_nhwc(x[0].cuda()); pn = ops.nchw_to_nhwc(p[0].cuda())
r = hip.forward(xn, pn, False, want=("out", "raw", "flow_w", "img_feat", "flow_feat"))
def cmp(name, got, want):
    g = got.permute(2, 0, 1).cpu()[:want.shape[0]]
    print("%-10s max|d|=%.3g  ref max=%.3g" % (name, (g - want).abs().max().item(), want.abs().max().item()))
cmp("img_feat", r["img_feat"], img_feat[0]); cmp("raw", r["raw"], raw[0])
cmp("flow_feat", r["flow_feat"], flow_feat[0])
cmp("flow", r["flow_w"][..., :2], flow[0]); cmp("weight", r["flow_w"][..., 2:3], wgt[0])
cmp("final", r["out"], final[0])
# oracle blend from HIP taps
g_raw = r["raw"].permute(2,0,1).cpu()[:3][None]; g_flow = r["flow_w"].permute(2,0,1).cpu()[:2][None]; g_w = r["flow_w"].permute(2,0,1).cpu()[2:3][None]
blend = g_raw * g_w + resample(p[:, -3:], g_flow) * (1 - g_w)
print("blend-from-hip-taps vs hip out", (blend[0] - r["out"].permute(2,0,1).cpu()[:3]).abs().max().item())
