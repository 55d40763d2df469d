"""Where does the HIP train step's D_f (and G) gradient error come from?  The 512x512 step of tests/test_gpu_device_oracle.py
against the fp64 device oracle, with the step's optimisations switched off one at a time.  Usage: step_parity_probe.py [size]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
spec = importlib.util.spec_from_file_location("tdo", os.path.join(ROOT, "tests", "test_gpu_device_oracle.py"))
tdo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tdo)
from text2video_amd import ops

torch.backends.cudnn.enabled = False
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
full = size >= 256
args = (size, 128, 3, 9, 64) if full else (size, 32, 2, 2, 16)
tr, mods, clip, boxes = tdo._step_setup(*args, seed=5)
l64, g64, f64 = tdo._oracle_step_on(mods, clip, boxes, tdo.DEV, torch.float64)
l32, g32, f32 = tdo._oracle_step_on(mods, clip, boxes, tdo.DEV, torch.float32)
eo = tdo._rel_err(g32, g64)
del tr
variants = [("default", {}), ("weight gradients on the main stream", {"T2V_WGRAD_STREAM": "0", "T2V_PACK_PREFETCH": "0"}),
                  ("autograd accumulation", {"T2V_GRAD_DIRECT": "0"}), ("two D forwards", {"T2V_D_SHARED_FWD": "0"}),
                  ("reduce launch instead of the in-kernel combine", {"T2V_WGRAD_COMBINE": "0"}),
                  ("direct kernels only", {"T2V_CONV_ALGO": "1"}), ("no fixed grid", {"T2V_WINO_GEMM_SK": "0", "T2V_WGRAD_SK": "0"})]
if len(sys.argv) > 2 and sys.argv[2] == "swap":
    variants = variants[:1]
for name, env in variants:
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ops.reload_env()
    tr, _, _, _ = tdo._step_setup(*args, seed=5)
    lh, gh, fh = tdo._hip_step(tr, clip, boxes)
    eh = tdo._rel_err(gh, g64)
    out = []
    for tag in ("G.", "D.", "Df."):
        a = np.array([eh[k] for k in eh if k.startswith(tag)])
        b = np.array([eo[k] for k in eh if k.startswith(tag)])
        out.append("%s HIP med %.1e p90 %.1e max %.1e (oracle32 %.1e %.1e %.1e)" % (tag, np.median(a), np.quantile(a, 0.9), a.max(),
                                                                                 np.median(b), np.quantile(b, 0.9), b.max()))
    print("%-48s %s" % (name, " | ".join(out)), flush=True)
    if name == "default":
        for k in eh:
            if k.startswith("Df."):
                print("      %-36s HIP %.1e  oracle32 %.1e" % (k, eh[k], eo[k]))
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    del tr
    torch.cuda.empty_cache()

# ---- is D_f's gradient error the generated frames' error seen through D_f?  D_f's own loss in fp64 on three versions of the
# fake frames: the fp64 oracle's, the fp32 oracle's, the HIP path's
if full:
    import copy
    os.environ.pop("T2V_CONV_ALGO", None)
    ops.reload_env()
    tr, _, _, _ = tdo._step_setup(*args, seed=5)
    _, _, fh = tdo._hip_step(tr, clip, boxes)
    Dfr = copy.deepcopy(mods[2]).to(device=tdo.DEV, dtype=torch.float64)
    pose, real = clip[0].to(tdo.DEV, torch.float64), clip[1].to(tdo.DEV, torch.float64)
    A = pose[:, 6:9]
    mse = torch.nn.MSELoss()

    def crop(t):
        return torch.stack([t[i, :, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(boxes)])

    def df_grad(fake):
        fr = Dfr(torch.cat([crop(A), crop(real)], 1))
        ff = Dfr(torch.cat([crop(A), crop(fake.to(tdo.DEV, torch.float64))], 1))
        loss = 0.5 * (sum(mse(p[-1], torch.zeros_like(p[-1])) for p in ff) + sum(mse(p[-1], torch.ones_like(p[-1])) for p in fr))
        return {k: g for (k, _), g in zip(Dfr.named_parameters(), torch.autograd.grad(loss, list(Dfr.parameters())))}
    g0, g_32, g_h = df_grad(f64), df_grad(f32), df_grad(fh)
    print("D_f loss gradient in fp64, fake frames swapped (relative change per tensor vs the fp64 frames):")
    for k, r in g0.items():
        s = r.abs().max().item()
        if s > 1e-12:
            print("      %-32s fp32-oracle frames %.1e   HIP frames %.1e" % (k, (g_32[k] - r).abs().max().item() / s, (g_h[k] - r).abs().max().item() / s))
    d32, dh = (f32.double() - f64), (fh.double() - f64)
    print("frame error: fp32 oracle rms %.2e max %.2e | HIP rms %.2e max %.2e" % (d32.pow(2).mean().sqrt().item(), d32.abs().max().item(),
                                                                               dh.pow(2).mean().sqrt().item(), dh.abs().max().item()))
