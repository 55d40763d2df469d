"""Full-size parity against a device-side second opinion.

The CPU oracle needs minutes for a 1024x1024 single-scale frame or a 512x512 two-frame train step, so round 3 checked those
sizes HIP-against-HIP (Winograd family vs direct family).  Here the SAME oracle modules (oracle/generator_ref.py: stock torch.nn)
are evaluated on the GPU by torch's own native operators -- with the MIOpen backend switched off every convolution is ATen's
im2col + rocBLAS GEMM (sgemm / dgemm: no run-time kernel compilation, the same code path in fp32 and fp64), nothing of
libt2v_hip.so -- after a test has shown that the device evaluation IS the CPU oracle at a size the CPU finishes in seconds.
Then:
  * the single-scale 1024x1024 frame (configs[3]) against the device oracle, |delta| <= 1e-3 per pixel;
  * the whole configs[4] step at 512x512 (generator with its flow branch, 2-scale D, face D, every loss): every loss and every
    parameter gradient of G (flow branch included), D and D_f against the oracle step in fp64, bounded by a small multiple of the
    fp32 oracle's own distance from that fp64 evaluation (the conditioning-normalised bound of
    test_fullwidth_gradient_error_is_within_the_fp32_oracles_own);
  * the generator's gradient at full width and full size WITH the flow branch (flows of a few pixels through the compositor).
"""
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _exact_fp32():
    """the second opinion: ATen's native kernels + rocBLAS in exact fp32 / fp64 (no MIOpen: its solvers would be compiled at
    run time on a fresh box, and the fp64 path has none)"""
    old = (torch.backends.cudnn.enabled, torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32)
    torch.backends.cudnn.enabled = False
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield
    torch.backends.cudnn.enabled, torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old


def _pose(n, H, W, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(np.where(rng.random((n, 1, H, W)) < 0.02, rng.uniform(-1, 1, (n, 9, H, W)), -1.0).astype(np.float32))


def _frames(n, c, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.tanh(torch.randn(n, c, H, W, generator=g))


def _generator_pair(no_flow, seed=1, flow_gain=0.1, init="uniform_fan_in"):
    from oracle.generator_ref import CompositeGenerator
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=no_flow, norm="batch")
    sd = synthetic_state_dict(spec, seed, init, flow_gain=flow_gain)
    ref = CompositeGenerator(9, 3, 6, 128, 3, 9, no_flow, "batch").train()
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not unexpected and all(("running" in k or "num_batches" in k) for k in missing)
    return spec, sd, ref


# ------------------------------------------------------------------------------------------------------------------
# configs[3], single scale
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("no_flow", [True, False], ids=["noflow", "flow"])
def test_config3_single_scale_1024_frame_matches_the_device_oracle(no_flow):
    """G0 at 1024x1024 (10.3 / 13.3 TFLOP per frame).  Step 1: at 256x256 the oracle evaluated on the GPU reproduces the CPU
    oracle (same modules, same weights, same inputs) -- measured ~1e-5, gate 1e-4 = a tenth of the frame tolerance.  Step 2:
    the HIP frame at 1024x1024 against the device oracle, teacher-forced, |delta| <= 1e-3 per pixel, flow maps included."""
    from text2video_amd import ops
    from text2video_amd.generator import HipGenerator
    spec, sd, ref = _generator_pair(no_flow)
    H = W = 256
    x, p = _pose(1, H, W, 41), _frames(1, 6, H, W, 42)
    with torch.no_grad():
        cpu = ref(x, p, False)
        ref = ref.to(DEV)
        dev = ref(x.to(DEV), p.to(DEV), False)
    d_small = (dev[0].cpu() - cpu[0]).abs().max().item()
    print("256x256 %s frame: device oracle vs CPU oracle max|delta| = %.3g" % ("no-flow" if no_flow else "flow", d_small))
    assert d_small <= 1e-4 and cpu[0].abs().max().item() > 0.05
    hip = HipGenerator(spec, DEV).load_state_dict(sd)
    got_small = hip.forward(ops.nchw_to_nhwc(x[0].to(DEV)), ops.nchw_to_nhwc(p[0].to(DEV), 8), False)["out"]
    assert (got_small[..., :3].permute(2, 0, 1).cpu() - cpu[0][0]).abs().max().item() <= 1e-3
    # ---- full size ----
    H = W = 1024
    x, p = _pose(1, H, W, 43).to(DEV), _frames(1, 6, H, W, 44).to(DEV)
    with torch.no_grad():
        want = ref(x, p, False)
    res = hip.forward(ops.nchw_to_nhwc(x[0]), ops.nchw_to_nhwc(p[0], 8), False, want=("out",) if no_flow else ("out", "flow_w", "raw"))
    err = (res["out"][..., :3].permute(2, 0, 1) - want[0][0]).abs().max().item()
    print("single-scale 1024x1024 %s frame vs the device oracle: max|delta| = %.3g" % ("no-flow" if no_flow else "flow", err))
    assert err <= 1e-3 and want[0].abs().max().item() > 0.05 and want[0].std().item() > 0.01
    if not no_flow:
        ferr = (res["flow_w"][..., :2].permute(2, 0, 1) - want[1][0]).abs().max().item()
        werr = (res["flow_w"][..., 2] - want[2][0, 0]).abs().max().item()
        rerr = (res["raw"][..., :3].permute(2, 0, 1) - want[3][0]).abs().max().item()
        print("   flow %.3g px (|flow| max %.1f px), weight %.3g, raw %.3g" % (ferr, want[1].abs().max().item(), werr, rerr))
        assert ferr <= 5e-3 and werr <= 1e-3 and rerr <= 1e-3


# ------------------------------------------------------------------------------------------------------------------
# the train step restated on the oracle modules (any device, any dtype)
# ------------------------------------------------------------------------------------------------------------------
def oracle_train_step(Gr, Dr, Dfr, pose, real, real_prev, prev0, boxes, lam_feat=10.0, lam_F=10.0, lam_T=10.0, n_layers=3):
    """Vid2VidTrainer._train_step (text2video_amd/train.py; [RECALL upstream Vid2VidModelD.forward]) on torch autograd:
    a continuing chunk (prev0 given), flow branch on, zero reference flow with the ||real - real_prev|| < 0.02 confidence
    rule, LSGAN + feature matching on the blended AND the raw frames, face D on the boxes (face terms x2 on G's side).
    pose [F,9,H,W], real / real_prev [F,3,H,W], prev0 [1,6,H,W]; boxes [(ys,ye,xs,xe)] per frame.
    -> (losses, {name: gradient} for G, D, D_f)"""
    from oracle.generator_ref import resample
    mse, l1 = torch.nn.MSELoss(), torch.nn.L1Loss()

    def gan(pred, is_real):
        return sum(mse(p[-1], torch.ones_like(p[-1]) if is_real else torch.zeros_like(p[-1])) for p in pred)

    def fm(pf, pr):
        nd = len(pf)
        return sum(l1(pf[i][j], pr[i][j].detach()) * ((1.0 / nd) * (4.0 / (n_layers + 1)) * lam_feat)
                   for i in range(nd) for j in range(len(pf[i]) - 1))

    def ml1(a, b, m):
        m = m.expand(-1, a.shape[1], -1, -1)
        return l1(a * m, b * m)

    F_ = pose.shape[0]
    prev, outs, prevs = prev0, [], []
    for f in range(F_):
        o = Gr(pose[f:f + 1], prev, False)
        outs.append(o)
        prevs.append(prev)
        prev = torch.cat([prev[:, 3:], o[0].detach()], 1)
    fake, flow, weight, raw = (torch.cat([o[k] for o in outs], 0) for k in range(4))
    A = pose[:, 6:9]
    pr = Dr(torch.cat([A, real], 1))
    pfd, pfg = Dr(torch.cat([A, fake.detach()], 1)), Dr(torch.cat([A, fake], 1))
    pfd_r, pfg_r = Dr(torch.cat([A, raw.detach()], 1)), Dr(torch.cat([A, raw], 1))
    loss_D = 0.5 * ((gan(pfd, False) + gan(pfd_r, False)) + (gan(pr, True) + gan(pr, True)))
    l_gan = gan(pfg, True) + gan(pfg_r, True)
    l_fm = fm(pfg, pr) + fm(pfg_r, pr)
    conf = ((real - real_prev).norm(dim=1, keepdim=True) < 0.02).to(real.dtype)
    zero = torch.zeros_like(flow)
    fake_prev = torch.cat([q[:, 3:] for q in prevs], 0)
    l_flow = ml1(flow, zero, conf) * lam_F
    l_fwarp = ml1(resample(real_prev, flow), real, conf) * lam_T
    l_w = ml1(weight, torch.zeros_like(weight), conf)
    l_gwarp = ml1(fake, resample(fake_prev, zero).detach(), conf) * lam_T

    def crop(t):
        return torch.stack([t[i, :, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(boxes)])
    fr = Dfr(torch.cat([crop(A), crop(real)], 1))
    ffd, ffg = Dfr(torch.cat([crop(A), crop(fake.detach())], 1)), Dfr(torch.cat([crop(A), crop(fake)], 1))
    l_df = 0.5 * (gan(ffd, False) + gan(fr, True))
    l_fgan, l_ffm = gan(ffg, True) * 2.0, fm(ffg, fr) * 2.0
    loss_G = l_gan + l_fm + l_flow + l_fwarp + l_w + l_gwarp + l_fgan + l_ffm
    gG = torch.autograd.grad(loss_G, list(Gr.parameters()), retain_graph=True)
    d_params = list(Dr.parameters()) + list(Dfr.parameters())
    gD = torch.autograd.grad(loss_D + l_df, d_params)
    grads = {"G." + k: g for (k, _), g in zip(Gr.named_parameters(), gG)}
    names = ["D." + k for k, _ in Dr.named_parameters()] + ["Df." + k for k, _ in Dfr.named_parameters()]
    grads.update(dict(zip(names, gD)))
    losses = {"G_GAN": l_gan, "G_GAN_Feat": l_fm, "D": loss_D, "F_Flow": l_flow, "F_Warp": l_fwarp, "W": l_w, "G_Warp": l_gwarp,
              "G_f_GAN": l_fgan, "G_f_GAN_Feat": l_ffm, "D_f": l_df}
    return {k: float(v.detach()) for k, v in losses.items()}, {k: v.detach() for k, v in grads.items()}, fake.detach()


def _step_setup(size, ngf, n_down, n_blocks, ndf, seed, data_seed=0):
    """a Vid2VidTrainer (HIP) with its optimiser steps patched away, the oracle modules holding the same weights, one clip
    (data_seed: another clip)"""
    from oracle.generator_ref import CompositeGenerator, MultiscaleDiscriminator
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2",
                                "--no_vgg", "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img",
                                "--add_face_disc", "--fineSize", str(size), "--ngf", str(ngf), "--n_downsample_G", str(n_down),
                                "--n_blocks", str(n_blocks), "--ndf", str(ndf)])
    tr = T.Vid2VidTrainer(opt, DEV, seed=seed)
    tr.optG.step = lambda: None
    tr.optD.step = lambda: None
    # flows of a few pixels, as a trained network predicts: a random-init flow head x20 throws every sample tens of pixels
    # away, where a 1e-6 difference in the flow picks other bilinear taps and two correct fp32 evaluations stop agreeing
    with torch.no_grad():
        for k, p in tr.G.named_upstream_parameters().items():
            if k.startswith("model_final_flow."):
                p.mul_(0.1)
                T.invalidate_packs(p)

    def load(mod, named):
        sd = {k: v.detach().cpu().clone() for k, v in named.items()}
        missing, unexpected = mod.load_state_dict(sd, strict=False)
        assert not unexpected and all(("running" in k or "num_batches" in k) for k in missing), (missing, unexpected)
        return mod.train()
    Gr = load(CompositeGenerator(9, 3, 6, ngf, n_down, n_blocks, False, "batch"), tr.G.named_upstream_parameters())
    Dr = load(MultiscaleDiscriminator(6, ndf, 3, 2, "batch"), tr.D.named_upstream_parameters())
    Dfr = load(MultiscaleDiscriminator(6, ndf, 3, 1, "batch"), tr.Df.named_upstream_parameters())
    H = W = size
    pose = _pose(2, H, W, 51 + data_seed)
    real = _frames(2, 3, H, W, 52 + data_seed)
    real_prev = torch.cat([_frames(1, 3, H, W, 53 + data_seed), real[:1]], 0)
    real_prev[:, :, :, : W // 2] = real[:, :, :, : W // 2]        # a region where the zero reference flow is "confident"
    prev0 = _frames(1, 6, H, W, 54 + data_seed)
    side = max(8, size // 32 * 8)
    boxes = [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * 2
    return tr, (Gr, Dr, Dfr), (pose, real, real_prev, prev0), boxes


def _hip_step(tr, clip, boxes):
    pose, real, real_prev, prev0 = clip

    def nhwc(t, cs):
        out = torch.zeros(t.shape[0], t.shape[2], t.shape[3], cs, device=DEV)
        out[..., :t.shape[1]] = t.permute(0, 2, 3, 1).to(DEV)
        return out
    with contextlib.redirect_stdout(None):
        losses, fifo = tr.train_step(nhwc(pose, 12), nhwc(real, 4), boxes, nhwc(prev0, 8), real_prev=nhwc(real_prev, 4))
    fake = fifo[0, ..., :6].permute(2, 0, 1).reshape(2, 3, fifo.shape[1], fifo.shape[2])     # the FIFO after two frames = the fakes
    grads = {}
    for tag, net in (("G.", tr.G), ("D.", tr.D), ("Df.", tr.Df)):
        for k, p in net.named_upstream_parameters().items():
            grads[tag + k] = None if p.grad is None else p.grad.detach().clone()
    return losses, grads, fake


def _oracle_step_on(mods, clip, boxes, device, dtype):
    import copy
    mods = [copy.deepcopy(m).to(device=device, dtype=dtype) for m in mods]
    clip = [t.to(device=device, dtype=dtype) for t in clip]
    return oracle_train_step(mods[0], mods[1], mods[2], clip[0], clip[1], clip[2], clip[3], boxes)


def _rel_err(g, ref, zero_ref=None):
    """per tensor: max |g - ref| / max |ref| over the tensors whose gradient is not mathematically zero (conv biases in front
    of a norm layer: <= 1e-9 in the fp64 evaluation `zero_ref`, rounding noise in any fp32 one)"""
    out = {}
    zero_ref = ref if zero_ref is None else zero_ref
    for k, r in ref.items():
        if zero_ref[k].abs().max().item() <= 1e-9 * max(1.0, max(v.abs().max().item() for v in zero_ref.values())):
            continue
        assert g[k] is not None, k
        out[k] = (g[k].double().cpu() - r.double().cpu()).abs().max().item() / r.abs().max().item()
    return out


def test_device_oracle_train_step_is_the_cpu_oracle_step_and_the_hip_step_matches_both_at_64():
    """The whole step at a size the CPU oracle finishes in seconds (64x64, ngf 32): (i) the oracle step evaluated on the GPU
    gives the CPU oracle's losses and gradients (fp32: 1e-4 relative, fp64: 1e-9) -- the licence for using it as the reference
    at 512x512; (ii) Vid2VidTrainer.train_step (HIP) against the CPU oracle step: every loss, every parameter gradient of G
    (flow branch included), D and D_f."""
    tr, mods, clip, boxes = _step_setup(64, 32, 2, 2, 16, seed=5)
    l_cpu, g_cpu, _ = _oracle_step_on(mods, clip, boxes, "cpu", torch.float32)
    l_dev, g_dev, _ = _oracle_step_on(mods, clip, boxes, DEV, torch.float32)
    l_cpu64, g_cpu64, f_cpu64 = _oracle_step_on(mods, clip, boxes, "cpu", torch.float64)
    l_dev64, g_dev64, _ = _oracle_step_on(mods, clip, boxes, DEV, torch.float64)
    e32 = _rel_err(g_dev, g_cpu, g_cpu64)
    e64 = _rel_err(g_dev64, g_cpu64)
    print("64x64 step, device oracle vs CPU oracle: fp32 gradients max rel %.2e (median %.1e), fp64 max rel %.2e"
          % (max(e32.values()), float(np.median(list(e32.values()))), max(e64.values())))
    assert max(e64.values()) <= 1e-9
    # (fp32 on two devices: different summation orders, through a bilinear warp -- the worst tensor a few 1e-3)
    assert float(np.median(list(e32.values()))) <= 1e-4 and max(e32.values()) <= 1e-2
    for k in l_cpu:
        assert abs(l_dev[k] - l_cpu[k]) <= 1e-5 * max(1.0, abs(l_cpu[k])), k
        assert abs(l_dev64[k] - l_cpu64[k]) <= 1e-11 * max(1.0, abs(l_cpu64[k])), k
    l_hip, g_hip, f_hip = _hip_step(tr, clip, boxes)
    assert (f_hip.double().cpu() - f_cpu64).abs().max().item() <= 2e-4
    assert l_cpu["F_Warp"] > 0 and l_cpu["W"] > 0 and l_cpu["G_f_GAN"] > 0
    for k in l_cpu:
        assert abs(l_hip[k] - l_cpu64[k]) <= 2e-4 * max(1.0, abs(l_cpu64[k])), (k, l_hip[k], l_cpu64[k])
    eh, eo = _rel_err(g_hip, g_cpu64), _rel_err(g_cpu, g_cpu64)
    for tag in ("G.", "D.", "Df."):
        a = np.array([v for k, v in eh.items() if k.startswith(tag)])
        b = np.array([v for k, v in eo.items() if k.startswith(tag)])
        print("   %-3s HIP vs fp64: median %.1e max %.1e | CPU fp32 oracle vs fp64: median %.1e max %.1e" % (tag, np.median(a), a.max(), np.median(b), b.max()))
        assert a.max() <= max(3e-3, 2 * b.max()) and np.median(a) <= max(4e-4, 2 * np.median(b))
    assert any(k.startswith("G.model_res_flow") for k in eh) and any(k.startswith("G.model_final_w") for k in eh)


def test_second_step_with_kept_input_transforms_matches_the_device_oracle_at_128():
    """From a layer's SECOND step on the forward convs leave their input transforms in the weight gradient's workspace, the
    transform of dy forms the gradient in front of the norm itself, and the data gradients read the forward packing of the
    weights in place (train._keep_v_slot, T2V_DY_NORM_FUSED, T2V_DGRAD_FORWARD_WEIGHTS) -- paths the one-step oracle tests
    never enter.  Here two steps on the same weights (the optimiser steps are patched away) and the same clip, at a size
    whose bottleneck (32 x 32 x 128) runs F(4x4,3x3): the second step -- checked to have taken the kept path -- against the
    oracle step evaluated on the GPU in fp64 (licensed by the 64x64 test above), to the bounds the first step meets."""
    from text2video_amd import ops
    tr, mods, clip, boxes = _step_setup(128, 32, 2, 2, 16, seed=7)
    l64, g64, f64 = _oracle_step_on(mods, clip, boxes, DEV, torch.float64)
    l32, g32, _ = _oracle_step_on(mods, clip, boxes, DEV, torch.float32)
    eo = _rel_err(g32, g64)
    kept = {"n": 0}
    dy_norm = ops.conv2d_backward_weight_winograd_dy_norm
    ops.conv2d_backward_weight_winograd_dy_norm = lambda *a, **k: (kept.__setitem__("n", kept["n"] + 1), dy_norm(*a, **k))[1]
    try:
        steps = [_hip_step(tr, clip, boxes) for _ in range(2)]
        counts = kept["n"]
    finally:
        ops.conv2d_backward_weight_winograd_dy_norm = dy_norm
    assert counts > 0, "the second step did not take the kept-V path"
    for n, (l_hip, g_hip, f_hip) in enumerate(steps):
        assert (f_hip.double() - f64).abs().max().item() <= 2e-4
        for k in l64:
            assert abs(l_hip[k] - l64[k]) <= 2e-4 * max(1.0, abs(l64[k])), (n, k, l_hip[k], l64[k])
        eh = _rel_err(g_hip, g64)
        for tag in ("G.", "D.", "Df."):
            a = np.array([v for k, v in eh.items() if k.startswith(tag)])
            b = np.array([eo[k] for k in eh if k.startswith(tag)])
            print("128x128 step %d %-3s vs fp64: HIP median %.1e max %.1e | fp32 device oracle median %.1e max %.1e"
                  % (n + 1, tag, np.median(a), a.max(), np.median(b), b.max()))
            assert a.max() <= max(3e-3, 2 * b.max()) and np.median(a) <= max(4e-4, 2 * np.median(b)), (n, tag)


def _df_on_hip_frames(mods, clip, boxes, f_hip, g_hip, seed):
    # D_f by itself, on the frames the HIP generator produced: its gradient evaluated in fp64 there is what the HIP step must
    # deliver -- the generated frames' 2e-5 rms error seen through D_f is the whole of the D_f error above
    # (scripts/step_parity_probe.py).  Every tensor within a few fp32 roundings of it -- except where a LeakyReLU kink flips: D_f
    # is a LeakyReLU / batch-norm stack on a 128x128 crop (a million pre-activations per pass), and one of them within fp32
    # rounding of zero takes the other slope in an fp32 evaluation: with the norm's beta = 0 that element has xhat = 0, so it
    # moves that channel's SUM of dy (the norm bias gradient, and through the norm's mean(dy) term everything upstream of it:
    # layer 1's conv weight, layer 0) by 0.8 |dy| and leaves dgamma and the later layers alone.  Measured with
    # scripts/df_step_probe.py: 2.6e-6 on every tensor on one set of frames; on frames that differ from those by the rounding
    # of the stems' k order 1e-6 on nine tensors and 4e-4 .. 2e-3 on exactly those four -- the step's gradient being, bit for
    # bit, what D_f alone computes on the returned frames.  The fp32 oracle is subject to the same event on other elements
    # (~8 % per pass and implementation).  So: the median against the fp32 oracle's own, and a cap on what one flip can do.
    import copy
    from oracle.generator_ref import MultiscaleDiscriminator      # noqa: F401
    mse = torch.nn.MSELoss()

    def crop(t):
        return torch.stack([t[i, :, bb[0]:bb[1], bb[2]:bb[3]] for i, bb in enumerate(boxes)])

    def df_gradients(dtype):
        Dfr = copy.deepcopy(mods[2]).to(device=DEV, dtype=dtype)
        pose_t, real_t, fake_t = clip[0].to(DEV, dtype), clip[1].to(DEV, dtype), f_hip.to(DEV, dtype)
        fr = Dfr(torch.cat([crop(pose_t[:, 6:9]), crop(real_t)], 1))
        ff = Dfr(torch.cat([crop(pose_t[:, 6:9]), crop(fake_t)], 1))
        l_df = 0.5 * (sum(mse(q[-1], torch.zeros_like(q[-1])) for q in ff) + sum(mse(q[-1], torch.ones_like(q[-1])) for q in fr))
        return {"Df." + k: g for (k, _), g in zip(Dfr.named_parameters(), torch.autograd.grad(l_df, list(Dfr.parameters())))}
    g_df, g_df32 = df_gradients(torch.float64), df_gradients(torch.float32)
    zero_ref = g_df      # (conv biases in front of a norm layer: an exactly zero gradient, <= 1e-9 in fp64)
    e_df, e_df32 = _rel_err({k: g_hip[k] for k in g_df}, g_df, zero_ref), _rel_err(g_df32, g_df, zero_ref)
    a, b = np.array([e_df[k] for k in e_df]), np.array([e_df32[k] for k in e_df])
    print("   D_f on the HIP frames vs its fp64 gradient there: HIP median %.1e max %.1e | fp32 oracle median %.1e max %.1e"
          % (np.median(a), a.max(), np.median(b), b.max()))
    # (round 5: with the frames of the strip-form 7x7 head -- another summation order, another 2e-5 of rounding -- a kink flips
    # in an EARLY layer, for the fp32 oracle evaluated on the same frames exactly as for the HIP step: both 6.2e-4 median /
    # 4.8e-3 max over the 13 tensors.  The count is therefore of tensors where the HIP step is worse than 3x what the fp32
    # oracle itself does on these frames.  Round 6: held on three seeds -- the per-seed counts are printed and kept in
    # profiles/r06_df_bound_seeds.txt)
    n_hip, n_o32 = int(np.sum(a > np.maximum(1e-4, 3 * b))), int(np.sum(b > 1e-4))
    print("   seed %d: D_f tensors above max(1e-4, 3 x fp32 oracle): HIP %d of %d | fp32 oracle above 1e-4: %d of %d"
          % (seed, n_hip, len(a), n_o32, len(b)))
    # (seed 6: a kink flips for BOTH fp32 evaluations on the same elements -- HIP and fp32 oracle 3.0e-3 / 2.9e-3 ... 3.79e-2 /
    # 3.79e-2 tensor by tensor, no tensor worse than 3x the oracle's -- so the cap on the worst tensor is relative to the fp32
    # oracle's own worst as well: an absolute 1e-2 alone would reject the oracle itself there)
    assert np.median(a) <= max(1e-5, 3 * np.median(b)) and n_hip <= 5 and a.max() <= max(1e-2, 2 * b.max())


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_config4_face_discriminator_gradient_on_the_hip_frames(seed):
    """The D_f criterion of the 512x512 step (see _df_on_hip_frames) on three seeds: weights, clip and so the generated frames
    differ per seed; the bound -- at most 5 of D_f's 13 tensors more than max(1e-4, 3 x the fp32 oracle's own error) away from
    D_f's fp64 gradient on the HIP frames, none beyond 1e-2 -- holds on each.  (The full-step comparison runs on seed 5:
    test_config4_train_step_512_matches_the_device_oracle_step.)"""
    tr, mods, clip, boxes = _step_setup(512, 128, 3, 9, 64, seed=seed, data_seed=10 * (seed - 5))
    _, g_hip, f_hip = _hip_step(tr, clip, boxes)
    _df_on_hip_frames(mods, clip, boxes, f_hip, g_hip, seed)


def test_config4_train_step_512_matches_the_device_oracle_step():
    """BASELINE configs[4], one GPU's work, at full size and full width: Vid2VidTrainer.train_step (2 frames of 512x512,
    ngf 128 / 9 blocks with the flow branch, 2-scale D, face D on 128x128 crops, all losses) against the oracle step on the GPU.
    fp64 evaluation = the yardstick; the fp32 device oracle = "another correct fp32 implementation".  At this width the
    parameter gradients are ill-conditioned in fp32, so the bound is relative to the fp32 oracle's own distance
    from fp64: the HIP gradients must be within 3x (median) / 4x (90th percentile) / 5x (worst tensor) of it -- for G, whose
    flow branch (model_res_flow, model_up_flow, the two heads, the compositor's adjoint) is part of it, and for D and D_f."""
    tr, mods, clip, boxes = _step_setup(512, 128, 3, 9, 64, seed=5)
    l64, g64, f64 = _oracle_step_on(mods, clip, boxes, DEV, torch.float64)
    l32, g32, f32 = _oracle_step_on(mods, clip, boxes, DEV, torch.float32)
    torch.cuda.empty_cache()
    l_hip, g_hip, f_hip = _hip_step(tr, clip, boxes)
    b = boxes[0]
    d_h, d_o = (f_hip.double() - f64).abs(), (f32.double() - f64).abs()
    print("512x512 step, generated frames vs fp64: HIP max %.2e (face crop %.2e) | fp32 device oracle max %.2e (face crop %.2e)"
          % (d_h.max().item(), d_h[:, :, b[0]:b[1], b[2]:b[3]].max().item(), d_o.max().item(), d_o[:, :, b[0]:b[1], b[2]:b[3]].max().item()))
    assert d_h.max().item() <= 1e-3
    assert l64["F_Warp"] > 0 and l64["W"] > 0 and l64["F_Flow"] > 0
    for k in l64:
        d_h, d_o = abs(l_hip[k] - l64[k]), abs(l32[k] - l64[k])
        print("   loss %-13s fp64 %.6f  HIP %+.1e  fp32 oracle %+.1e" % (k, l64[k], l_hip[k] - l64[k], l32[k] - l64[k]))
        assert d_h <= max(5 * d_o, 1e-4 * max(1.0, abs(l64[k]))), (k, l_hip[k], l64[k], l32[k])
    eh, eo = _rel_err(g_hip, g64), _rel_err(g32, g64)
    for tag in ("G.", "D.", "Df."):
        a = np.array([eh[k] for k in eh if k.startswith(tag)])
        b = np.array([eo[k] for k in eh if k.startswith(tag)])
        print("512x512 step %-3s vs fp64: HIP median %.1e p90 %.1e max %.1e | fp32 device oracle median %.1e p90 %.1e max %.1e"
              % (tag, np.median(a), np.quantile(a, 0.9), a.max(), np.median(b), np.quantile(b, 0.9), b.max()))
        for k in sorted(eh, key=lambda k: -eh[k])[:3] if tag != "Df." else [k for k in eh if k.startswith(tag)]:
            if k.startswith(tag):
                print("      %-40s HIP %.1e  fp32 oracle %.1e" % (k, eh[k], eo[k]))
        # D_f's gradient responds to the generated frames' rounding error (2e-5 rms in either fp32 implementation) very
        # unevenly: 1e-5 for the fp32 oracle's error pattern, 2e-3 for any other -- it is compared exactly, on the HIP
        # frames, below
        if tag == "Df.":
            continue        # (checked exactly below)
        assert np.median(a) <= 3 * np.median(b) and np.quantile(a, 0.9) <= 4 * np.quantile(b, 0.9) and a.max() <= 5 * b.max(), tag
    _df_on_hip_frames(mods, clip, boxes, f_hip, g_hip, 5)
    flow_keys = [k for k in eh if k.startswith(("G.model_res_flow", "G.model_up_flow", "G.model_final_flow", "G.model_final_w"))]
    a = np.array([eh[k] for k in flow_keys])
    b = np.array([eo[k] for k in flow_keys])
    print("   flow branch alone (%d tensors): HIP median %.1e max %.1e | fp32 oracle median %.1e max %.1e"
          % (len(flow_keys), np.median(a), a.max(), np.median(b), b.max()))
    assert len(flow_keys) >= 30 and np.median(a) <= 4 * np.median(b) and a.max() <= 5 * b.max()


@pytest.mark.parametrize("no_flow", [False, True], ids=["flow", "noflow"])
def test_fullwidth_generator_gradient_512_against_the_device_oracle(no_flow):
    """The generator alone at 512x512 (ngf 128, 9 blocks; linear loss on the output frame, so the flow branch is reached only
    through the compositor out = raw*w + warp(prev, flow)*(1-w) with flows of a few pixels): HIP gradient vs the device oracle in
    fp64, bounded by the fp32 device oracle's own error -- test_fullwidth_gradient_error_is_within_the_fp32_oracles_own[512]
    with the flow branch ON (and, for comparison, off), on the reference the 256x256 CPU variant of that test licenses."""
    import copy
    from text2video_amd import train as T
    spec, sd, ref = _generator_pair(no_flow, seed=6, init="vid2vid")
    H = W = 512
    pose, prev, R = _pose(1, H, W, 61), _frames(1, 6, H, W, 62), torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(63))

    def oracle(dtype):
        net = copy.deepcopy(ref).to(device=DEV, dtype=dtype)
        out = net(pose.to(DEV, dtype), prev.to(DEV, dtype), False)[0]
        g = torch.autograd.grad((out * R.to(DEV, dtype)).sum() / R.numel(), list(net.parameters()))
        return out.detach(), {k: v for (k, _), v in zip(net.named_parameters(), g)}
    o64, g64 = oracle(torch.float64)
    o32, g32 = oracle(torch.float32)
    torch.cuda.empty_cache()
    G = T.TrainableGenerator(spec, sd, DEV)
    p = torch.zeros(1, H, W, 12, device=DEV)
    p[..., :9] = pose.permute(0, 2, 3, 1).to(DEV)
    q = torch.zeros(1, H, W, 8, device=DEV)
    q[..., :6] = prev.permute(0, 2, 3, 1).to(DEV)
    r = torch.zeros(1, H, W, 4, device=DEV)
    r[..., :3] = R.permute(0, 2, 3, 1).to(DEV)
    params = list(G.parameters())
    with T.batched_weight_gradients(params):
        out = G(p, q)
        gh = torch.autograd.grad((out * r).sum() / R.numel(), params, allow_unused=True)
        gh = T.flush_pending_weight_gradients(params, gh)
    gh = {k: v for (k, _), v in zip(G.named_upstream_parameters().items(), gh)}
    assert (out.detach()[..., :3].permute(0, 3, 1, 2).double() - o64).abs().max().item() <= 5e-4
    eh, eo = _rel_err(gh, g64), _rel_err(g32, g64)
    a, b = np.array([eh[k] for k in eh]), np.array([eo[k] for k in eh])
    print("512x512 generator gradient (%s) vs fp64: HIP median %.1e p90 %.1e max %.1e | fp32 device oracle median %.1e p90 %.1e max %.1e"
          % ("no flow" if no_flow else "flow branch on", np.median(a), np.quantile(a, 0.9), a.max(), np.median(b), np.quantile(b, 0.9), b.max()))
    assert np.median(a) <= 3 * np.median(b) and np.quantile(a, 0.9) <= 4 * np.quantile(b, 0.9) and a.max() <= 5 * b.max()
    if not no_flow:
        assert any(k.startswith("model_res_flow") for k in eh) and any(k.startswith("model_final_flow") for k in eh)
