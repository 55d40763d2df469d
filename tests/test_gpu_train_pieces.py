"""GPU parity of the train-step pieces built so far (forward only):
PatchGAN / multiscale discriminator, LSGAN + feature-matching losses, fused Adam -- vs the CPU
oracle modules (oracle/generator_ref.py) and torch.optim.Adam."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _nhwc_batch(x, cs):
    from text2video_amd import ops
    return torch.stack([ops.nchw_to_nhwc(x[b].to("cuda:0").contiguous(), cs) for b in range(x.shape[0])])


def _init(net, seed):
    g = torch.Generator().manual_seed(seed)
    from oracle.generator_ref import weights_init
    net.apply(lambda m: weights_init(m, g))
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.bias.data.normal_(0, 0.1, generator=g)
    return net.train()


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_multiscale_discriminator_forward_and_losses(norm):
    from oracle.generator_ref import MultiscaleDiscriminator
    from text2video_amd.discriminator import HipMultiscaleDiscriminator, feature_matching_loss, gan_loss
    B, H, W = 2, 64, 96
    ref = _init(MultiscaleDiscriminator(6, 64, 3, 2, norm), 5)
    hip = HipMultiscaleDiscriminator(6, 64, 3, 2, norm, "cuda:0").load_state_dict(ref.state_dict())
    fake, real = _rand(B, 6, H, W, seed=1), _rand(B, 6, H, W, seed=2)
    with torch.no_grad():
        want_f, want_r = ref(fake), ref(real)
    got_f, got_r = hip.forward(_nhwc_batch(fake, 8)), hip.forward(_nhwc_batch(real, 8))
    for i in range(2):
        for j in range(5):
            w = want_f[i][j]
            g = got_f[i][j][..., :w.shape[1]].permute(0, 3, 1, 2).cpu()
            assert g.shape == w.shape, (i, j, g.shape, w.shape)
            assert (g - w).abs().max().item() <= 2e-4 * max(1.0, w.abs().max().item()), (i, j)
    mse, l1 = torch.nn.MSELoss(), torch.nn.L1Loss()
    want_gan = sum(mse(want_f[i][-1], torch.ones_like(want_f[i][-1])) for i in range(2))
    assert abs(gan_loss(got_f, True).item() - want_gan.item()) <= 1e-4 * max(1.0, abs(want_gan.item()))
    want_gan0 = sum(mse(want_r[i][-1], torch.zeros_like(want_r[i][-1])) for i in range(2))
    assert abs(gan_loss(got_r, False).item() - want_gan0.item()) <= 1e-4 * max(1.0, abs(want_gan0.item()))
    want_fm = sum(0.5 * 1.0 * l1(want_f[i][j], want_r[i][j]) * 10.0 for i in range(2) for j in range(4))
    assert abs(feature_matching_loss(got_f, got_r).item() - want_fm.item()) <= 1e-4 * max(1.0, abs(want_fm.item()))


def test_face_discriminator_128_crop():
    """--add_face_disc: single-scale PatchGAN on a 128x128 crop."""
    from oracle.generator_ref import MultiscaleDiscriminator
    from text2video_amd.discriminator import HipMultiscaleDiscriminator
    ref = _init(MultiscaleDiscriminator(6, 64, 3, 1, "batch"), 6)
    hip = HipMultiscaleDiscriminator(6, 64, 3, 1, "batch", "cuda:0").load_state_dict(ref.state_dict())
    x = _rand(1, 6, 128, 128, seed=3)
    with torch.no_grad():
        want = ref(x)
    got = hip.forward(_nhwc_batch(x, 8))
    assert got[0][-1].shape[1:3] == (19, 19)
    for j in range(5):
        w = want[0][j]
        assert (got[0][j][..., :w.shape[1]].permute(0, 3, 1, 2).cpu() - w).abs().max().item() <= 2e-4 * max(1.0, w.abs().max().item())


from oracle.optim_ref import adam_041_step as _adam_041   # pinned to the reference's torch-0.4.1 adam.py (tests/golden/adam041.npz)


def test_fused_adam_matches_torch041_semantics():
    from text2video_amd import ops
    n = 100003
    p0, g = _rand(n, seed=7), _rand(n, seed=8, scale=0.1)
    pr, mr, vr = p0.clone().double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda:0"), torch.zeros(n, device="cuda:0")
    for step in range(1, 5):
        gs = g * step
        _adam_041(pr, gs.double(), mr, vr, 2e-4, 0.5, 0.999, 1e-8, step)   # vid2vid: lr 2e-4, beta1 0.5
        ops.adam_step(p, gs.cuda(), m, v, 2e-4, 0.5, 0.999, 1e-8, step)
        assert (p.cpu().double() - pr).abs().max().item() <= 6e-7   # fp32 ulp of |p| < 8 is 4.8e-7
        assert (m.cpu().double() - mr).abs().max().item() <= 1e-7
    # the golden steps of the reference's own optimiser file (gradients from 1e-6 to 30, some exact zeros)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adam041.npz"))
    p = torch.from_numpy(gold["p0"].copy()).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(gold["grads"].shape[0]):
        ops.adam_step(p, torch.from_numpy(gold["grads"][s]).cuda(), m, v, float(gold["lr"]), float(gold["beta1"]),
                      float(gold["beta2"]), float(gold["eps"]), s + 1)
        assert np.abs(p.cpu().numpy() - gold["p_after"][s]).max() <= 3e-8    # |p| < 0.25: a few fp32 ulps
    assert np.abs(m.cpu().numpy() - gold["exp_avg"]).max() <= 1e-6 * np.abs(gold["exp_avg"]).max()
    # second moments too: 1 - beta2 is formed in double and rounded once, as adam.py does (0.999f would give 9.9999e-4)
    assert np.abs(v.cpu().numpy() - gold["exp_avg_sq"]).max() <= 1e-6 * np.abs(gold["exp_avg_sq"]).max()


def test_fused_adam_counts_steps_per_parameter():
    """adam.py:58-60,82: state['step'] belongs to the parameter and only advances when it has a gradient."""
    from text2video_amd import train as T
    a = torch.nn.Parameter(torch.ones(8, device="cuda:0"))
    b = torch.nn.Parameter(torch.ones(8, device="cuda:0"))
    opt = T.FusedAdam([a, b], lr=1e-2)
    g = torch.full((8,), 0.5, device="cuda:0")
    a.grad, b.grad = g.clone(), None
    opt.step()
    a.grad, b.grad = g.clone(), g.clone()
    opt.step()
    assert opt.steps == [2, 1]
    # b's first update is bias-corrected as step 1: exactly one lr-sized move (m/sqrt(v) = 1 after correction)
    assert abs((1.0 - b.detach().cpu()[0].item()) - 1e-2) <= 1e-6


def test_reductions_are_deterministic_and_accurate():
    from text2video_amd import ops
    a, b = _rand(3, 50, 70, seed=9), _rand(3, 50, 70, seed=10)
    want = (a.double() - b.double()).abs().sum().item()
    got = [ops.sum_abs_diff(a.cuda(), b.cuda()).item() for _ in range(3)]
    assert got[0] == got[1] == got[2] and abs(got[0] - want) <= 1e-5 * want
    want2 = ((a.double() - 1.0) ** 2).sum().item()
    assert abs(ops.sum_sq_diff_const(a.cuda(), 1.0).item() - want2) <= 1e-5 * want2


def test_temporal_discriminator_input_and_gradients():
    """netD_T (13 channels = 3 frames x RGB + 2 zero flows): forward features, input gradient and parameter gradients
    of the HIP discriminator against torch autograd on the oracle module."""
    from oracle.generator_ref import MultiscaleDiscriminator, weights_init
    from text2video_amd import train as T
    H, W, B = 48, 32, 2
    ref = MultiscaleDiscriminator(13, 16, 3, 2, "batch").train()
    gen = torch.Generator().manual_seed(3)
    ref.apply(lambda m: weights_init(m, gen))
    sd = {k: v.clone() for k, v in ref.state_dict().items() if "running" not in k and "num_batches" not in k}
    hip = T.TrainableDiscriminator(13, sd, 16, 3, 2, "batch", "cuda:0")
    x = torch.randn(B, 13, H, W, generator=gen)
    x[:, 9:] = 0                                   # the flow channels are zero
    xr = x.clone().requires_grad_(True)
    pr = ref(xr)
    loss_r = sum((p[-1] - 1).pow(2).mean() for p in pr) + sum(f.abs().mean() for p in pr for f in p[:-1])
    gx_ref, = torch.autograd.grad(loss_r, [xr], retain_graph=True)
    gp_ref = torch.autograd.grad(loss_r, list(ref.parameters()))
    xh = torch.zeros(B, H, W, 16, device="cuda:0")
    xh[..., :13] = x.permute(0, 2, 3, 1).cuda()
    xh.requires_grad_(True)
    ph = hip(xh)
    loss_h = sum((p[-1][..., :1] - 1).pow(2).mean() for p in ph) + \
        sum(f[..., :r.shape[1]].abs().mean() for p, q in zip(ph, pr) for f, r in zip(p[:-1], q[:-1]))
    assert abs(loss_h.item() - loss_r.item()) <= 1e-4 * abs(loss_r.item())
    gx, = torch.autograd.grad(loss_h, [xh], retain_graph=True)
    assert (gx[..., :13].permute(0, 3, 1, 2).cpu() - gx_ref).abs().max().item() <= 2e-4 * gx_ref.abs().max().item()
    gp = torch.autograd.grad(loss_h, list(hip.parameters()), allow_unused=True)
    got = {k: g for (k, _), g in zip(hip.named_upstream_parameters().items(), gp)}
    for (k, _), r in zip(ref.named_parameters(), gp_ref):
        if r.abs().max().item() > 1e-5:
            assert (got[k].cpu() - r).abs().max().item() <= 2e-3 * r.abs().max().item(), k


def test_trainer_uses_temporal_discriminators_across_chunks(tmp_path):
    """Vid2VidTrainer: windows (t-2d, t-d, t), d = 3^s, built from the sequence history across chunks of 2 frames."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "x", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--no_first_img",
                                "--ngf", "16", "--n_blocks", "2", "--num_D", "1", "--fineSize", "64", "--n_scales_temporal", "2",
                                "--max_frames_per_gpu", "2", "--checkpoints_dir", str(tmp_path), "--synthetic_data"])
    tr = T.Vid2VidTrainer(opt, "cuda:0")
    assert len(tr.DT) == 2 and tr.DT[0].input_nc == 13
    g = torch.Generator().manual_seed(0)
    w0 = [p.detach().clone() for p in tr.DT[0].parameters()]
    w1 = [p.detach().clone() for p in tr.DT[1].parameters()]
    prev, seen = None, []
    for chunk in range(4):                                   # frames 0..7 of one sequence
        pose = torch.zeros(2, 64, 64, 12, device="cuda:0")
        pose[..., :9] = torch.rand(2, 64, 64, 9, generator=g).cuda() * 2 - 1
        real = torch.zeros(2, 64, 64, 4, device="cuda:0")
        real[..., :3] = torch.tanh(torch.randn(2, 64, 64, 3, generator=g)).cuda()
        losses, prev = tr.train_step(pose, real, None, prev)
        seen.append(sorted(k for k in losses if k.startswith("D_T")))
        assert all(np.isfinite(v) for v in losses.values())
    # scale 0 needs 3 consecutive frames (first window ends on frame 2), scale 1 frames 0,3,6 (ends on frame 6)
    assert seen == [[], ["D_T0"], ["D_T0"], ["D_T0", "D_T1"]]
    assert any(not torch.equal(a, b) for a, b in zip(w0, tr.DT[0].parameters()))
    assert any(not torch.equal(a, b) for a, b in zip(w1, tr.DT[1].parameters()))
    _, prev = tr.train_step(pose, real, None, None)          # prev=None: a new sequence, history cleared
    assert len(tr._hist_real) == 2
    tr.save("latest")
    assert sorted(os.listdir(tmp_path / "x")) == ["latest_net_D.pth", "latest_net_D_T0.pth", "latest_net_D_T1.pth", "latest_net_G0.pth"]


def test_trainer_batches_the_frames_winograd_weight_gradients(tmp_path, monkeypatch, t2v_env):
    """The two frames of a chunk run through every ResnetBlock conv; their Winograd-domain weight gradients are
    reduced together by the backward node that runs last (one K = 2 x tiles reduction).  Same gradients as the
    frame-by-frame reductions (T2V_WGRAD_BATCH=0) up to fp32 summation order."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    args = ["--name", "x", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--no_first_img", "--ngf", "32",
            "--n_blocks", "2", "--n_downsample_G", "2", "--num_D", "1", "--fineSize", "128", "--max_frames_per_gpu", "2",
            "--checkpoints_dir", str(tmp_path), "--synthetic_data", "--no_flow"]    # 32x32 bottleneck: F(4x4,3x3) territory
    g = torch.Generator().manual_seed(0)
    S = 128
    pose = torch.zeros(2, S, S, 12, device="cuda:0")
    pose[..., :9] = torch.rand(2, S, S, 9, generator=g).cuda() * 2 - 1
    real = torch.zeros(2, S, S, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.randn(2, S, S, 3, generator=g)).cuda()
    prev = torch.zeros(1, S, S, 8, device="cuda:0")
    prev[..., :6] = torch.tanh(torch.randn(1, S, S, 6, generator=g)).cuda()
    calls, grads = [], {}
    orig = T._batched_winograd_wgrad
    monkeypatch.setattr(T, "_batched_winograd_wgrad", lambda w, x, dc, d, *rest: (calls.append(x.shape[0]), orig(w, x, dc, d, *rest))[1])
    for mode in ("1", "0"):
        t2v_env("T2V_WGRAD_BATCH", mode)
        tr = T.Vid2VidTrainer(TrainOptions().parse(args), "cuda:0")
        n0 = len(calls)
        tr.train_step(pose, real, None, prev.clone())
        grads[mode] = [p.grad.clone() for p in tr.optG.params]
        if mode == "1":
            # n_blocks 2 -> one ResnetBlock in each encoder tail + one in the trunk: 3 blocks x 2 convs, 2 frames
            assert len(calls) - n0 == 2 * 6, "every ResnetBlock conv of both frames takes the batched path"
            assert all(getattr(p, "_t2v_wg_state", None) is None for p in tr.optG.params), "every reduction was flushed"
        else:
            assert len(calls) == n0
    worst = 0.0
    for a, b in zip(grads["1"], grads["0"]):
        worst = max(worst, (a - b).abs().max().item() / max(1e-12, b.abs().max().item()))
    print("batched vs frame-by-frame weight gradients: worst relative difference %.1e" % worst)
    assert worst <= 1e-4


def test_maxpool2x2_forward_backward_matches_torch():
    from text2video_amd import ops
    g = torch.Generator().manual_seed(4)
    for (B, H, W, C) in [(1, 8, 8, 4), (2, 14, 10, 64), (1, 7, 9, 8)]:     # odd sizes: floor mode drops the last row / column
        if B > 1 and H % 2:
            continue
        x = torch.randn(B, H, W, C, generator=g)
        x[0, :2, :2, 0] = 0.5                                              # a tie: gradient to the first maximum
        xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        yr = torch.nn.functional.max_pool2d(xr, 2, 2)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xd = x.cuda().contiguous()
        y = ops.maxpool2x2(xd if B > 1 else xd[0])
        dx = ops.maxpool2x2_backward(xd if B > 1 else xd[0], dy.permute(0, 2, 3, 1).contiguous().cuda() if B > 1
                                     else dy.permute(0, 2, 3, 1)[0].contiguous().cuda())
        y, dx = (y, dx) if B > 1 else (y[None], dx[None])
        assert torch.equal(y.cpu(), yr.detach().permute(0, 2, 3, 1))
        assert torch.equal(dx.cpu(), xr.grad.permute(0, 2, 3, 1))


def test_vgg_perceptual_loss_and_its_gradient_match_the_oracle():
    """VGGLoss = sum_i w_i L1(vgg19 relu_i_1(fake), vgg19 relu_i_1(real)); seeded random weights in
    torchvision's key names (the pretrained file is not in the tree).  Value and d/d fake against torch autograd."""
    from oracle.generator_ref import VGG19Features, vgg_loss_ref
    from text2video_amd import train as T
    sd = T.vgg19_random_state_dict(3)
    ref = VGG19Features()
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    fake = torch.tanh(torch.randn(2, 3, 64, 64, generator=g)).requires_grad_(True)
    real = torch.tanh(torch.randn(2, 3, 64, 64, generator=g))
    lr = vgg_loss_ref(ref, fake, real)
    lr.backward()
    hip = T.HipVGG19Features(sd, "cuda:0")

    def nhwc4(t):
        out = torch.zeros(t.shape[0], t.shape[2], t.shape[3], 4, device="cuda:0")
        out[..., :3] = t.detach().permute(0, 2, 3, 1).cuda()
        return out

    hf = nhwc4(fake).requires_grad_(True)
    lh = T.vgg_loss(hip, hf, nhwc4(real))
    (gh,) = torch.autograd.grad(lh, [hf])
    assert abs(lh.item() - lr.item()) <= 1e-5 * max(1.0, abs(lr.item()))
    gr = fake.grad.permute(0, 2, 3, 1)
    d = (gh[..., :3].cpu() - gr).abs()
    err = d.max().item() / gr.abs().max().item()
    l2 = (d.pow(2).sum() / gr.pow(2).sum()).sqrt().item()
    nbad = int((d > 1e-3 * gr.abs().max()).sum())
    print("VGG loss %.6f vs %.6f, input gradient: max error %.1e of the scale, relative L2 %.1e, %d of %d elements off by "
          "> 1e-3 of the scale" % (lh.item(), lr.item(), err, l2, nbad, d.numel()))
    # 13 ReLUs and 4 max-pools deep, a handful of the ~1.5 M activations sit within fp32 rounding of a kink (ReLU gate,
    # pool argmax) and the two implementations take different branches there: a few receptive fields of the input
    # gradient differ visibly, everything else agrees to rounding -- the CPU oracle evaluated in fp64 differs from its
    # own fp32 run by the same statistics (74 elements, relative L2 2.3e-3).  (The layers' backward kernels are each
    # checked exactly in test_gpu_backward.py / the max-pool test above.)
    med = d.median().item() / gr.abs().max().item()
    assert med <= 1e-5 and l2 <= 1e-2 and nbad <= 0.01 * d.numel(), (med, l2, nbad)
    assert gh[..., 3].abs().max().item() == 0.0
    # the taps themselves
    with torch.no_grad():
        for a, b in zip(hip(nhwc4(real)), ref(real)):
            assert (a.permute(0, 3, 1, 2).cpu() - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


def test_trainer_adds_the_vgg_term(tmp_path):
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    args = ["--name", "x", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--no_first_img", "--ngf", "16",
            "--n_blocks", "2", "--num_D", "1", "--fineSize", "64", "--max_frames_per_gpu", "2", "--n_scales_temporal", "0",
            "--checkpoints_dir", str(tmp_path), "--synthetic_data", "--vgg_random_init"]
    tr = T.Vid2VidTrainer(TrainOptions().parse(args), "cuda:0")
    assert tr.vgg is not None and not any(p.requires_grad for w in tr.vgg.w.values() for p in w)
    g = torch.Generator().manual_seed(0)
    pose = torch.zeros(2, 64, 64, 12, device="cuda:0")
    pose[..., :9] = torch.rand(2, 64, 64, 9, generator=g).cuda() * 2 - 1
    real = torch.zeros(2, 64, 64, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.randn(2, 64, 64, 3, generator=g)).cuda()
    losses, _ = tr.train_step(pose, real, None, None)
    assert "G_VGG" in losses and np.isfinite(losses["G_VGG"]) and losses["G_VGG"] > 0
    tr2 = T.Vid2VidTrainer(TrainOptions().parse(args[:-1]), "cuda:0")       # no weights given: the term is off (with a notice)
    assert tr2.vgg is None


def test_multi_tensor_adam_equals_the_per_tensor_kernel(t2v_env):
    """FusedAdam.step as ONE launch over all parameter tensors (chunk table on the device) against one launch per tensor:
    identical bits -- tensors longer than a chunk, shorter than a wave, and one that has no gradient in some steps."""
    from text2video_amd import train as T
    sizes = [5, 70001, 3 * (1 << 16), 64, (1 << 16) + 1]
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(n, generator=g) for n in sizes]
    grads = [[torch.randn(n, generator=g) * 0.1 for n in sizes] for _ in range(3)]
    outs = {}
    for mode in ("1", "0"):
        t2v_env("T2V_ADAM_MULTI", mode)
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        opt = T.FusedAdam(ps, lr=1e-2)
        for s in range(3):
            for i, p in enumerate(ps):
                p.grad = None if (i == 3 and s != 1) else grads[s][i].cuda()
            opt.step()
        torch.cuda.synchronize()
        outs[mode] = ([p.detach().cpu() for p in ps], [m.cpu() for m in opt.m], [v.cpu() for v in opt.v], list(opt.steps))
    assert outs["1"][3] == outs["0"][3] == [3, 3, 3, 1, 3]
    for k in range(3):
        for a, b in zip(outs["1"][k], outs["0"][k]):
            assert torch.equal(a, b)
    assert not torch.equal(outs["1"][0][0], init[0])


def test_batchnorm_running_statistics_follow_torch_and_reach_the_checkpoint(tmp_path):
    """Training-mode BatchNorm2d also moves running_mean / running_var (momentum 0.1, UNBIASED variance;
    /root/reference/venv_vid2vid/lib/python3.7/site-packages/torch/nn/modules/batchnorm.py:57-64).  The discriminator
    (statistics over the batch) and the generator (a batch of one per call) against torch.nn.BatchNorm2d inside the CPU
    oracle after two forwards each; then the saved checkpoint carries them and loads strictly into the oracle's modules."""
    from oracle.generator_ref import CompositeGenerator, MultiscaleDiscriminator
    from text2video_amd import train as T
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    from text2video_amd.options import TrainOptions
    # discriminator: batch statistics over B = 2 images
    ref = _init(MultiscaleDiscriminator(6, 16, 3, 2, "batch"), 5)
    dsd = {k: v.clone() for k, v in ref.state_dict().items() if "running" not in k and "num_batches" not in k}
    Dh = T.TrainableDiscriminator(6, dsd, 16, 3, 2, "batch", "cuda:0")
    for seed in (1, 2):
        x = _rand(2, 6, 64, 96, seed=seed)
        ref(x)
        Dh(torch.cat([_nhwc_batch(x, 8)], 0))
    checked = 0
    for k, p in Dh.named_upstream_parameters().items():
        if k.endswith(".weight") and p.dim() == 1:
            rs = T.running_stats(p, create=False)
            assert rs is not None and rs[2] == 2, k
            base = k[:-len("weight")]
            want_m, want_v = ref.state_dict()[base + "running_mean"], ref.state_dict()[base + "running_var"]
            assert (rs[0].cpu() - want_m).abs().max().item() <= 1e-5 + 1e-4 * want_m.abs().max().item(), k
            assert ((rs[1].cpu() - want_v).abs() / want_v.abs()).max().item() <= 1e-4, k
            assert int(ref.state_dict()[base + "num_batches_tracked"]) == rs[2]
            checked += 1
    assert checked >= 4
    # generator: BatchNorm2d(train) on a batch of ONE per frame; through the trainer, then its checkpoint
    opt = TrainOptions().parse(["--name", "bn", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--no_flow",
                                "--ngf", "16", "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "1", "--ndf", "16",
                                "--no_vgg", "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img",
                                "--checkpoints_dir", str(tmp_path)])
    tr = T.Vid2VidTrainer(opt, "cuda:0", seed=3)
    spec = tr.spec
    Gr = CompositeGenerator(9, 3, 6, 16, 2, 2, True, "batch").train()
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.G.named_upstream_parameters().items()}
    Gr.load_state_dict(sd0, strict=False)
    poses = _rand(2, 9, 64, 64, seed=4).clamp(-1, 1)
    real = torch.tanh(_rand(2, 3, 64, 64, seed=5))
    # the oracle's two frames with the SAME previous frames the trainer builds (zero FIFO, then its own detached output)
    with torch.no_grad():
        prev0 = torch.zeros(1, 6, 64, 64)
        o1 = Gr(poses[0:1], prev0, True)
        Gr(poses[1:2], torch.cat([prev0[:, 3:], o1[0]], 1), False)
    pz = torch.zeros(2, 64, 64, 12, device="cuda:0")
    pz[..., :9] = poses.permute(0, 2, 3, 1).cuda()
    rz = torch.zeros(2, 64, 64, 4, device="cuda:0")
    rz[..., :3] = real.permute(0, 2, 3, 1).cuda()
    tr.train_step(pz, rz, None, None)
    tr.save("latest", (1, 1))
    ck = torch.load(str(tmp_path / "bn" / "latest_net_G0.pth"), map_location="cpu")
    want = Gr.state_dict()
    n = 0
    for k in want:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert k in ck, k
            tol = 1e-5 + 2e-4 * want[k].abs().max().item()
            assert (ck[k] - want[k]).abs().max().item() <= tol, (k, (ck[k] - want[k]).abs().max().item())
            n += 1
        elif k.endswith("num_batches_tracked"):
            assert int(ck[k]) == int(want[k]) == 2, k
    assert n >= 10
    assert not Gr.load_state_dict(ck, strict=True).missing_keys       # the file is a complete torch state dict
