"""GPU parity of one full train iteration (generator + multiscale discriminator, LSGAN + feature
matching, Adam) on the HIP path against torch autograd over the CPU oracle modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _nhwc(x_bchw, cs):
    """[B,C,H,W] cpu -> [B,H,W,cs] device"""
    B, C, H, W = x_bchw.shape
    out = torch.zeros(B, H, W, cs, device="cuda:0")
    out[..., :C] = x_bchw.permute(0, 2, 3, 1).cuda()
    return out


from oracle.optim_ref import adam_041_step as _adam_041   # pinned to the reference's torch-0.4.1 adam.py


@pytest.mark.parametrize("size", [32, 64], ids=["32_direct", "64_winograd"])
def test_one_train_iteration_matches_autograd_oracle(size):
    from oracle.generator_ref import CompositeGenerator, MultiscaleDiscriminator, weights_init
    from text2video_amd import train as T
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    H = W = size
    # the discriminator's conv biases keep torch's default init (weights_init only draws the weights): seed it.
    # Unseeded, about a third of the draws put some pre-activation of these tiny maps (8x8 bottleneck) within
    # rounding distance of its ReLU / LeakyReLU kink, where the two fp32 implementations take different gates
    # and single weight gradients move by 1e-2 of their scale -- a property of the comparison, not of either side
    torch.manual_seed(0)
    if size == 64:   # 16x16 bottleneck: the ResnetBlock convs and their data gradients take the Winograd path
        from text2video_amd import ops
        assert ops.best_conv_algo(ops.conv_desc(16, 16, 128, 128, 3, 1, 1, ops.PAD_REFLECT), 128) != ops.ALGO_DIRECT
        assert ops.best_conv_algo(ops.conv_desc(16, 16, 128, 128, 3, 1, 2, ops.PAD_ZERO), 128) != ops.ALGO_DIRECT
    spec = GeneratorSpec(ngf=32, n_downsample=2, n_blocks=2, no_flow=True, norm="batch")
    sd = synthetic_state_dict(spec, 3, "vid2vid")
    # ---------------- oracle (CPU, torch autograd)
    Gr = CompositeGenerator(9, 3, 6, 32, 2, 2, True, "batch").train()
    Gr.load_state_dict(sd, strict=False)
    Dr = MultiscaleDiscriminator(6, 16, 3, 2, "batch").train()
    gen = torch.Generator().manual_seed(7)
    Dr.apply(lambda m: weights_init(m, gen))
    dsd = {k: v.clone() for k, v in Dr.state_dict().items() if "running" not in k and "num_batches" not in k}
    poses = _rand(2, 9, H, W, seed=1).clamp(-1, 1)          # two frames (two sliding windows)
    real = torch.tanh(_rand(2, 3, H, W, seed=2))
    mse, l1 = torch.nn.MSELoss(), torch.nn.L1Loss()

    def gan(pred, real_target):
        return sum(mse(p[-1], torch.ones_like(p[-1]) if real_target else torch.zeros_like(p[-1])) for p in pred)

    def fm(pf, pr):
        return sum(0.5 * 1.0 * l1(pf[i][j], pr[i][j].detach()) * 10.0 for i in range(2) for j in range(4))

    # non-zero previous frames: an all-zero prev image makes the image encoder's norm layers 0/0
    # (rstd = 1/sqrt(eps) = 316 per layer), which turns its gradients into amplified rounding noise in
    # ANY fp32 implementation -- not a meaningful parity target
    prev0 = torch.tanh(_rand(1, 6, H, W, seed=5))
    f1 = Gr(poses[0:1], prev0, True)[0]
    f2 = Gr(poses[1:2], torch.cat([prev0[:, 3:], f1.detach()], 1), True)[0]
    fake = torch.cat([f1, f2], 0)
    A = poses[:, 6:9]                                          # newest pose map of each window
    pred_real = Dr(torch.cat([A, real], 1))
    pred_fake_d = Dr(torch.cat([A, fake.detach()], 1))
    loss_D = 0.5 * (gan(pred_fake_d, False) + gan(pred_real, True))
    pred_fake_g = Dr(torch.cat([A, fake], 1))
    loss_G = gan(pred_fake_g, True) + fm(pred_fake_g, pred_real)
    gG = torch.autograd.grad(loss_G, list(Gr.parameters()), retain_graph=True)
    gD = torch.autograd.grad(loss_D, list(Dr.parameters()))
    ref_gG = {k: g for (k, _), g in zip(Gr.named_parameters(), gG)}
    ref_gD = {k: g for (k, _), g in zip(Dr.named_parameters(), gD)}

    # ---------------- HIP path
    Gh = T.TrainableGenerator(spec, sd, "cuda:0")
    Dh = T.TrainableDiscriminator(6, dsd, 16, 3, 2, "batch", "cuda:0")
    pz = _nhwc(poses, 12)
    hprev0 = _nhwc(prev0, 8)
    h1 = Gh(pz[0:1], hprev0)
    prev2 = torch.zeros(1, H, W, 8, device="cuda:0")
    prev2[..., 0:3] = hprev0[..., 3:6]
    prev2[..., 3:6] = h1.detach()[..., :3]
    h2 = Gh(pz[1:2], prev2)
    hfake = torch.cat([h1, h2], 0)
    assert (hfake[..., :3].permute(0, 3, 1, 2).cpu() - fake.detach()).abs().max().item() <= 1e-4
    A8 = _nhwc(A, 3)
    z2 = torch.zeros(2, H, W, 2, device="cuda:0")

    def d_in(img4):
        return torch.cat([A8, img4[..., :3], z2], -1).contiguous()

    hp_real = Dh(d_in(_nhwc(real, 4)))
    hp_fake_d = Dh(d_in(hfake.detach()))
    hloss_D = 0.5 * (T.gan_loss(hp_fake_d, False) + T.gan_loss(hp_real, True))
    hp_fake_g = Dh(d_in(hfake))
    hloss_G = T.gan_loss(hp_fake_g, True) + T.feature_matching_loss(hp_fake_g, hp_real)
    assert abs(hloss_D.item() - loss_D.item()) <= 1e-4 * max(1, abs(loss_D.item()))
    assert abs(hloss_G.item() - loss_G.item()) <= 1e-4 * max(1, abs(loss_G.item()))
    hgG = torch.autograd.grad(hloss_G, list(Gh.parameters()), retain_graph=True, allow_unused=True)
    hgD = torch.autograd.grad(hloss_D, list(Dh.parameters()), allow_unused=True)
    got_gG = {k: g for (k, _), g in zip(Gh.named_upstream_parameters().items(), hgG)}
    got_gD = {k: g for (k, _), g in zip(Dh.named_upstream_parameters().items(), hgD)}

    def check(got, ref, tag):
        errs = {}
        for k, r in ref.items():
            g = got[k]
            assert g is not None, (tag, k)
            if r.abs().max().item() <= 1e-5:
                continue   # conv biases in front of a norm: mathematically zero gradient, only rounding noise
            errs[k] = (g.cpu() - r).abs().max().item() / r.abs().max().item()
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
        print(tag, "largest relative gradient errors:", ["%s %.1e" % kv for kv in top],
              "median %.1e" % float(np.median(list(errs.values()))))
        # Winograd F(2x2,3x3) in the forward AND the data gradient of every ResnetBlock conv roughly doubles the
        # rounding noise of the direct kernels (observed median 0.6e-4 .. 1.4e-4 against the fp32 CPU oracle)
        assert max(errs.values()) <= 2e-3 and float(np.median(list(errs.values()))) <= (1e-4 if size == 32 else 4e-4)
        return max(errs.values())

    wG = check(got_gG, ref_gG, "G")
    wD = check(got_gD, ref_gD, "D")
    print("worst relative gradient error  G %.2e  D %.2e" % (wG, wD))

    # ---------------- one Adam step on both sides (torch-0.4.1 update rule)
    optG = T.FusedAdam(Gh.parameters())
    for p, g in zip(Gh.parameters(), hgG):
        p.grad = g
    optG.step()
    for (k, p), g in zip(Gr.named_parameters(), gG):
        pr = p.detach().clone().double()
        _adam_041(pr, g.double(), torch.zeros_like(pr), torch.zeros_like(pr), 2e-4, 0.5, 0.999, 1e-8, 1)
        got = Gh.named_upstream_parameters()[k].detach().cpu().double()
        if g.abs().max().item() > 1e-5:
            assert (got - pr).abs().max().item() <= 2.5e-4, k    # first Adam step moves every weight by ~lr


@pytest.mark.parametrize("size", [32, 64], ids=["32_direct", "64_winograd"])
def test_flow_branch_train_iteration_matches_autograd_oracle(size):
    """The generator WITH its flow branch: frame 1 raw-only (--no_first_img), frame 2 through the flow-warp
    compositor; LSGAN + feature matching on the blended AND the raw frames, the flow / warp / weight losses against
    a zero reference flow.  Every parameter gradient -- model_res_flow, model_up_flow, model_final_flow and
    model_final_w included -- against torch autograd on oracle.CompositeGenerator (whose `resample` restates
    torch 0.4.1's grid sampler), i.e. the adjoint of raw*w + warp*(1-w) into raw, weight and flow."""
    from oracle.generator_ref import CompositeGenerator, MultiscaleDiscriminator, resample, weights_init
    from text2video_amd import train as T
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    H = W = size
    torch.manual_seed(0)
    spec = GeneratorSpec(ngf=32, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    # flow_gain 0.1: flows of a few pixels, as a trained network predicts (a random flow head x20 throws every
    # sample +-40 px away, where 1e-6 differences in the flow pick other taps)
    sd = synthetic_state_dict(spec, 3, "vid2vid", flow_gain=0.1)
    Gr = CompositeGenerator(9, 3, 6, 32, 2, 2, False, "batch").train()
    missing = Gr.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("running_" in k or "num_batches" in k for k in missing.missing_keys)
    Dr = MultiscaleDiscriminator(6, 16, 3, 2, "batch").train()
    gen = torch.Generator().manual_seed(7)
    Dr.apply(lambda m: weights_init(m, gen))
    dsd = {k: v.clone() for k, v in Dr.state_dict().items() if "running" not in k and "num_batches" not in k}
    poses = _rand(2, 9, H, W, seed=1).clamp(-1, 1)
    real = torch.tanh(_rand(2, 3, H, W, seed=2))
    real_prev = torch.cat([torch.tanh(_rand(1, 3, H, W, seed=4)), real[:1]], 0)
    # a confidence mask with both values (the rule ||real - real_prev|| < 0.02 is all-zero on random frames)
    conf = (torch.from_numpy(np.random.default_rng(9).random((2, 1, H, W))) < 0.7).float()
    mse, l1 = torch.nn.MSELoss(), torch.nn.L1Loss()
    lam_F = lam_T = 10.0

    def gan(pred, real_target):
        return sum(mse(p[-1], torch.ones_like(p[-1]) if real_target else torch.zeros_like(p[-1])) for p in pred)

    def fm(pf, pr):
        return sum(0.5 * 1.0 * l1(pf[i][j], pr[i][j].detach()) * 10.0 for i in range(2) for j in range(4))

    def ml1(a, b, m):
        m = m.expand(-1, a.shape[1], -1, -1)
        return l1(a * m, b * m)

    prev0 = torch.tanh(_rand(1, 6, H, W, seed=5))
    o1 = Gr(poses[0:1], prev0, True)
    prev1 = torch.cat([prev0[:, 3:], o1[0].detach()], 1)
    o2 = Gr(poses[1:2], prev1, False)
    fake = torch.cat([o1[0], o2[0]], 0)
    flow = torch.cat([o1[1], o2[1]], 0)
    weight = torch.cat([o1[2], o2[2]], 0)
    raw = torch.cat([o1[3], o2[3]], 0)
    assert (flow.abs().max().item() < 8.0) and (flow.abs().mean().item() > 0.05)
    A = poses[:, 6:9]
    pred_real = Dr(torch.cat([A, real], 1))
    pfg, pfg_r = Dr(torch.cat([A, fake], 1)), Dr(torch.cat([A, raw], 1))
    loss_G = gan(pfg, True) + fm(pfg, pred_real) + gan(pfg_r, True) + fm(pfg_r, pred_real)
    fake_prev = torch.cat([prev0[:, 3:], prev1[:, 3:]], 0)
    zero_flow = torch.zeros_like(flow)
    loss_G = loss_G + ml1(flow, zero_flow, conf) * lam_F + ml1(resample(real_prev, flow), real, conf) * lam_T \
        + ml1(weight, torch.zeros_like(weight), conf) + ml1(fake, resample(fake_prev, zero_flow).detach(), conf) * lam_T
    gG = torch.autograd.grad(loss_G, list(Gr.parameters()))
    ref_gG = {k: g for (k, _), g in zip(Gr.named_parameters(), gG)}

    Gh = T.TrainableGenerator(spec, sd, "cuda:0")
    Dh = T.TrainableDiscriminator(6, dsd, 16, 3, 2, "batch", "cuda:0")
    pz = _nhwc(poses, 12)
    hprev0 = _nhwc(prev0, 8)
    h1, r1, fw1 = Gh(pz[0:1], hprev0, use_raw_only=True, full=True)
    hprev1 = torch.zeros(1, H, W, 8, device="cuda:0")
    hprev1[..., 0:3] = hprev0[..., 3:6]
    hprev1[..., 3:6] = h1.detach()[..., :3]
    h2, r2, fw2 = Gh(pz[1:2], hprev1, use_raw_only=False, full=True)
    hfake, hraw, hfw = torch.cat([h1, h2], 0), torch.cat([r1, r2], 0), torch.cat([fw1, fw2], 0)
    assert (hfake[..., :3].permute(0, 3, 1, 2).cpu() - fake.detach()).abs().max().item() <= 2e-4
    assert (hfw[..., :2].permute(0, 3, 1, 2).cpu() - flow.detach()).abs().max().item() <= 2e-4
    assert (hfw[..., 2:3].permute(0, 3, 1, 2).cpu() - weight.detach()).abs().max().item() <= 1e-4
    A8 = _nhwc(A, 3)
    z2 = torch.zeros(2, H, W, 2, device="cuda:0")

    def d_in(img4):
        return torch.cat([A8, img4[..., :3], z2], -1).contiguous()

    hp_real = Dh(d_in(_nhwc(real, 4)))
    hpfg, hpfg_r = Dh(d_in(hfake), frozen=True), Dh(d_in(hraw), frozen=True)
    hloss = T.gan_loss(hpfg, True) + T.feature_matching_loss(hpfg, hp_real) + T.gan_loss(hpfg_r, True) \
        + T.feature_matching_loss(hpfg_r, hp_real)
    hconf = conf[:, 0].cuda().contiguous()
    hreal, hreal_prev = _nhwc(real, 4), _nhwc(real_prev, 4)
    hfake_prev = torch.cat([hprev0, hprev1], 0)
    zero4 = torch.zeros(2, H, W, 4, device="cuda:0")
    from text2video_amd import ops
    with torch.no_grad():
        hfpw = torch.stack([ops.flow_warp(zero4[i], hfake_prev[i], 3) for i in range(2)])
    hloss = hloss + T.masked_l1(hfw, zero4, hconf, 2, 0) * lam_F \
        + T.masked_l1(T._Resample.apply(hfw, hreal_prev, 0), hreal, hconf, 3) * lam_T \
        + T.masked_l1(hfw, None, hconf, 1, 2) + T.masked_l1(hfake, hfpw, hconf, 3) * lam_T
    assert abs(hloss.item() - loss_G.item()) <= 2e-4 * max(1, abs(loss_G.item()))
    hgG = torch.autograd.grad(hloss, list(Gh.parameters()), allow_unused=True)
    got = {k: g for (k, _), g in zip(Gh.named_upstream_parameters().items(), hgG)}
    errs = {}
    for k, r in ref_gG.items():
        assert got[k] is not None, k
        if r.abs().max().item() <= 1e-5:
            continue
        errs[k] = (got[k].cpu() - r).abs().max().item() / r.abs().max().item()
    for prefix in ("model_res_flow", "model_up_flow", "model_final_flow", "model_final_w"):
        ks = [k for k in errs if k.startswith(prefix) and k.endswith("weight")]
        assert ks, prefix
        print(prefix, "max rel err %.1e over %d tensors" % (max(errs[k] for k in ks), len(ks)))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("largest relative gradient errors:", ["%s %.1e" % kv for kv in top], "median %.1e" % float(np.median(list(errs.values()))))
    assert max(errs.values()) <= 3e-3 and float(np.median(list(errs.values()))) <= (1e-4 if size == 32 else 4e-4)


def test_trainer_step_with_flow_branch_runs_and_is_deterministic():
    """Vid2VidTrainer.train_step with the flow branch on (the README recipe passes no --no_flow): all loss terms
    appear and are finite, the flow-branch parameters move, and two runs from the same state give the same losses."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img", "--add_face_disc"])
    assert not opt.no_flow
    H = W = 64
    rng = np.random.default_rng(0)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    real_prev[0, :, : W // 2] = real[0, :, : W // 2]       # a region where the zero reference flow is "confident"
    boxes = [(8, 40, 16, 48)] * 2
    runs = []
    for _ in range(2):
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
        before = {k: v.detach().clone() for k, v in tr.G.named_upstream_parameters().items()}
        l1, prev = tr.train_step(pose, real, boxes, None, real_prev=real_prev)
        l2, _ = tr.train_step(pose, real, boxes, prev, real_prev=real_prev)
        for name in ("G_GAN", "G_GAN_Feat", "D", "F_Flow", "F_Warp", "W", "G_Warp", "G_f_GAN", "D_f"):
            assert name in l1 and np.isfinite(l1[name]), name
        assert l1["W"] > 0 and l1["F_Warp"] > 0
        moved = {k: (v.detach() - before[k]).abs().max().item() for k, v in tr.G.named_upstream_parameters().items()}
        for prefix in ("model_res_flow", "model_up_flow", "model_final_flow", "model_final_w"):
            assert max(v for k, v in moved.items() if k.startswith(prefix)) > 0, prefix
        runs.append((l1, l2))
    for a, b in zip(runs[0], runs[1]):
        for k in a:
            assert a[k] == b[k], (k, a[k], b[k])


def test_first_frame_g_warp_target_is_the_warped_real_previous_frame():
    """No generated previous frame exists for a sequence's first frame: G_Warp there compares the fake with the REAL
    previous frame warped by the reference flow (upstream: fake_B_prev = real_B_prev[:, 0:1] without a previous chunk
    [RECALL compute_fake_B_prev]), not with the generator's all-zero FIFO.  One-frame first chunk, zero reference flow,
    explicit confidence mask: G_Warp == MaskedL1(fake0, real_prev0, conf) * lambda_T."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "1", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "1", "--n_scales_temporal", "0", "--no_first_img"])
    H = W = 64
    rng = np.random.default_rng(3)
    pose = torch.zeros(1, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (1, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(1, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.zeros(1, H, W, 4, device="cuda:0")
    real_prev[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 3)).astype(np.float32))).cuda()
    conf = (torch.from_numpy(rng.random((1, H, W))) < 0.6).float().cuda()
    tr = T.Vid2VidTrainer(opt, "cuda:0", seed=7)
    losses, prev = tr.train_step(pose, real, None, None, real_prev=real_prev, conf_ref=conf)
    fake0 = prev[0, ..., 3:6]                                   # the FIFO's newest frame = the (detached) first fake
    m = conf[0].unsqueeze(-1)
    want = ((fake0 * m) - (real_prev[0, ..., :3] * m)).abs().mean().item() * opt.lambda_T
    zero_target = (fake0 * m).abs().mean().item() * opt.lambda_T          # what the all-zero FIFO would have given
    assert abs(losses["G_Warp"] - want) <= 1e-5 * max(1.0, want), (losses["G_Warp"], want)
    assert abs(want - zero_target) > 1e-3
    # the weight-map loss only exists under --no_first_img
    opt2 = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                 "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "1", "--ndf", "16", "--no_vgg",
                                 "--max_frames_per_gpu", "1", "--n_scales_temporal", "0"])
    tr2 = T.Vid2VidTrainer(opt2, "cuda:0", seed=7)
    prev_in = torch.zeros(1, H, W, 8, device="cuda:0")
    l2, _ = tr2.train_step(pose, real, None, prev_in, real_prev=real_prev, conf_ref=conf)
    assert l2["W"] == 0.0 and losses["W"] > 0.0


def test_flow_branch_gets_its_weight_gradients_when_a_new_sequence_has_no_flow_losses():
    """train_step on a NEW sequence (prev=None) with the flow branch but without real_prev: frame 0 is raw-only and no
    loss reads its flow / weight maps.  Its flow branch is not run (it would be a dead part of the graph whose batched
    Winograd weight-gradient slots nobody reduces), frame 1's is: every flow-branch 3x3 weight must move.  And the
    general safety net: a graph built WITH the dead branch is flushed after the backward pass and yields the gradients of
    the one-by-one reduction."""
    from text2video_amd import ops
    from text2video_amd import train as T
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "1", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img"])
    H, W = 128, 256       # bottleneck 32x64: the ResnetBlock convs run (and differentiate) in the Winograd domain
    rng = np.random.default_rng(5)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    tr = T.Vid2VidTrainer(opt, "cuda:0", seed=9)
    before = {k: v.detach().clone() for k, v in tr.G.named_upstream_parameters().items()}
    tr.train_step(pose, real, None, None)
    moved = {k: (v.detach() - before[k]).abs().max().item() for k, v in tr.G.named_upstream_parameters().items()}
    for k, d in moved.items():
        if k.startswith(("model_res_flow", "model_up_flow")) and k.endswith("weight") and before[k].dim() == 4:
            assert d > 0, k
    # the flush: build the graph with frame 0's flow branch in it but unused, batched reduction on
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    sd = synthetic_state_dict(spec, 3, "vid2vid", flow_gain=0.1)

    def grads(batched, flush, passes=1):
        G = T.TrainableGenerator(spec, sd, "cuda:0")
        params = list(G.parameters())
        prev0 = torch.zeros(1, H, W, 8, device="cuda:0")
        for _ in range(passes):      # (from the second pass on the forward convs keep V in the weight gradients' workspaces)
            ctx = T.batched_weight_gradients(params) if batched else __import__("contextlib").nullcontext()
            with ctx:
                f0, _, _ = G(pose[0:1], prev0, use_raw_only=True, full=True, need_flow=True)     # flow branch built, never used
                prev1 = torch.zeros_like(prev0)
                prev1[..., 3:6] = f0.detach()[..., :3]
                f1, _, _ = G(pose[1:2], prev1, use_raw_only=False, full=True)
                loss = (f0[..., :3] * real[0:1, ..., :3]).sum() + (f1[..., :3] * real[1:2, ..., :3]).sum()
                g = torch.autograd.grad(loss, params, allow_unused=True)
                if flush:
                    g = T.flush_pending_weight_gradients(params, g)
        if passes > 1:
            assert max(getattr(p, "_t2v_wg_expect", 0) for p in params) == 2
        return {k: gi for (k, _), gi in zip(G.named_upstream_parameters().items(), g)}

    ref = grads(False, False)
    with pytest.raises(RuntimeError, match="flush_pending_weight_gradients"):      # the failure mode the flush exists for:
        grads(True, False)                                                          # loud at scope exit, not a lost gradient
    got = grads(True, True)
    flow3x3 = [k for k in ref if k.startswith("model_res_flow") and k.endswith("weight") and ref[k] is not None and ref[k].dim() == 4]
    assert flow3x3
    for k in flow3x3:
        assert got[k] is not None, k
        err = (got[k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-12)
        assert err <= 2e-3, (k, err)
    # ... and with V kept by the forward pass (second pass over the same weights): frame 0's slot of a flow-branch layer holds
    # its V but never receives A dy A^T -- the flush zeroes it and reduces; the same bits as the un-kept flush
    kept = grads(True, True, passes=2)
    for k in flow3x3:
        assert kept[k] is not None and torch.equal(kept[k], got[k]), k


def test_direct_gradient_delivery_equals_autograd_accumulation_and_launches_no_aten_adds():
    """The backward nodes write their parameter gradients straight into the persistent flat buckets (first node writes,
    later nodes add through the kernels' accumulate flags) and hand None to autograd.  Against T2V_GRAD_DIRECT=0 --
    autograd's own accumulation, the sums copied into the buckets afterwards -- the gradients must be bit-equal (same
    node order, same fp32 additions), every gradient must BE a slice of its bucket, and the ATen elementwise-add launches
    of the parameter-gradient accumulation must be gone."""
    import os
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "1", "--no_first_img", "--add_face_disc"])
    H, W = 128, 256
    rng = np.random.default_rng(11)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    boxes = [(16, 48, 100, 132)] * 2
    grads, adds = {}, {}
    old = os.environ.get("T2V_GRAD_DIRECT")
    try:
        for mode in ("1", "0"):
            os.environ["T2V_GRAD_DIRECT"] = mode
            tr = T.Vid2VidTrainer(opt, "cuda:0", seed=13)
            _, prev = tr.train_step(pose, real, boxes, None, real_prev=real_prev)        # fills the temporal history
            # capture the gradients of the second step before Adam consumes them: patch the optimiser steps away
            tr.optG.step = lambda: None
            tr.optD.step = lambda: None
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                tr.train_step(pose, real, boxes, prev, real_prev=real_prev)
                torch.cuda.synchronize()
            names = [e.key for e in prof.key_averages()]
            adds[mode] = sum(e.count for e in prof.key_averages() if "CUDAFunctor_add" in e.key)
            grads[mode] = {k: (None if p.grad is None else p.grad.clone()) for k, p in
                           list(tr.G.named_upstream_parameters().items()) + [("D." + k, p) for k, p in tr.D.named_upstream_parameters().items()]}
            if mode == "1":
                for p, sl in zip(tr.optG.params, tr.bucketsG.slots):
                    assert p.grad is None or p.grad.data_ptr() == sl.view.data_ptr()
                lo = tr.bucketsG.flat.data_ptr()
                assert all(lo <= p.grad.data_ptr() < lo + 4 * tr.bucketsG.flat.numel() for p in tr.optG.params if p.grad is not None)
                assert any("accumulate_kernel" in n or "unzip2_kernel" in n for n in names)
    finally:
        if old is None:
            os.environ.pop("T2V_GRAD_DIRECT", None)
        else:
            os.environ["T2V_GRAD_DIRECT"] = old
    assert grads["1"].keys() == grads["0"].keys()
    for k in grads["1"]:
        a, b = grads["1"][k], grads["0"][k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert torch.equal(a, b), (k, (a - b).abs().max().item())
    print("ATen add launches per step: direct delivery %d, autograd accumulation %d" % (adds["1"], adds["0"]))
    # what is left are the scalar loss sums and autograd's accumulation of ACTIVATION gradients (a frame read by several
    # losses), not parameter gradients
    assert adds["0"] > 50 and adds["1"] <= adds["0"] // 2


def test_shared_discriminator_forward_equals_the_two_forward_form(t2v_env):
    """One discriminator forward on the fake frames serves D's loss (through D's parameters) and G's loss (through the
    frames, D's parameter gradients switched off for that backward pass).  Against the two-forward form upstream runs
    (T2V_D_SHARED_FWD=0: D(fake.detach()) and D(fake) with frozen parameters): the same losses, the same updated weights,
    the same BatchNorm running statistics -- bit for bit -- with the image, face and temporal discriminators on."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "1", "--no_first_img", "--add_face_disc"])
    H, W = 64, 128
    rng = np.random.default_rng(21)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    boxes = [(8, 40, 40, 72)] * 2
    runs = {}
    t2v_env("T2V_D_BATCHED", "0")       # (both forms one pass per launch; the batched default against this form: next test)
    for mode in ("1", "0"):
        t2v_env("T2V_D_SHARED_FWD", mode)
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=17)
        l1, prev = tr.train_step(pose, real, boxes, None, real_prev=real_prev)
        l2, _ = tr.train_step(pose, real, boxes, prev, real_prev=real_prev)          # temporal windows full from here on
        nets = [tr.G, tr.D, tr.Df] + tr.DT
        weights = [p.detach().clone() for n in nets for p in n.parameters()]
        stats = [T.running_stats(p, create=False) for n in nets for k, p in n.named_upstream_parameters().items()
                 if k.endswith(".weight") and p.dim() == 1]
        runs[mode] = (l1, l2, weights, [(s[0].clone(), s[1].clone(), s[2]) for s in stats if s is not None])
    a, b = runs["1"], runs["0"]
    for la, lb in ((a[0], b[0]), (a[1], b[1])):
        assert la.keys() == lb.keys() and all(la[k] == lb[k] for k in la), [(k, la[k], lb[k]) for k in la if la[k] != lb[k]]
    assert "D_T0" in a[1] and "D_f" in a[1]
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert len(a[3]) == len(b[3]) > 4
    for (m1, v1, n1), (m2, v2, n2) in zip(a[3], b[3]):
        assert n1 == n2 and torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.parametrize("extra", [[], ["--no_flow", "--norm", "instance", "--no_ganFeat"]], ids=["flow_batchnorm", "noflow_instancenorm_noFM"])
def test_batched_discriminator_passes_and_one_launch_loss_terms_equal_the_pass_by_pass_step(t2v_env, extra):
    """Round 6 (default path): every discriminator runs its real / fake / raw passes as ONE batch per layer -- BatchNorm
    statistics per pass -- and all LSGAN / feature-matching terms are one launch that also writes the gradient seeds both
    backward passes start from (train.LossBook, t2v_loss_terms).  Against the pass-by-pass step with the scalar graph on
    autograd (T2V_D_BATCHED=0), image + face + temporal discriminators on, flow branch on, two steps (temporal windows full
    in the second): the same forward -- BatchNorm running statistics bit for bit --, the same losses to fp32 rounding of the
    term weights, every delivered gradient equal to rounding."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "16",
                                "--n_downsample_G", "2", "--n_blocks", "2", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "1", "--no_first_img", "--add_face_disc"] + extra)
    # (second case: two passes per discriminator instead of three, statistics per image, no feature-matching terms)
    H, W = 64, 128
    rng = np.random.default_rng(21)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous() if not extra else None
    boxes = [(8, 40, 40, 72)] * 2
    runs = {}
    for mode in ("1", "0"):
        t2v_env("T2V_D_BATCHED", mode)
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=17)
        tr.optG.step = lambda: None      # keep the weights: the two steps' gradients are compared, not Adam's sign decisions
        tr.optD.step = lambda: None
        out = []
        prev = None
        for _ in range(2):
            losses, prev = tr.train_step(pose, real, boxes, prev, real_prev=real_prev)
            grads = [s.view.double().clone() for b in (tr.bucketsG, tr.bucketsD) for s in b.slots if s is not None and s.filled]
            out.append((losses, grads))
        nets = [tr.G, tr.D, tr.Df] + tr.DT
        stats = [T.running_stats(p, create=False) for n in nets for k, p in n.named_upstream_parameters().items()
                 if k.endswith(".weight") and p.dim() == 1]
        runs[mode] = (out, [(s[0].clone(), s[1].clone(), s[2]) for s in stats if s is not None])
    (a, sa), (b, sb) = runs["1"], runs["0"]
    assert "D_T0" in a[1][0] and "D_f" in a[1][0] and "G_f_GAN_Feat" in a[1][0]
    assert (a[1][0]["G_f_GAN_Feat"] == 0.0) == bool(extra)         # (--no_ganFeat: reported as 0 by both forms)
    for (la, ga), (lb, gb) in zip(a, b):
        assert list(la.keys()) == list(lb.keys()), (list(la), list(lb))
        for k in la:
            assert abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
        assert len(ga) == len(gb) > 40
        for i, (x, y) in enumerate(zip(ga, gb)):
            d = (x - y).norm().item() / max(y.norm().item(), 1e-30)
            assert d <= 5e-5, (i, tuple(x.shape), d)       # (seeds rounded once on the host instead of through a chain of fp32 products)
    assert len(sa) == len(sb) and (len(sa) > 4 or extra)        # (no running statistics without affine norms)
    for (m1, v1, n1), (m2, v2, n2) in zip(sa, sb):
        assert n1 == n2 and torch.equal(m1, m2) and torch.equal(v1, v2)


def test_weight_gradients_on_the_side_stream_leave_the_step_unchanged(t2v_env):
    """The weight-gradient kernels of the backward nodes run on a second stream next to the following layers' data
    gradients (train.py: wgrad_fork / wgrad_join), and so do the packed / transformed weight copies of the NEXT step
    right after the optimiser step (prefetch_packs).  Against everything on one stream (both switched off): the same
    losses and the same updated weights, bit for bit, over three steps on one trainer each (a slot read before its
    side-stream kernels finished, or a buffer handed back to the allocator too early, would show here) -- at a size
    where the ResnetBlock convs take the Winograd-domain path and the others the direct one."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "32",
                                "--n_downsample_G", "2", "--n_blocks", "3", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img", "--add_face_disc"])
    H, W = 128, 128
    rng = np.random.default_rng(33)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    boxes = [(16, 80, 32, 96)] * 2
    runs = {}
    for mode in ("1", "0"):
        t2v_env("T2V_WGRAD_STREAM", mode)
        t2v_env("T2V_PACK_PREFETCH", mode)     # (the packed weights of the next step, made on the same side stream)
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=5)
        prev, ls = None, []
        for _ in range(3):
            l, prev = tr.train_step(pose, real, boxes, prev, real_prev=real_prev)
            ls.append(l)
        nets = [tr.G, tr.D, tr.Df]
        runs[mode] = (ls, [p.detach().clone() for n in nets for p in n.parameters()])
        if mode == "1":
            assert T._WG_SIDE["stream"] is not None
            ahead = [p for p in tr.G.parameters() if getattr(p, "_t2v_pack_event", None) is not None]
            assert len(ahead) > 10            # the generator's packed weights for the next step are already under way
    for la, lb in zip(runs["1"][0], runs["0"][0]):
        assert la.keys() == lb.keys() and all(la[k] == lb[k] for k in la), [(k, la[k], lb[k]) for k in la if la[k] != lb[k]]
    assert all(torch.equal(x, y) for x, y in zip(runs["1"][1], runs["0"][1]))


def test_discriminator_backward_on_its_own_stream_equals_the_engines_pass(t2v_env):
    """The discriminators' own backward pass is walked by hand on a second stream before the generator's pass is enqueued
    (Vid2VidTrainer._d_backward_on_its_own_stream) instead of running under the autograd engine after it (T2V_D_BWD_STREAM=0):
    the same nodes, kernels and operands -- losses and updated weights of G, D and the face D bit for bit over three steps,
    with and without the weights' packed copies made ahead (a copy made on the spot by one stream is used by the other)."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "32",
                                "--n_downsample_G", "2", "--n_blocks", "3", "--num_D", "2", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img", "--add_face_disc"])
    H, W = 128, 128
    rng = np.random.default_rng(34)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    boxes = [(16, 80, 32, 96)] * 2
    runs = {}
    for mode in ("1", "1 lazy packs", "0"):
        t2v_env("T2V_D_BWD_STREAM", mode[0])
        t2v_env("T2V_PACK_PREFETCH", "0" if "lazy" in mode else "1")
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=6)
        prev, ls = None, []
        for _ in range(3):
            l, prev = tr.train_step(pose, real, boxes, prev, real_prev=real_prev)
            ls.append(l)
        runs[mode] = (ls, [p.detach().clone() for n in (tr.G, tr.D, tr.Df) for p in n.parameters()])
        assert (getattr(tr, "_d_stream", None) is not None) == (mode[0] == "1")
    for other in ("1 lazy packs", "0"):
        for la, lb in zip(runs["1"][0], runs[other][0]):
            assert la.keys() == lb.keys() and all(la[k] == lb[k] for k in la), [(k, la[k], lb[k]) for k in la if la[k] != lb[k]]
        assert all(torch.equal(x, y) for x, y in zip(runs["1"][1], runs[other][1]))


def test_train_step_with_the_fixed_grid_kernels_forced_is_bit_identical(t2v_env, monkeypatch):
    """Both fixed-grid kernels (Winograd GEMM stage in the forward convs and the transposed data gradient; Winograd-domain
    weight-gradient reduction) forced on at a size whose tile counts are far below the grid -- short runs, many blocks idle,
    every tile whole or cut once -- against one block per tile: losses and updated weights bit for bit."""
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "64",
                                "--n_downsample_G", "1", "--n_blocks", "2", "--num_D", "1", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img"])
    H, W = 128, 128      # bottleneck 64 x 64 x 128 channels (ngf 64, one downsampling): 256 tile rows, one 128-wide channel tile
    rng = np.random.default_rng(8)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    runs = {}
    hints = []
    real_hint = T.ops.set_overlap_hint
    monkeypatch.setattr(T.ops, "set_overlap_hint", lambda on: (hints.append(on), real_hint(on))[1])
    # "2": forced, and from the step's first side-stream weight gradient on ONE block per CU (overlap hint 2: the two streams'
    # fixed-grid launches resident side by side); "2 two-per-CU": forced, two per CU throughout; "0": one block per tile
    for mode in ("2", "2 two-per-CU", "0"):
        t2v_env("T2V_WINO_GEMM_SK", mode[0])
        t2v_env("T2V_WGRAD_SK", mode[0])
        t2v_env("T2V_TRAIN_SK_HINT", "0" if "two-per-CU" in mode else "1")
        del hints[:]
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=9)
        prev, ls = None, []
        for _ in range(2):
            l, prev = tr.train_step(pose, real, None, prev, real_prev=real_prev)
            ls.append(l)
        runs[mode] = (ls, [p.detach().clone() for n in (tr.G, tr.D) for p in n.parameters()])
        assert (2 in hints) == ("two-per-CU" not in mode) and hints[-1] == 0      # raised inside the step, lowered at its end
    for other in ("2 two-per-CU", "0"):
        for la, lb in zip(runs["2"][0], runs[other][0]):
            assert la.keys() == lb.keys() and all(la[k] == lb[k] for k in la), [(k, la[k], lb[k]) for k in la if la[k] != lb[k]]
        assert all(torch.equal(x, y) for x, y in zip(runs["2"][1], runs[other][1]))




@pytest.mark.parametrize("first,second", [(2, 1), (1, 2), (2, 3)], ids=["fewer", "more", "three"])
def test_kept_workspaces_when_a_step_uses_a_layer_more_or_less_often_than_the_one_before(first, second):
    """The workspace that keeps V is sized for the uses the PREVIOUS step counted (train._keep_v_slot).  A step with fewer
    uses zeroes the slots nobody filled before the one reduction; one with more reduces the extra uses on the spot, added
    to the same gradient.  Against the same graph on fresh weights (nothing kept): the ResnetBlock weights' gradients to
    rounding."""
    from text2video_amd import train as T
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=2, no_flow=True, norm="batch")
    sd = synthetic_state_dict(spec, 3, "vid2vid")
    H, W = 128, 256
    rng = np.random.default_rng(12)
    pose = torch.zeros(3, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (3, H, W, 9)).astype(np.float32)).cuda()
    tgt = torch.from_numpy(rng.standard_normal((3, H, W, 3)).astype(np.float32)).cuda()

    def run(G, frames):
        params = list(G.parameters())
        with T.batched_weight_gradients(params):
            loss = 0.0
            prev = torch.zeros(1, H, W, 8, device="cuda:0")
            for i in range(frames):
                f, _, _ = G(pose[i:i + 1], prev, use_raw_only=True, full=True)
                loss = loss + (f[..., :3] * tgt[i:i + 1]).sum()
            g = T.flush_pending_weight_gradients(params, torch.autograd.grad(loss, params, allow_unused=True))
        return {k: gi for (k, _), gi in zip(G.named_upstream_parameters().items(), g)}

    G = T.TrainableGenerator(spec, sd, "cuda:0")
    run(G, first)
    res = [p for k, p in G.named_upstream_parameters().items() if k.startswith("model_res") and p.dim() == 4]
    assert res and all(getattr(p, "_t2v_wg_expect", 0) == first for p in res)
    got = run(G, second)
    ref = run(T.TrainableGenerator(spec, sd, "cuda:0"), second)
    keys = [k for k in ref if k.startswith("model_res") and ref[k] is not None and ref[k].dim() == 4]
    assert keys
    for k in keys:
        err = (got[k] - ref[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-12)
        assert err <= 1e-5, (k, err)


def test_kept_input_transforms_and_in_place_forward_weights_leave_the_step_unchanged(t2v_env, monkeypatch):
    """Round 5: from a layer's second step on the forward conv writes its input transform V into the weight gradient's
    workspace (train._keep_v_slot: backward transforms dy only -- and forms the gradient in front of the norm inside that
    transform instead of writing it out), and the data gradient reads the forward packing of the
    weights in place (the [K][N] form of the fixed-grid GEMM) instead of a transposed copy.  Against all three switched off:
    losses and updated weights bit for bit over three steps -- the slots are handed out so that the reduction order is the
    un-kept path's -- and the kept path really ran."""
    from text2video_amd import ops, train as T
    from text2video_amd.options import TrainOptions
    opt = TrainOptions().parse(["--name", "t", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--ngf", "64",
                                "--n_downsample_G", "1", "--n_blocks", "2", "--num_D", "1", "--ndf", "16", "--no_vgg",
                                "--max_frames_per_gpu", "2", "--n_scales_temporal", "0", "--no_first_img"])
    H, W = 128, 128      # bottleneck 64 x 64 x 128 channels: 256 tile rows, one 128-wide channel tile
    rng = np.random.default_rng(8)
    pose = torch.zeros(2, H, W, 12, device="cuda:0")
    pose[..., :9] = torch.from_numpy(rng.uniform(-1, 1, (2, H, W, 9)).astype(np.float32)).cuda()
    real = torch.zeros(2, H, W, 4, device="cuda:0")
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((2, H, W, 3)).astype(np.float32))).cuda()
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    t2v_env("T2V_WINO_GEMM_SK", "2")          # (the fixed grid wherever the shape allows: these tile counts are below it)
    calls = {"dy": 0, "fw": 0}
    dy_only, dgrad = ops.conv2d_backward_weight_winograd_dy_norm, ops.conv2d_backward_data_winograd
    # (the transform of dy that also forms the gradient in front of the norm: it runs where V was kept -- every such layer here)
    monkeypatch.setattr(ops, "conv2d_backward_weight_winograd_dy_norm",
                        lambda *a, **k: (calls.__setitem__("dy", calls["dy"] + 1), dy_only(*a, **k))[1])
    monkeypatch.setattr(ops, "conv2d_backward_data_winograd",
                        lambda *a, **k: (calls.__setitem__("fw", calls["fw"] + int(k.get("forward_weights", False))), dgrad(*a, **k))[1])
    runs = {}
    for mode in ("1", "0"):
        t2v_env("T2V_WGRAD_KEEP_V", mode)
        t2v_env("T2V_DGRAD_FORWARD_WEIGHTS", mode)
        t2v_env("T2V_DY_NORM_FUSED", mode)
        tr = T.Vid2VidTrainer(opt, "cuda:0", seed=9)
        prev, ls = None, []
        for _ in range(3):
            l, prev = tr.train_step(pose, real, None, prev, real_prev=real_prev)
            ls.append(l)
        runs[mode] = (ls, [p.detach().clone() for n in (tr.G, tr.D) for p in n.parameters()], dict(calls))
        if mode == "1":
            assert max(getattr(p, "_t2v_wg_expect", 0) for p in tr.G.parameters()) == 2      # two frames per step
    kept, plain = runs["1"][2], runs["0"][2]
    # steps 2 and 3 kept V (the first has no expectation yet); with both off nothing more was added
    assert kept["dy"] > 0 and kept["dy"] % 4 == 0 and plain["dy"] == kept["dy"], (kept, plain)
    assert kept["fw"] > 0 and plain["fw"] == kept["fw"], (kept, plain)
    for la, lb in zip(runs["1"][0], runs["0"][0]):
        assert la.keys() == lb.keys() and all(la[k] == lb[k] for k in la), [(k, la[k], lb[k]) for k in la if la[k] != lb[k]]
    assert all(torch.equal(x, y) for x, y in zip(runs["1"][1], runs["0"][1]))


def test_weight_gradients_reduced_once_per_layer_equal_one_reduction_per_pass(monkeypatch):
    """Round 5: a direct-kernel layer that ran several times in the step's graph -- the generator's stride-2 / transposed
    layers on the clip's two frames, the discriminators' layers on the real, fake and raw pass (two frames each) -- reduces its
    weight gradient in ONE launch over all of them (train._paired_direct_wgrad; T2V_WGRAD_PAIR=0: one launch per backward node).
    Same products, another summation tree: every gradient the step delivers equal to rounding; same losses, bit for bit; and the
    discriminators' layers really were collected (three passes in one reduction)."""
    from text2video_amd import ops
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    size, dev, F = 128, "cuda:0", 2
    argv = ["--name", "b", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2", "--max_frames_per_gpu", "2",
            "--n_scales_temporal", "0", "--no_first_img", "--fineSize", str(size), "--no_vgg", "--add_face_disc", "--ngf", "32",
            "--n_blocks", "3", "--ndf", "32"]
    rng = np.random.default_rng(0)
    H = W = size
    pose = torch.zeros(F, H, W, 12, device=dev)
    pose[..., :9] = torch.from_numpy(np.where(rng.random((F, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F, H, W, 9)), -1.0).astype(np.float32)).to(dev)
    real = torch.zeros(F, H, W, 4, device=dev)
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F, H, W, 3)).astype(np.float32))).to(dev)
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    boxes = [(16, 16 + 64, 32, 32 + 64)] * F
    prev = torch.zeros(1, H, W, 8, device=dev)
    prev[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).to(dev)
    batches = []
    real_bw = ops.conv2d_backward_weight

    def spy(x, dy, desc, accumulate_into=None):
        batches.append(int(x.shape[0]) if x.dim() == 4 else 1)
        return real_bw(x, dy, desc, accumulate_into)
    res = {}
    monkeypatch.setenv("T2V_D_BATCHED", "0")       # (the batched default has one node per discriminator layer to begin with)
    for mode in ("0", "1"):
        monkeypatch.setenv("T2V_WGRAD_PAIR", mode)
        tr = T.Vid2VidTrainer(TrainOptions().parse(argv), dev, seed=1)
        tr.optG.step = lambda: None      # keep the weights: only the gradients of this one step are compared
        tr.optD.step = lambda: None
        del batches[:]
        monkeypatch.setattr(ops, "conv2d_backward_weight", spy)
        losses = tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
        torch.cuda.synchronize()
        monkeypatch.setattr(ops, "conv2d_backward_weight", real_bw)
        res[mode] = (tr, {k: float(v) for k, v in losses.items()}, list(batches))
    assert res["0"][1] == res["1"][1]                                  # the forward pass is the same launches
    assert max(res["0"][2]) == 2 and 6 in res["1"][2]                  # real + fake + raw passes of two frames each: one batch of 6
    assert len(res["1"][2]) < len(res["0"][2])
    for name in ("bucketsG", "bucketsD"):
        b0, b1 = getattr(res["0"][0], name), getattr(res["1"][0], name)
        n = 0
        for s0, s1 in zip(b0.slots, b1.slots):
            if s0 is None or s1 is None or not s0.filled:
                continue
            a, b = s0.view.double(), s1.view.double()
            d = (a - b).norm().item() / max(a.norm().item(), 1e-30)
            assert d <= 2e-5, (name, n, tuple(a.shape), d)
            n += 1
        assert n >= 20, (name, n)
