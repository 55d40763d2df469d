"""GPU parity of the backward kernels against torch autograd on the CPU (fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _nhwc_batch(x, cs=None):
    from text2video_amd import ops
    return torch.stack([ops.nchw_to_nhwc(x[b].to("cuda:0").contiguous(), cs) for b in range(x.shape[0])])


def _ref_forward(x, w, b, k, stride, pad, pad_mode, transposed):
    if transposed:
        return F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    if pad_mode == 1 and pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)


WGRAD_CASES = [
    # name, B, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed
    ("rb3x3_reflect", 2, 16, 16, 64, 128, 3, 1, 1, 1, False),
    ("rb3x3_reflect_wide", 1, 12, 20, 160, 192, 3, 1, 1, 1, False),     # 2x2 channel tiles with tails
    ("down3x3_s2", 2, 16, 16, 32, 64, 3, 2, 1, 0, False),
    ("convT", 2, 8, 8, 64, 32, 3, 2, 1, 0, True),
    ("stem7x7_cin9", 1, 20, 20, 9, 32, 7, 1, 3, 1, False),
    ("head7x7_cout3", 1, 20, 16, 32, 3, 7, 1, 3, 1, False),
    # few output channels on a wide input: taps folded onto the dY side over a padded copy of x
    ("head7x7_cout3_wide", 1, 20, 16, 128, 3, 7, 1, 3, 1, False),
    ("head7x7_cout3_wide_batch", 2, 18, 24, 64, 3, 7, 1, 3, 1, False),
    ("k3_cout4_zero_pad", 2, 16, 16, 96, 4, 3, 1, 1, 0, False),
    ("k3_cout6_nopad", 1, 16, 20, 64, 6, 3, 1, 0, 0, False),
    ("disc4x4_s2_p2", 2, 16, 16, 8, 64, 4, 2, 2, 0, False),
    ("disc4x4_s1_p2", 1, 9, 9, 64, 1, 4, 1, 2, 0, False),
    ("stem7x7_pixel_split", 1, 96, 96, 9, 32, 7, 1, 3, 1, False),       # 49 blocks, 288 stages -> split reduction
    ("disc_first_pixel_split", 2, 64, 64, 6, 64, 4, 2, 2, 0, False),
    ("convT_pixel_split", 1, 48, 48, 32, 32, 3, 2, 1, 0, True),
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_weight_and_bias_gradient(case):
    from text2video_amd import ops
    name, B, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed = case
    x = _rand(B, Cin, H, W, seed=1)
    w = _rand(*((Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)), seed=2, scale=0.1).requires_grad_()
    b = _rand(Cout, seed=3, scale=0.1).requires_grad_()
    y = _ref_forward(x, w, b, k, stride, pad, pad_mode, transposed)
    dy = _rand(*y.shape, seed=4)
    y.backward(dy)
    desc = ops.conv_desc(H, W, Cin, Cout, k, stride, pad, pad_mode, transposed)
    xs = _nhwc_batch(x)
    dys = _nhwc_batch(dy)
    dwp = ops.conv2d_backward_weight(xs, dys, desc)
    got = ops.unpack_conv_weight(dwp, desc, xs.shape[-1]).cpu()
    scale = max(1.0, w.grad.abs().max().item())
    assert got.shape == w.grad.shape
    assert (got - w.grad).abs().max().item() <= 2e-4 * scale, name
    # accumulate: a second pass doubles it
    ops.conv2d_backward_weight(xs, dys, desc, accumulate_into=dwp)
    got2 = ops.unpack_conv_weight(dwp, desc, xs.shape[-1]).cpu()
    assert (got2 - 2 * w.grad).abs().max().item() <= 4e-4 * scale
    db = ops.channel_sum(dys, Cout).cpu()
    assert (db - b.grad).abs().max().item() <= 1e-4 * max(1.0, b.grad.abs().max().item())


def test_pack_unpack_roundtrip():
    from text2video_amd import ops
    for (Cin, Cout, k, tr) in [(9, 32, 7, False), (64, 48, 3, False), (32, 16, 3, True)]:
        desc = ops.conv_desc(16, 16, Cin, Cout, k, 2 if tr else 1, 1 if tr else k // 2, 0, tr)
        w = _rand(*((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)), seed=5).cuda()
        xcs = ops.round_up(Cin, 4)
        assert torch.equal(ops.unpack_conv_weight(ops.pack_conv_weight(w, desc, xcs), desc, xcs), w)


DGRAD_CASES = [c for c in WGRAD_CASES if c[0] not in ("stem7x7_cin9",) and "split" not in c[0]] + [
    ("disc4x4_s2_p2_odd", 1, 17, 17, 8, 16, 4, 2, 2, 0, False),      # odd input: output_padding 1, odd convT output
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv_data_gradient_via_adjoint_forward_conv(case):
    from text2video_amd import ops
    from text2video_amd.backward import ConvDataGrad
    name, B, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed = case
    x = _rand(1, Cin, H, W, seed=11).requires_grad_()
    w = _rand(*((Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)), seed=12, scale=0.1)
    y = _ref_forward(x, w, None, k, stride, pad, pad_mode, transposed)
    dy = _rand(*y.shape, seed=13)
    y.backward(dy)
    desc = ops.conv_desc(H, W, Cin, Cout, k, stride, pad, pad_mode, transposed)
    dg = ConvDataGrad(desc).refresh(w.cuda())
    dx = dg(_nhwc_batch(dy)[0])
    got = dx[..., :Cin].permute(2, 0, 1).cpu()
    assert got.shape == x.grad.shape[1:]
    assert (got - x.grad[0]).abs().max().item() <= 2e-4 * max(1.0, x.grad.abs().max().item()), name


@pytest.mark.parametrize("relu,affine", [(0, False), (1, True), (2, True), (1, False)])
def test_norm_backward_with_fused_activation(relu, affine):
    from text2video_amd import ops
    B, C, H, W = 2, 64, 12, 10
    x = _rand(B, C, H, W, seed=21).requires_grad_()
    g = (1 + _rand(C, seed=22, scale=0.1)).requires_grad_()
    bt = _rand(C, seed=23, scale=0.3).requires_grad_()
    # batch statistics over (B,H,W): BatchNorm2d in train mode
    y = torch.nn.functional.batch_norm(x, None, None, g if affine else None, bt if affine else None, True, 0.1, 1e-5)
    y = torch.relu(y) if relu == 1 else (torch.nn.functional.leaky_relu(y, 0.2) if relu == 2 else y)
    dy = _rand(B, C, H, W, seed=24)
    y.backward(dy)
    xs = _nhwc_batch(x.detach())
    mean = x.detach().mean((0, 2, 3))
    rstd = 1.0 / torch.sqrt(x.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    mr = torch.stack([mean, rstd], 1).contiguous().cuda()
    dx, sums = ops.instance_norm_backward(xs, _nhwc_batch(dy), mr, g.detach().cuda() if affine else None,
                                          bt.detach().cuda() if affine else None, relu)
    assert (dx.permute(0, 3, 1, 2).cpu() - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item())
    if affine:
        assert (sums[:, 0].cpu() - bt.grad).abs().max().item() <= 1e-4 * max(1.0, bt.grad.abs().max().item())
        assert (sums[:, 1].cpu() - g.grad).abs().max().item() <= 1e-4 * max(1.0, g.grad.abs().max().item())


def test_pointwise_and_pooling_backward():
    from text2video_amd import ops
    # reflect pad adjoint
    x = _rand(1, 8, 9, 11, seed=31).requires_grad_()
    for p in (1, 3):
        x.grad = None
        yp = F.pad(x, (p, p, p, p), mode="reflect")
        d = _rand(*yp.shape, seed=32)
        yp.backward(d)
        got = ops.reflect_pad_backward(_nhwc_batch(d)[0], p)
        assert (got.permute(2, 0, 1).cpu() - x.grad[0]).abs().max().item() <= 1e-5
    # avg pool (count_include_pad=False)
    for (H, W) in [(16, 16), (17, 13)]:
        a = _rand(1, 4, H, W, seed=33).requires_grad_()
        y = F.avg_pool2d(a, 3, 2, 1, count_include_pad=False)
        d = _rand(*y.shape, seed=34)
        y.backward(d)
        got = ops.avgpool3x3s2_backward(_nhwc_batch(d)[0], H, W)
        assert (got.permute(2, 0, 1).cpu() - a.grad[0]).abs().max().item() <= 1e-6
    # tanh / sigmoid / leaky from the output
    pre = _rand(1000, seed=35).requires_grad_()
    for act, fn in [(ops.ACT_TANH, torch.tanh), (2, torch.sigmoid), (ops.ACT_LRELU, lambda t: F.leaky_relu(t, 0.2))]:
        pre.grad = None
        y = fn(pre)
        d = _rand(1000, seed=36)
        y.backward(d)
        got = ops.act_backward(d.cuda(), y.detach().cuda(), act, 0.2)
        assert (got.cpu() - pre.grad).abs().max().item() <= 1e-6
    # losses
    a, b = _rand(500, seed=37).requires_grad_(), _rand(500, seed=38)
    (((a - 1.0) ** 2).sum() * 0.01).backward()
    assert (ops.sum_sq_diff_const_backward(a.detach().cuda(), 1.0, 0.01).cpu() - a.grad).abs().max().item() <= 1e-6
    a.grad = None
    ((a - b).abs().sum() * 0.02).backward()
    assert (ops.sum_abs_diff_backward(a.detach().cuda(), b.cuda(), 0.02).cpu() - a.grad).abs().max().item() <= 1e-7


@pytest.mark.parametrize("case", [(1, 64, 32, 32, 64, 1, 1), (2, 16, 32, 64, 32, 1, 1), (1, 18, 30, 32, 32, 0, 2), (1, 64, 85, 32, 32, 1, 1)],
                         ids=["T128", "batch2", "zero_pad2", "ragged_64x85"])
def test_weight_gradient_in_winograd_domain(case):
    """dW through F(4x4,3x3) transforms + 36 pixel-reduction GEMMs == torch autograd (and the direct kernel)."""
    import torch.nn.functional as F
    from text2video_amd import ops
    B, H, W, Cin, Cout, reflect, pad = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).requires_grad_(True)
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect") if reflect else F.pad(x, (pad, pad, pad, pad))
    y = F.conv2d(xp, w)
    dy = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward()
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, pad, ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyh = dy.permute(0, 2, 3, 1).contiguous().cuda()
    assert ops.backward_weight_winograd_supported(desc, Cin, Cout)
    dw = ops.conv2d_backward_weight_winograd(xh, dyh, desc)
    ref = w.grad
    assert (dw.cpu() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    dwd = ops.unpack_conv_weight(ops.conv2d_backward_weight(xh, dyh, desc), desc, Cin)
    assert (dw - dwd).abs().max().item() <= 2e-4 * ref.abs().max().item()
    # accumulate
    dw2 = ops.conv2d_backward_weight_winograd(xh, dyh, desc, accumulate_into=dw.clone())
    assert (dw2 - 2 * dw).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("H,W,flow_scale", [(24, 40, 3.0), (17, 23, 12.0), (64, 64, 0.7)])
def test_flow_warp_composite_backward_matches_oracle_autograd(H, W, flow_scale):
    """t2v_flow_warp_composite_backward (SpatialGridSamplerBilinear_updateGradInput + the blend's adjoint) against
    autograd through the oracle's `resample` (torch 0.4.1's evaluation order, oracle/generator_ref.py) and, away from
    the image border, through modern F.grid_sample as a second opinion."""
    from oracle.generator_ref import resample, resample_modern
    from text2video_amd import ops
    raw = torch.tanh(_rand(1, 3, H, W, seed=1)).requires_grad_()
    prev = torch.tanh(_rand(1, 6, H, W, seed=2)).requires_grad_()
    flow = (_rand(1, 2, H, W, seed=3) * flow_scale).requires_grad_()      # pixels; large ones leave the image
    wt = torch.sigmoid(_rand(1, 1, H, W, seed=4)).requires_grad_()
    g = _rand(1, 3, H, W, seed=5)
    warp = resample(prev[:, 3:6], flow)
    out = raw * wt + warp * (1 - wt)
    ref = torch.autograd.grad((out * g).sum(), [raw, flow, wt, prev])
    out_m = raw * wt + resample_modern(prev[:, 3:6], flow) * (1 - wt)
    ref_m = torch.autograd.grad((out_m * g).sum(), [flow])[0]

    def nhwc(t, cs):
        o = torch.zeros(H, W, cs, device="cuda:0")
        o[..., :t.shape[1]] = t.detach()[0].permute(1, 2, 0).cuda()
        return o
    fw = torch.zeros(H, W, 4, device="cuda:0")
    fw[..., :2] = flow.detach()[0].permute(1, 2, 0).cuda()
    fw[..., 2] = wt.detach()[0, 0].cuda()
    d_raw, d_fw, d_prev = ops.flow_warp_composite_backward(nhwc(g, 4), None, nhwc(raw, 4), fw, nhwc(prev, 8), 3, True)
    assert (d_raw[..., :3].permute(2, 0, 1).cpu() - ref[0][0]).abs().max().item() <= 1e-6
    assert (d_fw[..., 2].cpu() - ref[2][0, 0]).abs().max().item() <= 1e-5 * max(1.0, ref[2].abs().max().item())
    scale = ref[1].abs().max().item()
    assert (d_fw[..., :2].permute(2, 0, 1).cpu() - ref[1][0]).abs().max().item() <= 2e-5 * max(1.0, scale)
    assert d_fw[..., 3].abs().max().item() == 0.0
    dp = d_prev.permute(2, 0, 1).cpu()
    # (atomic accumulation: the summation order over the taps that hit one pixel is not fixed)
    assert (dp[:6] - ref[3][0]).abs().max().item() <= 2e-5 * max(1.0, ref[3].abs().max().item()) and dp[6:].abs().max().item() == 0.0
    # second opinion: modern grid_sample agrees wherever the sampling position is strictly inside the image and not
    # within rounding distance of an integer position (there the interpolant has a kink)
    xs = torch.arange(W).view(1, W) + flow.detach()[0, 0]
    ys = torch.arange(H).view(H, 1) + flow.detach()[0, 1]
    safe = (xs > 0.01) & (xs < W - 1.01) & (ys > 0.01) & (ys < H - 1.01) & \
           ((xs - xs.round()).abs() > 1e-3) & ((ys - ys.round()).abs() > 1e-3)
    assert safe.float().mean().item() > 0.1
    dm = (d_fw[..., :2].permute(2, 0, 1).cpu() - ref_m[0]).abs() * safe
    assert dm.max().item() <= 2e-5 * max(1.0, scale)
    # outside the image the flow gradient of that axis is exactly zero
    outx = (xs < 0) | (xs > W - 1)
    if outx.any():
        assert d_fw[..., 0].cpu()[outx].abs().max().item() == 0.0
    # plain resample (the warp losses): d_warp path, no blend
    ref_w = torch.autograd.grad((resample(prev[:, 3:6], flow) * g).sum(), [flow])[0]
    _, d_fw2, _ = ops.flow_warp_composite_backward(None, nhwc(g, 4), None, fw, nhwc(prev, 8), 3, False)
    assert (d_fw2[..., :2].permute(2, 0, 1).cpu() - ref_w[0]).abs().max().item() <= 2e-5 * max(1.0, ref_w.abs().max().item())
    assert d_fw2[..., 2].abs().max().item() == 0.0
    got = ops.flow_warp(fw, nhwc(prev, 8), 3)
    assert (got[..., :3].permute(2, 0, 1).cpu() - resample(prev[:, 3:6], flow).detach()[0]).abs().max().item() <= 1e-5


def test_masked_l1_and_flow_head_activation_backward():
    from text2video_amd import ops
    from text2video_amd import train as T
    H, W = 20, 28
    a = _rand(2, H, W, 4, seed=1).cuda().requires_grad_()
    b = _rand(2, H, W, 4, seed=2).cuda()
    mask = (torch.from_numpy(np.random.default_rng(3).random((2, H, W))) < 0.6).float().cuda()
    for c0, C, tgt in [(0, 2, b), (2, 1, None), (0, 3, b)]:
        got = T.masked_l1(a, tgt, mask, C, c0)
        t = tgt[..., c0:c0 + C] if tgt is not None else 0.0
        want = ((a[..., c0:c0 + C] * mask[..., None]) - (t * mask[..., None])).abs().mean()
        assert abs(got.item() - want.item()) <= 1e-6 * max(1.0, abs(want.item()))
        gg, = torch.autograd.grad(got, [a])
        gw, = torch.autograd.grad(want, [a])
        assert (gg - gw).abs().max().item() <= 1e-9
    # fused flow / weight head: y = (20*z0, 20*z1, sigmoid(z2), .)
    z = _rand(H, W, 4, seed=5).cuda()
    y = torch.stack([20 * z[..., 0], 20 * z[..., 1], torch.sigmoid(z[..., 2]), torch.zeros_like(z[..., 3])], -1).contiguous()
    dy = _rand(H, W, 4, seed=6).cuda()
    d = ops.act_backward(dy, y, 4, 20.0)
    want = torch.stack([20 * dy[..., 0], 20 * dy[..., 1], dy[..., 2] * y[..., 2] * (1 - y[..., 2]), torch.zeros_like(z[..., 3])], -1)
    assert (d - want).abs().max().item() <= 1e-6


def test_in_kernel_combine_of_split_weight_gradients_equals_the_reduce_launch(t2v_env):
    """Direct weight gradients whose pixel reduction is cut into a few ranges: the block that draws the last arrival
    ticket of a (tap, n, c) tile sums the published partials in split order inside the launch (default, <= 4 partials)
    -- bit for bit what the separate reduce launch over zero-filled slabs produces (T2V_WGRAD_COMBINE=0), run after
    run, with and without accumulation into an existing gradient."""
    from text2video_amd import ops
    torch.manual_seed(0)
    for (B, H, W, Cin, Cout, stride, tr) in [(2, 16, 16, 64, 128, 1, False), (1, 64, 64, 256, 256, 1, False),
                                              (1, 32, 32, 256, 512, 2, False), (1, 16, 16, 512, 256, 2, True)]:
        desc = ops.conv_desc(H, W, Cin, Cout, 3, stride, 1, ops.PAD_ZERO if stride == 2 else ops.PAD_REFLECT, tr)
        ho, wo = ops.conv_out_dims(desc)
        x = torch.randn(B, H, W, Cin, device="cuda:0")
        dy = torch.randn(B, ho, wo, Cout, device="cuda:0")
        outs = {}
        for mode in ("1", "0"):
            t2v_env("T2V_WGRAD_COMBINE", mode)
            t2v_env("T2V_WGRAD_COMBINE_MAX", "64")       # also the many-partial shapes the default leaves to the reduce launch
            for rep in range(2):
                dwp = ops.conv2d_backward_weight(x, dy, desc)
                acc = ops.conv2d_backward_weight(x, dy, desc, accumulate_into=dwp.clone())
                outs[(mode, rep)] = (dwp.clone(), acc.clone())
        for rep in range(2):
            assert torch.equal(outs[("1", rep)][0], outs[("0", 0)][0]), (B, H, W, Cin, Cout, stride, tr)
            assert torch.equal(outs[("1", rep)][1], outs[("0", 0)][1])
        assert outs[("0", 0)][0].abs().max().item() > 1.0


@pytest.mark.parametrize("geom", [(64, 64, 1024, 1024, 2), (64, 88, 640, 896, 2), (32, 32, 512, 488, 3)], ids=["rb1024x2", "ragged", "cout488"])
def test_fixed_grid_winograd_domain_weight_gradient_equals_tile_per_block(geom, t2v_env):
    """The 36 Winograd-domain reductions dU[xi] = M_dy[xi]^T V[xi] on a fixed grid (conv_wgrad.hip: wino_wgrad_sk_kernel;
    a tile cut between two blocks is finished from the first block's accumulators) against one block per tile
    (T2V_WGRAD_SK=0): the same t-ordered MFMA chain per element, so dW is BIT-identical -- launch after launch on one
    workspace, with channel counts that are not multiples of the 128-wide tiles, and with accumulation."""
    from text2video_amd import ops
    H, W, Cin, Cout, B = geom
    torch.manual_seed(4)
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT)
    x = torch.randn(B, H, W, Cin, device="cuda:0")
    dy = torch.randn(B, H, W, Cout, device="cuda:0")
    ws = ops.backward_weight_winograd_workspace(desc, Cin, B, x.device)
    ws.fill_(float("nan"))
    ops.conv2d_backward_weight_winograd_stages(x, dy, desc, ws, B, 0, False)
    t2v_env("T2V_WGRAD_SK", "0")
    want = ops.conv2d_backward_weight_winograd_reduce(desc, ws, B, Cin, Cout).clone()
    base = torch.randn_like(want)
    want_acc = ops.conv2d_backward_weight_winograd_reduce(desc, ws, B, Cin, Cout, out=base.clone(), accumulate=True).clone()
    t2v_env("T2V_WGRAD_SK", "1")
    for rep in range(6):
        got = ops.conv2d_backward_weight_winograd_reduce(desc, ws, B, Cin, Cout)
        assert torch.equal(got, want), "launch %d: %d of %d differ" % (rep, int((got != want).sum()), got.numel())
    got_acc = ops.conv2d_backward_weight_winograd_reduce(desc, ws, B, Cin, Cout, out=base.clone(), accumulate=True)
    assert torch.equal(got_acc, want_acc)
    assert bool(torch.isfinite(want).all()) and want.abs().max().item() > 1.0


@pytest.mark.parametrize("H,W,Cin,Cout", [(32, 32, 64, 64), (16, 32, 128, 64), (64, 64, 128, 128), (64, 128, 256, 128)],
                         ids=["32x32", "16x32", "64x64", "64x128"])
def test_transposed_winograd_data_gradient_matches_autograd(H, W, Cin, Cout, t2v_env):
    """Data gradient of the ResnetBlock conv (3x3, ReflectionPad 1) by the transposed Winograd algorithm: A dy A^T is read out
    of the weight gradient's batch workspace (two images, both slots), dV = U^T dM on the layer's own tiles, patches
    overlap-added and folded.  Against torch autograd on the CPU (fp64) and against the full-correlation form."""
    from text2video_amd import ops
    from text2video_amd.backward import ConvDataGrad
    g = torch.Generator().manual_seed(3)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(B, Cout, H, W, generator=g, dtype=torch.float64)
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), w)
    (want,) = torch.autograd.grad(y, x, dy)
    desc = ops.with_algo(ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT), ops.ALGO_WINOGRAD_F4)
    assert ops.backward_data_winograd_supported(desc, Cin, Cout)
    xs = x.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
    dys = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.float().cuda()
    ws = ops.backward_weight_winograd_workspace(desc, Cin, B, "cuda:0")
    ops.conv2d_backward_weight_winograd_stages(xs[0:1], dys[0:1], desc, ws, B, 0, False)      # slot 0
    ops.conv2d_backward_weight_winograd_stages(xs[1:2], dys[1:2], desc, ws, B, 1, False)      # slot 1
    ut = ops.pack_conv_weight_transposed(wd, desc, Cin)
    full = ConvDataGrad(desc).refresh(wd)
    scale = want.abs().max().item()
    for b in range(B):
        got = ops.conv2d_backward_data_winograd(desc, B, b, ws, Cin, ut).permute(2, 0, 1).cpu().double()
        err = (got - want[b]).abs().max().item() / scale
        ref = full(dys[b]).permute(2, 0, 1).cpu().double()
        err_full = (ref - want[b]).abs().max().item() / scale
        print("%dx%d C%d->%d image %d: transposed algorithm %.2e, full-correlation form %.2e (relative to max|dx|)"
              % (H, W, Cin, Cout, b, err, err_full))
        assert err <= 2e-5 and err <= 4 * err_full + 1e-6
    # the GEMM stage on the fixed grid (reading its rows out of the batch-wide matrix: row pitch != rows used) gives the
    # same bits as one block per tile -- forced here, the tile count of these shapes is below the grid
    if desc.H * desc.W // 16 % 128 == 0 and Cin % 128 == 0:
        outs = {}
        for mode in ("0", "2"):
            t2v_env("T2V_WINO_GEMM_SK", mode)
            outs[mode] = [ops.conv2d_backward_data_winograd(desc, B, b, ws, Cin, ut).clone() for b in range(B)]
        assert all(torch.equal(a, c) for a, c in zip(outs["0"], outs["2"]))
        # ... and so does the [K][N] form of that GEMM, which reads the FORWARD layer's packing in place of the transposed copy
        if Cout % 128 == 0:
            assert ops.backward_data_winograd_takes_forward_weights(desc, Cin, Cout)      # (T2V_WINO_GEMM_SK=2 still set)
            u = ops.pack_conv_weight(wd, desc, Cin)
            fw = [ops.conv2d_backward_data_winograd(desc, B, b, ws, Cin, u, forward_weights=True) for b in range(B)]
            assert all(torch.equal(a, c) for a, c in zip(fw, outs["2"]))
            t2v_env("T2V_WINO_GEMM_SK", "1")      # fewer tiles than blocks: not offered, and refused loudly
            assert not ops.backward_data_winograd_takes_forward_weights(desc, Cin, Cout)
            with pytest.raises(RuntimeError, match="takes_forward_weights"):
                ops.conv2d_backward_data_winograd(desc, B, 0, ws, Cin, u, forward_weights=True)


@pytest.mark.parametrize("npix,C,cs", [(512 * 512, 3, 4), (257 * 257 * 2, 64, 64), (35 * 35, 1, 4), (1000, 130, 132),
                                       (77, 512, 512), (999, 5, 5), (1, 8, 8)])
def test_channel_sum_matches_a_float64_sum(npix, C, cs):
    """the bias gradient's per-channel sum over pixels (16-byte loads where the channel storage allows, the 4-byte form
    otherwise): against a float64 sum, for heads (3 of 4 channels), a discriminator's first layer, a single channel, channel
    counts that are no multiple of 4 and a single pixel"""
    from text2video_amd import ops
    g = torch.Generator().manual_seed(npix % 97 + C)
    x = torch.randn(npix, cs, generator=g) + 0.25
    got = ops.channel_sum(x.cuda(), C).cpu().double()
    want = x.double().sum(0)[:C]
    assert got.shape == (C,)
    assert (got - want).abs().max().item() <= 2e-6 * x.double().abs().sum(0)[:C].max().item() + 1e-6


@pytest.mark.parametrize("relu,affine", [(1, False), (0, True), (2, True)], ids=["relu", "affine", "lrelu+affine"])
def test_dy_transform_with_the_norm_backward_inside_equals_apply_then_transform(relu, affine):
    """A dy A^T of the gradient in FRONT of a norm layer, formed inside the transform from the gradient behind it, the conv
    output and the norm backward's two sums (t2v_conv2d_backward_weight_winograd_dy_norm) -- bit for bit the transform of
    the tensor instance_norm_backward writes out; a 36 x 20 map (ragged tile grid), slot 1 of 2."""
    from text2video_amd import ops
    g = torch.Generator().manual_seed(5)
    H, W, C = 36, 20, 64
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT)
    c = torch.randn(H, W, C, generator=g).cuda()
    dy = torch.randn(H, W, C, generator=g).cuda()
    mr = torch.stack([c.mean((0, 1)), 1.0 / torch.sqrt(c.var((0, 1), unbiased=False) + 1e-5)], 1).contiguous()
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).cuda() if affine else None
    beta = (0.2 * torch.randn(C, generator=g)).cuda() if affine else None
    dc, sums = ops.instance_norm_backward(c, dy, mr, gamma, beta, relu)
    none, sums2 = ops.instance_norm_backward(c, dy, mr, gamma, beta, relu, sums_only=True)
    assert none is None and torch.equal(sums, sums2)
    ws = [ops.backward_weight_winograd_workspace(desc, C, 2, "cuda:0").fill_(float("nan")) for _ in range(2)]
    ops.conv2d_backward_weight_winograd_dy(dc, desc, ws[0], 2, 1, C)
    ops.conv2d_backward_weight_winograd_dy_norm(c, dy, mr, gamma, beta, relu, sums2, desc, ws[1], 2, 1, C)
    tp = ops.winograd_tile_rows(desc)
    md = [w[36 * 2 * tp * C:36 * 2 * tp * C * 2].view(36, 2, tp, C) for w in ws]
    assert torch.equal(md[0][:, 1], md[1][:, 1]) and bool(torch.isfinite(md[1][:, 1]).all())
    assert bool(torch.isnan(md[1][:, 0]).all())              # the other slot untouched
    assert md[1][:, 1].abs().max().item() > 0.1


def test_data_gradient_reads_the_forward_weights_in_place_at_the_generator_bottleneck():
    """1024 -> 1024 at 64x64 (the ResnetBlock conv of a 512x512 frame; 576 tiles on 512 blocks: accumulator hand-overs in
    play): the data gradient with the forward packing as its [K][N] operand is bit for bit the one with the transposed copy."""
    from text2video_amd import ops
    g = torch.Generator().manual_seed(11)
    H = W = 64
    C = 1024
    desc = ops.with_algo(ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT), ops.ALGO_WINOGRAD_F4)
    assert ops.backward_data_winograd_takes_forward_weights(desc, C, C)
    xs = torch.randn(1, H, W, C, generator=g).cuda()
    dys = torch.randn(1, H, W, C, generator=g).cuda()
    wd = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
    ws = ops.backward_weight_winograd_workspace(desc, C, 1, "cuda:0")
    ops.conv2d_backward_weight_winograd_stages(xs, dys, desc, ws, 1, 0, False)
    want = ops.conv2d_backward_data_winograd(desc, 1, 0, ws, C, ops.pack_conv_weight_transposed(wd, desc, C))
    u = ops.pack_conv_weight(wd, desc, C)
    for rep in range(3):
        got = ops.conv2d_backward_data_winograd(desc, 1, 0, ws, C, u, forward_weights=True)
        assert torch.equal(got, want), "launch %d: %d of %d differ" % (rep, int((got != want).sum()), got.numel())
    assert bool(torch.isfinite(want).all()) and want.abs().max().item() > 1.0


@pytest.mark.parametrize("case", [
    ("down 3x3 s2 64->128 @32x64", (32, 64, 64, 128, 3, 2, 1, 0, False)),
    ("up convT 3x3 s2 128->64 @16x16", (16, 16, 128, 64, 3, 2, 1, 0, True)),
    ("4x4 s2 p2 64->128 @62x62 (32x32 out)", (62, 62, 64, 128, 4, 2, 2, 0, False)),
], ids=lambda c: c[0] if isinstance(c, tuple) and isinstance(c[0], str) else None)
def test_weight_gradient_over_two_separate_buffers_equals_the_contiguous_batch(case):
    """t2v_conv2d_backward_weight_strided (ABI 14): the two frames of a clip reduced in ONE launch from buffers of their own
    (any distance apart, either order in memory) -- bit for bit the weight gradient of the contiguous batch of two, and
    within rounding of the sum of the two single-image gradients the train step used to launch."""
    from text2video_amd import ops
    _, (H, W, Cin, Cout, k, st, pad, pm, tr) = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr)
    ho, wo = ops.conv_out_dims(desc)
    xcs, ycs = ops.round_up(Cin, 4), ops.round_up(Cout, 4)
    assert ops.backward_weight_strided_supported(desc, xcs, ycs)
    x = torch.randn(2, H, W, xcs, generator=g).to(dev)
    dy = torch.randn(2, ho, wo, ycs, generator=g).to(dev)
    unpack = lambda packed: ops.unpack_conv_weight(packed, desc, xcs)      # noqa: E731 -- (padding rows of the packed layout hold junk)
    want = unpack(ops.conv2d_backward_weight(x, dy, desc))
    # separate allocations with junk in between; second image BELOW the first in memory for one of the operands
    pad_x, pad_y = torch.full((12345,), float("nan"), device=dev), torch.full((777,), float("nan"), device=dev)
    x1, x0 = x[1].clone(), x[0].clone()
    dy0, dy1 = dy[0].clone(), dy[1].clone()
    assert x0.data_ptr() != x1.data_ptr() and pad_x.numel() and pad_y.numel()
    got = unpack(ops.conv2d_backward_weight_pair(x0, dy0, x1, dy1, desc))
    assert torch.equal(got, want)
    one = unpack(ops.conv2d_backward_weight(x[0:1], dy[0:1], desc)) + unpack(ops.conv2d_backward_weight(x[1:2], dy[1:2], desc))
    assert (got - one).abs().max().item() <= 1e-4 * one.abs().max().item()
    acc = ops.conv2d_backward_weight(x, dy, desc)
    ops.conv2d_backward_weight_pair(x0, dy0, x1, dy1, desc, accumulate_into=acc)
    assert (unpack(acc) - 2 * want).abs().max().item() <= 1e-5 * want.abs().max().item()
    # shapes that keep their images contiguous say so, and the entry point refuses them
    stem = ops.conv_desc(32, 32, 9, 64, 7, 1, 3, 1, False)
    assert not ops.backward_weight_strided_supported(stem, 12, 64)
    with pytest.raises(RuntimeError, match="strided"):
        ops.conv2d_backward_weight_pair(torch.zeros(32, 32, 12, device=dev), torch.zeros(32, 32, 64, device=dev),
                                        torch.zeros(32, 32, 12, device=dev), torch.zeros(32, 32, 64, device=dev), stem)
