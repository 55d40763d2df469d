"""GPU parity tests, operator level: every HIP kernel behind the C ABI vs a plain torch fp32 CPU
reference of the same op (the semantics pinned in oracle/generator_ref.py's header).
Tolerances: fp32 MFMA is an exact fp32 fma chain, only the summation order differs from the
CPU reference => |delta| <= 1e-4 absolute on O(1) activations (north_star: 1e-3 end to end)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "GPU test without a GPU"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def _to_nhwc(x_chw, cs=None):
    from text2video_amd import ops
    return ops.nchw_to_nhwc(x_chw.to(_dev()).contiguous(), cs)


def _from_nhwc(y_hwc, C):
    return y_hwc[..., :C].permute(2, 0, 1).cpu()


def _ref_conv(x, w, b, k, stride, pad, pad_mode, transposed):
    x = x.unsqueeze(0)
    if transposed:
        return F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)[0]
    if pad_mode == 1 and pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)[0]


CONV_CASES = [
    # name, H, W, Cin, Cout, k, stride, pad, pad_mode(1=reflect), transposed
    ("rb3x3_fast", 16, 16, 64, 128, 3, 1, 1, 1, False),
    ("rb3x3_ragged_M", 20, 12, 32, 128, 3, 1, 1, 1, False),
    ("rb3x3_Ntail160", 8, 8, 32, 160, 3, 1, 1, 1, False),
    ("rb3x3_Cout8", 8, 8, 32, 8, 3, 1, 1, 1, False),
    ("stem7x7_cin9", 24, 24, 9, 128, 7, 1, 3, 1, False),
    ("stem7x7_cin6", 16, 20, 6, 16, 7, 1, 3, 1, False),
    # sizes / widths the halo-in-LDS stem kernel (conv_stem.hip) takes when norm statistics are requested
    ("stem_kernel_cin9_c128", 32, 48, 9, 128, 7, 1, 3, 1, False),
    ("stem_kernel_cin6_c128", 16, 16, 6, 128, 7, 1, 3, 1, False),
    ("stem_kernel_cin9_c64", 48, 32, 9, 64, 7, 1, 3, 1, False),
    ("stem_kernel_cin6_c64", 64, 16, 6, 64, 7, 1, 3, 1, False),
    ("stem_kernel_ragged_cin9", 40, 52, 9, 128, 7, 1, 3, 1, False),        # 3 x 4 tiles, ragged bottom row and right column
    ("stem_kernel_ragged_cin6", 17, 85, 6, 64, 7, 1, 3, 1, False),
    ("down3x3_s2", 32, 32, 32, 64, 3, 2, 1, 0, False),
    ("down3x3_s2_narrow", 16, 24, 8, 16, 3, 2, 1, 0, False),
    ("convT_fast", 8, 8, 64, 32, 3, 2, 1, 0, True),
    ("convT_narrow", 8, 12, 16, 8, 3, 2, 1, 0, True),
    ("convT_ragged", 10, 6, 32, 32, 3, 2, 1, 0, True),
    ("disc4x4_s2_p2", 16, 16, 8, 64, 4, 2, 2, 0, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_with_norm_stats(case):
    from text2video_amd import ops
    name, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed = case
    x = _rand(Cin, H, W, seed=1)
    w = _rand(*((Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)), seed=2, scale=0.1)
    b = _rand(Cout, seed=3, scale=0.1)
    ref = _ref_conv(x, w, b, k, stride, pad, pad_mode, transposed)
    desc = ops.conv_desc(H, W, Cin, Cout, k, stride, pad, pad_mode, transposed)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, xs.shape[-1])
    bd = b.to(_dev())
    # plain conv
    y = ops.conv2d(xs, pw, bd, desc)
    got = _from_nhwc(y, Cout)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    if y.shape[-1] > Cout:  # padded output channels are written as zeros
        assert y[..., Cout:].abs().max().item() == 0.0
    if Cout % 4 == 0 and Cout > 16:
        # conv + fused instance-norm statistics + apply(ReLU)
        ref_n = F.relu(F.instance_norm(ref.unsqueeze(0), eps=1e-5))[0]
        yn = ops.conv_norm_act(xs, pw, bd, desc, relu=True)
        assert (_from_nhwc(yn, Cout) - ref_n).abs().max().item() <= 2e-4


@pytest.mark.parametrize("act", ["tanh", "flow_w"])
def test_head_conv_small_cout(act):
    from text2video_amd import ops
    H, W, Cin = 24, 20, 32
    x = _rand(Cin, H, W, seed=4)
    w = _rand(3, Cin, 7, 7, seed=5, scale=0.05)
    b = _rand(3, seed=6, scale=0.1)
    ref = _ref_conv(x, w, b, 7, 1, 3, 1, False)
    if act == "tanh":
        ref = torch.tanh(ref)
        desc = ops.conv_desc(H, W, Cin, 3, 7, 1, 3, ops.PAD_REFLECT, False, ops.ACT_TANH)
    else:
        ref = torch.cat([ref[:2] * 40.0, torch.sigmoid(ref[2:3])], 0)
        desc = ops.conv_desc(H, W, Cin, 3, 7, 1, 3, ops.PAD_REFLECT, False, ops.ACT_FLOW_W, 40.0)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    y = ops.conv2d(xs, pw, b.to(_dev()), desc, y_cs=4)
    assert (_from_nhwc(y, 3) - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert y[..., 3].abs().max().item() == 0.0


@pytest.mark.parametrize("geom", [(66, 66, 512, 1), (34, 35, 512, 1), (9, 13, 256, 1), (3, 3, 512, 1), (18, 14, 256, 2)],
                         ids=lambda g: "%dx%dx%d_s%d" % g)
@pytest.mark.parametrize("lrelu", [False, True], ids=["linear", "lrelu"])
def test_single_output_channel_conv(geom, lrelu):
    """Conv2d(C, 1, 4, padding=2): the PatchGAN discriminators' last layer runs on the wave-per-pixel kernel
    (conv_cout1_kernel) instead of the implicit GEMM; every tap position relative to the zero border is covered."""
    from text2video_amd import ops
    H, W, Cin, stride = geom
    x = _rand(Cin, H, W, seed=11)
    w = _rand(1, Cin, 4, 4, seed=12, scale=0.02)
    b = _rand(1, seed=13, scale=0.1)
    ref = _ref_conv(x, w, b, 4, stride, 2, 0, False)
    act, slope = (ops.ACT_LRELU, 0.2) if lrelu else (ops.ACT_NONE, 1.0)
    if lrelu:
        ref = F.leaky_relu(ref, 0.2)
    desc = ops.conv_desc(H, W, Cin, 1, 4, stride, 2, ops.PAD_ZERO, False, act, slope)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    y = torch.full(tuple(ref.shape[1:]) + (4,), float("nan"), device=_dev())
    ops.conv2d(xs, pw, b.to(_dev()), desc, y_cs=4, out=y)
    assert (_from_nhwc(y, 1) - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert y[..., 1:].abs().max().item() == 0.0


def test_instance_norm_affine_residual_matches_trainmode_batchnorm():
    """BN(train, N=1) == IN + affine; apply adds two residuals after the ReLU-less norm."""
    from text2video_amd import ops
    H, W, Cin, C = 12, 12, 32, 64
    x = _rand(Cin, H, W, seed=7)
    w = _rand(C, Cin, 3, 3, seed=8, scale=0.1)
    b = _rand(C, seed=9)
    g = 1.0 + _rand(C, seed=10, scale=0.1)
    bt = _rand(C, seed=11, scale=0.1)
    r1 = _rand(C, H, W, seed=12)
    r2 = _rand(C, H, W, seed=13)
    conv = _ref_conv(x, w, b, 3, 1, 1, 1, False)
    bn = torch.nn.BatchNorm2d(C, affine=True)
    bn.train()
    with torch.no_grad():
        bn.weight.copy_(g)
        bn.bias.copy_(bt)
        ref = bn(conv.unsqueeze(0))[0] + r1 + r2
    desc = ops.conv_desc(H, W, Cin, C, 3, 1, 1, ops.PAD_REFLECT)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    y = ops.conv_norm_act(xs, pw, b.to(_dev()), desc, gamma=g.to(_dev()), beta=bt.to(_dev()), relu=False,
                          res1=_to_nhwc(r1), res2=_to_nhwc(r2))
    assert (_from_nhwc(y, C) - ref).abs().max().item() <= 2e-4


def test_instance_norm_large_mean_is_stable():
    """Chan-merged two-pass statistics: a large per-channel offset must not destroy the variance."""
    from text2video_amd import ops
    H, W, Cin, C = 32, 32, 32, 64
    x = _rand(Cin, H, W, seed=14)
    w = _rand(C, Cin, 3, 3, seed=15, scale=0.05)
    b = torch.full((C,), 300.0)
    conv = _ref_conv(x.double(), w.double(), b.double(), 3, 1, 1, 1, False)
    ref = F.instance_norm(conv.unsqueeze(0), eps=1e-5)[0].float()
    desc = ops.conv_desc(H, W, Cin, C, 3, 1, 1, ops.PAD_REFLECT)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    y = ops.conv_norm_act(xs, pw, b.to(_dev()), desc, relu=False)
    assert (_from_nhwc(y, C) - ref).abs().max().item() <= 2e-3  # fp32 conv output at |x|~300 carries ~3e-5 abs error


def test_flow_warp_composite_vs_grid_sample():
    from oracle.generator_ref import resample
    from text2video_amd import ops
    H, W = 40, 56
    prev = _rand(6, H, W, seed=16).clamp(-1, 1)
    raw = torch.tanh(_rand(3, H, W, seed=17))
    flow = _rand(2, H, W, seed=18, scale=6.0)   # large flows exercise the border clamp
    wgt = torch.sigmoid(_rand(1, H, W, seed=19))
    warp_ref = resample(prev[None, -3:], flow[None])[0]
    ref = raw * wgt + warp_ref * (1 - wgt)
    fw = _to_nhwc(torch.cat([flow, wgt], 0))
    out, warp = ops.flow_warp_composite(_to_nhwc(raw), fw, _to_nhwc(prev), 3, want_warp=True)
    assert (_from_nhwc(warp, 3) - warp_ref).abs().max().item() <= 1e-4
    assert (_from_nhwc(out, 3) - ref).abs().max().item() <= 1e-4


def test_flow_warp_identity_returns_input():
    """Sanity anchor derivable from the reference: zero flow, corner aligned => warp(x) == x."""
    from text2video_amd import ops
    H, W = 17, 33
    prev = _rand(6, H, W, seed=20)
    z = torch.zeros(3, H, W)
    out, warp = ops.flow_warp_composite(_to_nhwc(z), _to_nhwc(z), _to_nhwc(prev), 3, want_warp=True)
    assert (_from_nhwc(warp, 3) - prev[-3:]).abs().max().item() <= 2e-5


def test_avgpool_count_include_pad_false():
    from text2video_amd import ops
    for (H, W, C) in [(16, 16, 12), (18, 14, 8), (7, 9, 3)]:
        x = _rand(C, H, W, seed=21)
        ref = F.avg_pool2d(x[None], 3, 2, 1, count_include_pad=False)[0]
        y = ops.avgpool3x3s2(x.permute(1, 2, 0).contiguous().to(_dev()))
        assert (y.permute(2, 0, 1).cpu() - ref).abs().max().item() <= 1e-6


def test_pose_u8_and_tensor2im_roundtrip():
    from text2video_amd import ops
    rng = np.random.default_rng(22)
    img = torch.from_numpy(rng.integers(0, 256, size=(20, 24, 3), dtype=np.uint8))
    dst = torch.zeros(20, 24, 12, device=_dev())
    ops.pose_u8_to_f32(img.to(_dev()), dst, 3)
    ref = (img.float() / 255.0 - 0.5) / 0.5
    assert (dst[..., 3:6].cpu() - ref).abs().max().item() == 0.0
    assert dst[..., :3].abs().max().item() == 0.0 and dst[..., 6:].abs().max().item() == 0.0
    u8 = ops.tensor2im_u8(dst[..., 3:6].contiguous()).cpu()
    ref_u8 = ((ref + 1) / 2.0 * 255.0).clamp(0, 255).to(torch.uint8)
    assert (u8.int() - ref_u8.int()).abs().max().item() == 0
    # pinned: ToTensor + Normalize(.5,.5) as the reference's vendored torchvision computes them
    # (tests/golden/make_transforms_golden.py imports $SP/torchvision/transforms/functional.py:38-60,185-208)
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz"))
    for src, want in ((g["img"], g["img_norm"]), (np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2), g["all_values_norm"])):
        d = torch.zeros(src.shape[0], src.shape[1], 4, device=_dev())
        ops.pose_u8_to_f32(torch.from_numpy(src).to(_dev()), d, 0)
        assert torch.equal(d[..., :3].permute(2, 0, 1).cpu(), torch.from_numpy(want))


def test_fullsize_resblock_conv_1024():
    """The kernel that is 84 % of the FLOPs at its real size: 1024->1024 3x3 reflect @64x64."""
    from text2video_amd import ops
    H = W = 64
    C = 1024
    x = _rand(C, H, W, seed=23)
    w = _rand(C, C, 3, 3, seed=24, scale=0.02)
    b = _rand(C, seed=25, scale=0.1)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = _ref_conv(x, w, b, 3, 1, 1, 1, False)
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT)
    xs = _to_nhwc(x)
    pw = ops.pack_conv_weight(w.to(_dev()), desc, C)
    y = ops.conv2d(xs, pw, b.to(_dev()), desc)
    err = (_from_nhwc(y, C) - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err
    # linearity (size-independent property): conv(2x) - bias == 2*(conv(x) - bias)
    y2 = ops.conv2d(xs * 2, pw, b.to(_dev()), desc)
    lin = ((y2 - b.to(_dev())) - 2 * (y - b.to(_dev()))).abs().max().item()
    assert lin <= 1e-4


REPEAT_CASES = [
    # name, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed, stats, launches
    ("head_7x7_cout3_fullsize", 512, 512, 128, 3, 7, 1, 3, 1, False, False, 60),   # small-Cout tile, 3-slot ring
    ("resblock_1024_fullsize", 64, 64, 1024, 1024, 3, 1, 1, 1, False, True, 30),   # 128x128 tile, 3-slot ring
    ("convT_256_fullsize", 256, 256, 256, 128, 3, 2, 1, 0, True, True, 20),        # 4 phases, 2-slot ring
    ("stem_7x7_cin9_fullsize", 512, 512, 9, 128, 7, 1, 3, 1, False, True, 20),     # MODE 1 loader
    ("resblock_64x40_quarter_tiles", 64, 40, 1024, 1024, 3, 1, 1, 1, False, True, 30),  # 64x64 tiles
]


@pytest.mark.parametrize("case", REPEAT_CASES, ids=[c[0] for c in REPEAT_CASES])
def test_conv_bitwise_repeatable_race_screen(case):
    """Race screen for the LDS-DMA pipeline (counted vmcnt + raw barriers): the same launch must
    reproduce its first result bit for bit, output and norm partials, on NaN-poisoned buffers.
    (Caught a real bug: loader waves issuing fewer DMA instructions than the vmcnt count assumed.)"""
    from text2video_amd import ops
    name, H, W, Cin, Cout, k, stride, pad, pad_mode, transposed, stats, launches = case
    dev = _dev()
    desc = ops.conv_desc(H, W, Cin, Cout, k, stride, pad, pad_mode, transposed,
                         ops.ACT_TANH if Cout == 3 else ops.ACT_NONE)
    xcs = ops.round_up(Cin, 4)
    x = _rand(H, W, xcs, seed=31).to(dev)
    w = _rand(*((Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)), seed=32, scale=0.02).to(dev)
    pw = ops.pack_conv_weight(w, desc, xcs)
    b = _rand(Cout, seed=33).to(dev)
    ho, wo = ops.conv_out_dims(desc)
    ycs = Cout if Cout % 4 == 0 else 4
    sb = ops.conv_stats_buffer(desc, dev) if stats else None
    ref = ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb).clone()
    ref_s = sb.clone() if stats else None
    assert torch.isfinite(ref).all()
    for i in range(launches):
        y = torch.full((ho, wo, ycs), float("nan"), device=dev)
        if stats:
            sb.fill_(float("nan"))
        ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y)
        assert torch.equal(y, ref), "launch %d differs" % i
        if stats:
            assert torch.equal(sb, ref_s), "launch %d: norm partials differ" % i


@pytest.mark.parametrize("case", [(1, 32, 16, 32, 64), (1, 64, 64, 64, 96), (1, 16, 64, 128, 32),
                                  (2, 64, 32, 32, 64), (2, 64, 64, 64, 96), (2, 32, 128, 128, 32), (2, 128, 128, 32, 32),
                                  # ragged tile grids: odd sizes, sizes that are no multiple of the output tile,
                                  # tile counts that need zero-tile padding to 128 (the real fadg0 geometries
                                  # 512x680 / 512x320 have 64x85 / 64x40 bottlenecks)
                                  (1, 64, 85, 32, 64), (2, 64, 85, 32, 64), (2, 64, 40, 64, 32), (1, 9, 7, 32, 32),
                                  (2, 9, 7, 32, 32), (2, 2, 2, 32, 32), (1, 3, 2, 32, 4), (2, 34, 130, 32, 32)],
                         ids=["F2-T128", "F2-T1024", "F2-T512", "F4-T128", "F4-T256", "F4-T256w", "F4-T1024",
                              "F2-64x85", "F4-64x85", "F4-64x40", "F2-9x7", "F4-9x7", "F4-2x2", "F2-3x2", "F4-34x130"])
def test_winograd_conv_matches_direct_and_reference(case):
    """F(2x2,3x3) / F(4x4,3x3): input transform -> 16 | 36 grouped GEMMs -> output transform (+bias, +norm statistics)."""
    from text2video_amd import ops
    algo, H, W, Cin, Cout = case
    npos = 16 if algo == 1 else 36
    x = _rand(Cin, H, W, seed=41)
    w = _rand(Cout, Cin, 3, 3, seed=42, scale=0.1)
    b = _rand(Cout, seed=43, scale=0.1)
    ref = _ref_conv(x, w, b, 3, 1, 1, 1, False)
    desc_w = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT, algo=algo)
    assert ops.winograd_supported(desc_w, Cin)
    xs = _to_nhwc(x)
    U = ops.pack_conv_weight(w.to(_dev()), desc_w, Cin)
    assert U.numel() == npos * ((Cout + 127) // 128 * 128) * Cin
    stats = ops.conv_stats_buffer(desc_w, _dev())
    y = ops.conv2d_winograd(xs, U, b.to(_dev()), desc_w, stats=stats)
    assert (_from_nhwc(y, Cout) - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # staged execution (what bench.py times) gives the same bits
    ws = ops.winograd_workspace(desc_w, Cin, _dev())
    y2 = torch.empty_like(y)
    for st in (1, 2, 4):
        ops.conv2d_winograd(xs, U, b.to(_dev()), desc_w, stats=stats, out=y2, workspace=ws, stages=st)
    assert torch.equal(y, y2)
    # the statistics it emits feed the same finalize/apply as the direct kernel's
    mr = ops.instance_norm_finalize(stats, desc_w)
    yn = ops.instance_norm_apply(y, mr, relu=True)
    ref_n = F.relu(F.instance_norm(ref.unsqueeze(0), eps=1e-5))[0]
    assert (_from_nhwc(yn, Cout) - ref_n).abs().max().item() <= 3e-4
    # unsupported shapes are refused, not silently run
    bad = ops.conv_desc(10, 1, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT, algo=algo)   # reflection needs >= 2 pixels
    assert not ops.winograd_supported(bad, Cin)
    with pytest.raises(RuntimeError):
        ops.pack_conv_weight(w.to(_dev()), bad, Cin)


@pytest.mark.parametrize("algo", [1, 2], ids=["F2", "F4"])
@pytest.mark.parametrize("pad", [0, 1, 2])
def test_winograd_zero_padding(algo, pad):
    """Zero padding 0..2 (pad 2 = the data gradient of the pad-1 ResnetBlock conv: (H+2) x (W+2) outputs)."""
    from text2video_amd import ops
    H, W, Cin, Cout = 18, 30, 32, 64
    x = _rand(Cin, H, W, seed=51)
    w = _rand(Cout, Cin, 3, 3, seed=52, scale=0.1)
    b = _rand(Cout, seed=53, scale=0.1)
    ref = F.conv2d(x.unsqueeze(0), w, b, padding=pad)[0]
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, pad, ops.PAD_ZERO, algo=algo)
    assert ops.winograd_supported(desc, Cin)
    U = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    stats = ops.conv_stats_buffer(desc, _dev())
    y = ops.conv2d_auto(_to_nhwc(x), U, b.to(_dev()), desc, stats=stats)
    assert tuple(y.shape) == (H + 2 * pad - 2, W + 2 * pad - 2, Cout)
    assert (_from_nhwc(y, Cout) - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    yn = ops.instance_norm_apply(y, ops.instance_norm_finalize(stats, desc))
    assert (_from_nhwc(yn, Cout) - F.instance_norm(ref.unsqueeze(0), eps=1e-5)[0]).abs().max().item() <= 3e-4
    # the direct kernel agrees, and best_conv_algo never proposes what is not supported
    yd = ops.conv2d(_to_nhwc(x), ops.pack_conv_weight(w.to(_dev()), ops.with_algo(desc, 0), Cin), b.to(_dev()), ops.with_algo(desc, 0))
    assert (y - yd).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    assert ops.best_conv_algo(ops.conv_desc(H, W, Cin, Cout, 3, 2, 1, ops.PAD_ZERO), Cin) == ops.ALGO_DIRECT
    assert ops.best_conv_algo(ops.conv_desc(4, 4, Cin, Cout, 3, 1, 1, ops.PAD_ZERO), Cin) == ops.ALGO_DIRECT


@pytest.mark.parametrize("slope", [0.0, 0.2], ids=["relu", "lrelu"])
def test_winograd_f4_output_transform_applies_the_activation(slope):
    """Convs without a norm (VGG19 loss network: conv + ReLU) keep their activation on the F(4x4,3x3) path; F(2x2)
    and activation + statistics are refused."""
    from text2video_amd import ops
    H, W, Cin, Cout = 20, 36, 64, 64
    x = _rand(Cin, H, W, seed=61)
    w = _rand(Cout, Cin, 3, 3, seed=62, scale=0.1)
    b = _rand(Cout, seed=63, scale=0.1)
    ref = F.leaky_relu(F.conv2d(x.unsqueeze(0), w, b, padding=1)[0], slope)
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_ZERO, False, ops.ACT_LRELU, slope, algo=ops.ALGO_WINOGRAD_F4)
    assert ops.winograd_supported(desc, Cin)
    assert ops.best_conv_algo(ops.with_algo(desc, 0), Cin) == ops.ALGO_WINOGRAD_F4
    assert not ops.winograd_supported(ops.with_algo(desc, ops.ALGO_WINOGRAD), Cin)
    U = ops.pack_conv_weight(w.to(_dev()), desc, Cin)
    y = ops.conv2d_auto(_to_nhwc(x), U, b.to(_dev()), desc)
    assert (_from_nhwc(y, Cout) - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    with pytest.raises(RuntimeError):
        ops.conv2d_auto(_to_nhwc(x), U, b.to(_dev()), desc, stats=ops.conv_stats_buffer(desc, _dev()))


_FG_GEOMS = [(64, 64, 1024, 1024), (64, 88, 640, 640), (128, 128, 256, 256), (64, 128, 512, 1024), (128, 128, 512, 256),
             (64, 40, 1024, 1024), (64, 40, 512, 384), (128, 128, 1024, 1024), (64, 85, 1024, 1024), (64, 56, 1024, 1024),
             (64, 114, 1024, 1024), (60, 52, 256, 384)]
# too few tiles for one block per CU: T2V_WINO_GEMM_SK_RAGGED=2 stays on the two-per-CU ragged form there (= the "1" case)
_FG_NO_TALL = {(64, 40, 1024, 1024), (64, 40, 512, 384), (60, 52, 256, 384)}
_FG_NAMES = {"1": "ragged", "0": "whole_tiles", "2": "tall_ragged"}


def _fg_cases():
    """(geometry, T2V_WINO_GEMM_SK_RAGGED) pairs that differ: with whole 128-row tiles the ragged switch changes nothing (one
    case), the balanced one-block-per-CU tiles exist only where there are enough of them"""
    out = []
    for g in _FG_GEOMS:
        frags = -(-(-(-g[0] // 4) * -(-g[1] // 4)) // 32)
        for r in ("1", "0", "2"):
            if r == "1" or (frags % 4 != 0 and not (r == "2" and g in _FG_NO_TALL)):
                out.append(pytest.param(g, r, id="%dx%dx%dx%d-%s" % (g + (_FG_NAMES[r],))))
    return out


@pytest.mark.parametrize("geom,ragged", _fg_cases())
def test_fixed_grid_winograd_gemm_equals_tile_per_block(geom, ragged, t2v_env):
    """The batched Winograd GEMM on a fixed grid (conv_igemm.hip: wino_gemm_sk_kernel; tiles cut between two blocks are
    finished from the first block's accumulators) against one block per tile: the same K-ordered MFMA chain per output,
    so the conv must be BIT-identical -- also launch after launch on one workspace (a stale hand-over flag or a stale L2
    line of an earlier launch would show as a differing frame).  Tile counts not divisible by the 8 XCDs included, and the
    192 x 64 tiles of the 512x320 frames (64 x 40 maps: 160 tile rows padded to 192), and the second schedule (4.5 rounds of
    tiles at 128 x 128 x 1024: whole rounds + a half round cut in two).  `ragged`: tile rows that are no whole 128s (the
    reference's 512x680 frames: 64 x 85 maps, 352 rows = 4 + 4 + 3 fragments; the 16:9 speakers: 64 x 114 -> 464 rows, 64 x 56
    -> 224; 60 x 52 -> 195 rows, the last fragment part padding) on wino_gemm_skr_kernel's ragged M tiles, or with
    T2V_WINO_GEMM_SK_RAGGED=0 padded to whole tiles as before, or with =2 on wino_gemm_skt_kernel's balanced tiles of 3..6
    fragments, one block per CU (352 rows = 6 + 5 fragments, 464 = 5 + 5 + 5, 224 = 4 + 3, 195 = 4 + 3)."""
    from text2video_amd import ops
    H, W, Cin, Cout = geom
    rows = -(-H // 4) * -(-W // 4)
    t2v_env("T2V_WINO_GEMM_SK_RAGGED", ragged)
    assert ops.fixed_grid_enabled()
    if ragged == "2":
        t2v_env("T2V_WINO_GEMM_SK", "2")
        form = ops.winograd_gemm_form(ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4))
        assert "skt_kernel" in form, form        # (the geometries without that form are not in the list: _FG_NO_TALL)
    dev = _dev()
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.03).to(dev)
    b = _rand(Cout, seed=3).to(dev)
    pu = ops.pack_conv_weight(w, desc, Cin)
    ws = ops.winograd_workspace(desc, Cin, dev)
    xs = [_rand(H, W, Cin, seed=10 + i).to(dev) for i in range(3)]
    t2v_env("T2V_WINO_GEMM_SK", "0")
    want = [ops.conv2d_winograd(x, pu, b, desc, workspace=ws).clone() for x in xs]
    t2v_env("T2V_WINO_GEMM_SK", "2")      # wherever the shape allows, not only where it is faster
    ws.fill_(float("nan"))
    for rep in range(12):
        x, y0 = xs[rep % 3], want[rep % 3]
        y = ops.conv2d_winograd(x, pu, b, desc, workspace=ws)
        assert torch.equal(y, y0), "launch %d: %d of %d outputs differ" % (rep, int((y != y0).sum()), y.numel())
    # and against the fp32 reference of the conv (the tolerance of the other Winograd tests)
    ref = _ref_conv(xs[0].cpu().permute(2, 0, 1), w.cpu(), b.cpu(), 3, 1, 1, 1, False)
    assert float((_from_nhwc(want[0], Cout) - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max()))


def test_overlap_hint_selects_the_one_block_per_cu_form_and_keeps_the_bits(t2v_env):
    """t2v_set_overlap_hint (ABI 13): a caller that runs a second stream beside its launches -- t2v_generator_forward's
    two-stream frames do it themselves -- gets the ResnetBlock GEMM stage of TWO 512x512 images in lock-step (of one with
    T2V_OVERLAP_HINT_SINGLE=1) on 256 x 128 tiles with one block per CU instead of 128 x 128 with two.  Which kernel runs changes, the K-ordered MFMA chain of an output does not: same bits.
    T2V_OVERLAP_HINT=0 ignores the hint; the hint is per thread and returns its previous value."""
    from text2video_amd import ops
    H, W, C = 64, 64, 1024
    dev = _dev()
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    w = _rand(C, C, 3, 3, seed=2, scale=0.03).to(dev)
    b = _rand(C, seed=3).to(dev)
    pu = ops.pack_conv_weight(w, desc, C)
    ws = ops.winograd_workspace(desc, C, dev)
    x = _rand(H, W, C, seed=21).to(dev)
    assert ops.set_overlap_hint(False) == 0
    assert "128x128" in ops.winograd_gemm_form(desc)
    want = ops.conv2d_winograd(x, pu, b, desc, workspace=ws).clone()
    # hint 2 (ABI 18: the second stream runs fixed-grid GEMMs of its own -- a train step's weight gradients): the tile form
    # of no hint on ONE block per CU (half the grid); which block owns a tile changes, the bits do not
    assert ops.set_overlap_hint(2) == 0
    assert "128x128" in ops.winograd_gemm_form(desc)
    for rep in range(3):
        assert torch.equal(ops.conv2d_winograd(x, pu, b, desc, workspace=ws), want)
    assert ops.set_overlap_hint(True) == 2
    try:
        # one image keeps 128 x 128 tiles (round 6: its frame is no faster on the tall form any more), two in lock-step take it
        assert "128x128" in ops.winograd_gemm_form(desc), ops.winograd_gemm_form(desc)
        assert "256x128" in ops.winograd_gemm_form(desc, 2) and "128x128" in ops.winograd_gemm_form(desc, 4)
        t2v_env("T2V_OVERLAP_HINT_SINGLE", "1")      # (the rule of rounds 4-5)
        assert "256x128" in ops.winograd_gemm_form(desc), ops.winograd_gemm_form(desc)
        ws.fill_(float("nan"))
        for rep in range(4):
            assert torch.equal(ops.conv2d_winograd(x, pu, b, desc, workspace=ws), want)
        t2v_env("T2V_OVERLAP_HINT", "0")
        assert "128x128" in ops.winograd_gemm_form(desc)
    finally:
        assert ops.set_overlap_hint(False) == 1
    # (another thread never sees this thread's hint)
    import threading
    seen = []
    ops.set_overlap_hint(True)
    try:
        t = threading.Thread(target=lambda: seen.append(ops.set_overlap_hint(False)))
        t.start(); t.join()
    finally:
        ops.set_overlap_hint(False)
    assert seen == [0]


def test_fixed_grid_gemm_survives_graph_replay(t2v_env):
    """A captured launch is re-issued with the SAME kernel arguments, hand-over tag included: the consumer clears a tag it has
    taken, so a replay does not mistake the previous replay's accumulators for this one's.  Capture one Winograd conv
    (fixed-grid GEMM stage) in a HIP graph, replay it on three different inputs, compare with the eager results."""
    from text2video_amd import ops
    H, W, C = 64, 64, 1024
    dev = _dev()
    t2v_env("T2V_WINO_GEMM_SK", "2")
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    w = _rand(C, C, 3, 3, seed=5, scale=0.03).to(dev)
    b = _rand(C, seed=6).to(dev)
    pu = ops.pack_conv_weight(w, desc, C)
    ws = ops.winograd_workspace(desc, C, dev)
    xs = [_rand(H, W, C, seed=20 + i).to(dev) for i in range(3)]
    want = [ops.conv2d_winograd(x, pu, b, desc, workspace=ws).clone() for x in xs]
    x_static, y_static = xs[0].clone(), torch.empty_like(want[0])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.conv2d_winograd(x_static, pu, b, desc, workspace=ws, out=y_static)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.conv2d_winograd(x_static, pu, b, desc, workspace=ws, out=y_static)
    for rep in range(6):
        x_static.copy_(xs[rep % 3])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, want[rep % 3]), "replay %d: %d outputs differ" % (rep, int((y_static != want[rep % 3]).sum()))


def test_fixed_grid_kernels_race_screen_under_load():
    """scripts/stress_fixed_grid.py: fixed-grid GEMM stages and weight-gradient reductions on two streams with changing inputs
    and workspaces reused back to back, a third stream streaming 1 GiB at the same time -- every result bit-equal to one
    block per tile (the accumulator hand-over crosses XCDs: a stale line or a tag seen early would show under load first)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_fixed_grid.py"), "80"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 80 iterations differ" in r.stdout


def test_hand_over_timeout_is_reported_once_and_switches_to_one_block_per_tile():
    """The fixed-grid kernels' guards: the dispatch-order self-test of t2v_create passed on this box; a consumer
    wave whose producer's tag never shows up raises a sticky error word in pinned host memory (here raised through the test
    hook, exactly the store the wave makes): the NEXT entry point that could launch a fixed-grid kernel returns
    T2V_ERR_HANDOVER with a message, once, and from then on the process runs one block per tile -- same bits."""
    from text2video_amd import _lib, ops
    dev = _dev()
    H, W, C = 64, 64, 1024
    desc = ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    w = _rand(C, C, 3, 3, seed=5, scale=0.03).to(dev)
    b = _rand(C, seed=6).to(dev)
    x = _rand(H, W, C, seed=7).to(dev)
    pu = ops.pack_conv_weight(w, desc, C)
    ws = ops.winograd_workspace(desc, C, dev)
    lib = _lib.load()
    assert ops.fixed_grid_enabled(), "dispatch-order self-test failed on this box"
    want = ops.conv2d_winograd(x, pu, b, desc, workspace=ws).clone()      # fixed grid (576 tiles on 512 blocks)
    torch.cuda.synchronize()
    ops.check_async_errors()                                               # nothing pending
    try:
        lib.t2v_debug_async_error(1)
        with pytest.raises(RuntimeError, match="hand-over timed out"):
            ops.conv2d_winograd(x, pu, b, desc, workspace=ws)
        assert not ops.fixed_grid_enabled()
        ops.check_async_errors()                                           # reported once
        got = ops.conv2d_winograd(x, pu, b, desc, workspace=ws)            # one block per tile now
        assert torch.equal(got, want)
    finally:
        lib.t2v_debug_async_error(0)
    assert ops.fixed_grid_enabled()
    assert torch.equal(ops.conv2d_winograd(x, pu, b, desc, workspace=ws), want)


@pytest.mark.parametrize("geom", [(64, 64, 1024, 1024), (64, 128, 1024, 1024), (64, 64, 512, 1024)])
def test_one_block_per_cu_256x128_tiles_equal_tile_per_block(geom, t2v_env):
    """Under the overlap hint (ops.set_overlap_hint: what the generator's two-stream frames set): the 256 tile rows of a 512x512
    frame (512 of two) on 256 x 128 tiles, one block per CU
    (wino_gemm_sk_kernel<TileCfg<32,1,4,8,1>, ring 3>: 128 accumulator registers, 48 KiB stages) -- the same K-ordered MFMA chain
    per output as one block per tile: bit-identical, launch after launch on a NaN-filled workspace."""
    from text2video_amd import ops
    H, W, Cin, Cout = geom
    dev = _dev()
    desc = ops.conv_desc(H, W, Cin, Cout, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)
    w = _rand(Cout, Cin, 3, 3, seed=2, scale=0.03).to(dev)
    b = _rand(Cout, seed=3).to(dev)
    pu = ops.pack_conv_weight(w, desc, Cin)
    ws = ops.winograd_workspace(desc, Cin, dev)
    xs = [_rand(H, W, Cin, seed=20 + i).to(dev) for i in range(2)]
    t2v_env("T2V_WINO_GEMM_SK", "0")
    want = [ops.conv2d_winograd(x, pu, b, desc, workspace=ws).clone() for x in xs]
    t2v_env("T2V_WINO_GEMM_SK", "1")
    t2v_env("T2V_OVERLAP_HINT_SINGLE", "1")      # (one image's 256 rows on the tall form as well: the library's rule keeps it for two)
    prev = ops.set_overlap_hint(True)
    try:
        assert "256x128" in ops.winograd_gemm_form(desc), ops.winograd_gemm_form(desc)
        ws.fill_(float("nan"))
        for rep in range(8):
            y = ops.conv2d_winograd(xs[rep % 2], pu, b, desc, workspace=ws)
            assert torch.equal(y, want[rep % 2]), "launch %d: %d of %d outputs differ" % (rep, int((y != want[rep % 2]).sum()), y.numel())
    finally:
        ops.set_overlap_hint(prev)


@pytest.mark.parametrize("case", [
    # (H, W, Cin, Cout, k, stride, pad, pad_mode, transposed, stats, act, B)
    ("d1 129x129 64->128 k4 s2 + stats", (129, 129, 64, 128, 4, 2, 2, 0, False, True, 0, 4)),
    ("d3 33x33 256->512 k4 s1 + stats", (33, 33, 256, 512, 4, 1, 2, 0, False, True, 0, 3)),
    ("d0 64x64 6->64 k4 s2 lrelu", (64, 64, 6, 64, 4, 2, 2, 0, False, False, 3, 4)),
    ("dgrad of d2: convT k4 s2", (17, 17, 256, 128, 4, 2, 2, 0, True, False, 0, 2)),
    ("one output channel (dedicated kernel, image by image)", (20, 20, 512, 1, 4, 1, 2, 0, False, False, 0, 3)),
    ("7x7 head (dedicated kernel, image by image)", (32, 32, 64, 3, 7, 1, 3, 1, False, False, 1, 2)),
    ("7x7 stem + stats (dedicated kernel, image by image)", (32, 32, 9, 64, 7, 1, 3, 1, False, True, 0, 2)),
], ids=lambda c: c[0] if isinstance(c, tuple) and isinstance(c[0], str) else None)
def test_batched_direct_conv_equals_the_single_image_launches(case):
    """t2v_conv2d_forward_batch (ABI 14): B images as blockIdx.y of one implicit-GEMM launch -- what the train step's
    discriminator layers and their data gradients use.  Every image's output and statistics partials are the
    single-image call's, bit for bit."""
    from text2video_amd import ops
    _, (H, W, Cin, Cout, k, st, pad, pm, tr, stats, act, B) = case
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    desc = ops.conv_desc(H, W, Cin, Cout, k, st, pad, pm, tr, act, 0.2, output_padding=(0 if (tr and k == 4) else None))
    xcs = ops.round_up(Cin, 4)
    x = torch.zeros(B, H, W, xcs)
    x[..., :Cin] = torch.randn(B, H, W, Cin, generator=g)
    x = x.to(dev)
    w = (torch.randn(*((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)), generator=g) * 0.05).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    pw = ops.pack_conv_weight(w, desc, xcs)
    ycs = ops.round_up(Cout, 4)
    n = ops.conv_stats_buffer(desc, dev).numel() if stats else 0
    s_one = torch.zeros(B * n, device=dev) if stats else None
    y_one = torch.stack([ops.conv2d(x[i], pw, b, desc, y_cs=ycs, stats=s_one[i * n:(i + 1) * n] if stats else None) for i in range(B)])
    s_all = torch.full((B * n,), float("nan"), device=dev) if stats else None
    y_all = torch.full_like(y_one, float("nan"))
    ops.conv2d_batch(x, pw, b, desc, y_cs=ycs, stats=s_all, out=y_all)
    assert torch.equal(y_all, y_one)
    if stats:
        assert torch.equal(s_all, s_one)
    # and the dispatcher the trainer calls
    y_auto = ops.conv2d_auto_batch(x, pw, b, desc, y_cs=ycs, stats=torch.empty_like(s_all) if stats else None)
    assert torch.equal(y_auto, y_one)


@pytest.mark.parametrize("case", [
    # (H, W, Cin, Cout, transposed)
    ("down 512->1024 @128x128 (generator down3)", (128, 128, 512, 1024, False)),
    ("up 1024->512 @64x64 (generator up1)", (64, 64, 1024, 512, True)),
    ("down 256->512 @256x160 (512x320 frames)", (256, 160, 256, 512, False)),
    ("up 512->256 @128x170 (512x680 frames: ragged tiles)", (128, 170, 512, 256, True)),
    ("down 512->1024 @128x170 (ragged)", (128, 170, 512, 1024, False)),
    ("small ragged down 32->128 @10x14", (10, 14, 32, 128, False)),
    ("small ragged up 64->128 @7x9", (7, 9, 64, 128, True)),
], ids=lambda c: c[0] if isinstance(c, tuple) and isinstance(c[0], str) else None)
def test_polyphase_winograd_matches_torch_and_the_direct_kernel(case, t2v_env):
    """T2V_ALGO_POLYPHASE (csrc/polyphase.hip): the stride-2 3x3 conv (SpatialConvolutionMM, THCUNN.h:664) and
    ConvTranspose2d(3, 2, 1, output_padding 1) (SpatialFullDilatedConvolution, THCUNN.h:794) as polyphase Winograd F(4,2) --
    against torch in fp64 (the yardstick), within 3x the direct implicit-GEMM kernel's own fp32 error or 1e-5 of the output
    scale; its statistics partials through the finalize entry give the map's mean / rstd; bias added; launch after launch on a
    NaN-filled workspace (padding tiles and the hand-over scratch are never read before they are written)."""
    from text2video_amd import ops
    _, (H, W, Cin, Cout, tr) = case
    dev = _dev()
    d0 = ops.conv_desc(H, W, Cin, Cout, 3, 2, 1, ops.PAD_ZERO, tr)
    dp = ops.with_algo(d0, ops.ALGO_POLYPHASE)
    assert ops.polyphase_supported(dp, Cin) and ops.conv_out_dims(dp) == ops.conv_out_dims(d0)
    w = (_rand(Cin, Cout, 3, 3, seed=2, scale=0.03) if tr else _rand(Cout, Cin, 3, 3, seed=2, scale=0.03)).to(dev)
    b = _rand(Cout, seed=3, scale=0.1).to(dev)
    x = torch.relu(_rand(H, W, Cin, seed=4)).to(dev)
    xr = x.permute(2, 0, 1).unsqueeze(0).double()
    ref = (torch.nn.functional.conv_transpose2d(xr, w.double(), b.double(), stride=2, padding=1, output_padding=1) if tr else
           torch.nn.functional.conv2d(xr, w.double(), b.double(), stride=2, padding=1))[0].permute(1, 2, 0)
    p0, pp = ops.pack_conv_weight(w, d0, Cin), ops.pack_conv_weight(w, dp, Cin)
    s0, sp = ops.conv_stats_buffer(d0, dev), ops.conv_stats_buffer(dp, dev)
    y0 = ops.conv2d(x, p0, b, d0, y_cs=Cout, stats=s0)
    ws = ops.winograd_workspace(dp, Cin, dev)
    scale = ref.abs().max().item()
    e0 = (y0.double() - ref).abs().max().item()
    for rep in range(3):
        ws.fill_(float("nan"))
        sp.fill_(float("nan"))
        yp = ops.conv2d_auto(x, pp, b, dp, stats=sp) if rep == 2 else ops.conv2d_winograd(x, pp, b, dp, stats=sp, workspace=ws)
        ep = (yp.double() - ref).abs().max().item()
        assert torch.isfinite(yp).all() and ep <= max(3 * e0, 1e-5 * scale), (rep, ep, e0, scale)
    mr = ops.instance_norm_finalize(sp, dp).view(-1, 2)
    mref = torch.stack([ref.mean((0, 1)), 1.0 / torch.sqrt(ref.var((0, 1), unbiased=False) + 1e-5)], 1).float()
    assert torch.allclose(mr, mref, rtol=2e-5, atol=2e-6), (mr - mref).abs().max().item()
    assert torch.allclose(mr, ops.instance_norm_finalize(s0, d0).view(-1, 2), rtol=2e-5, atol=2e-6)
    # the 81 GEMMs on one block per tile (the fixed-grid kernels off: what a failed dispatch-order self-test leaves): every
    # output is the same K-ordered MFMA chain, so the conv is the same bits
    t2v_env("T2V_WINO_GEMM_SK", "0")
    ws.fill_(float("nan"))
    assert torch.equal(ops.conv2d_winograd(x, pp, b, dp, stats=sp, workspace=ws), yp)
    t2v_env("T2V_WINO_GEMM_SK", "1")
    print("%s: polyphase max|err| %.2e, direct %.2e (output scale %.2f)" % (case[0], ep, e0, scale))
