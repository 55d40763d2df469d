"""Operator-level Python surface over libt2v_hip.so.

Mirrors the reference's operator interface for the hot path -- the torch.nn.functional calls the
vid2vid generator makes on torch 0.4.1 ($SP/torch/nn/functional.py: conv2d, conv_transpose2d,
instance_norm:1258, grid_sample:2046, pad(reflect):2169, avg_pool2d) -- but on NHWC fp32 device
tensors and with the fusions the HIP kernels implement.  PyTorch is used only as the device
allocator / stream provider; every computation below is a call through the C ABI.
"""
import ctypes

from ._xp import torch     # the real torch, or leantorch under vid2vid/test.py's torch-free frame loop

from . import _lib
from ._lib import (ACT_FLOW_W, ACT_LRELU, ACT_NONE, ACT_TANH, ALGO_DIRECT, ALGO_WINOGRAD,  # noqa: F401
                   ALGO_POLYPHASE, ALGO_WINOGRAD_F4, PAD_REFLECT, PAD_ZERO, ConvDesc, check)

_contexts = {}


def context(device=None):
    """Per-device t2v context (created lazily; requires a visible gfx950 GPU)."""
    if not torch.cuda.is_available():
        raise RuntimeError("text2video_amd: no HIP device visible; the frame-synthesis path has no CPU fallback")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    c = _contexts.get(dev)
    if c is None:
        with torch.cuda.device(dev):
            c = _lib.Context(dev)
        _contexts[dev] = c
    return c


def reload_env():
    """The library reads its T2V_* switches once (t2v_create); a process that changes one afterwards (tests, A/B runs)
    calls this."""
    _lib.load().t2v_reload_env()


def check_async_errors():
    """Raise if a kernel of an EARLIER launch reported an error it could only report after the fact (a fixed-grid
    accumulator hand-over that timed out: that launch's output holds NaNs).  Call after synchronising."""
    check(_lib.load().t2v_check_async_errors(), "check_async_errors")


def set_overlap_hint(on):
    """Tell the library that this thread runs a second stream beside the launches it issues (returns the previous value);
    t2v_generator_forward does it for its own two-stream frames -- bench.py uses it to time the kernel those frames run.
    on == 2: the second stream carries fixed-grid GEMMs of its own (the train step's weight gradients): the whole-tile
    fixed-grid GEMM and the Winograd-domain weight gradient launch ONE block per CU, so that the two streams' launches are
    resident side by side (include/t2v.h, t2v_set_overlap_hint)."""
    return int(_lib.load().t2v_set_overlap_hint(2 if on == 2 else (1 if on else 0)))


def fixed_grid_enabled():
    return bool(_lib.load().t2v_fixed_grid_enabled())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous fp32 device tensor" % name)


def round_up(v, m):
    return (v + m - 1) // m * m


def conv_desc(H, W, Cin, Cout, k, stride=1, pad=0, pad_mode=PAD_ZERO, transposed=False, act=ACT_NONE,
              act_scale=1.0, output_padding=None, algo=0):
    if output_padding is None:
        output_padding = 1 if transposed else 0     # the generator's ConvTranspose2d(k3,s2,p1,op1)
    return ConvDesc(H, W, Cin, Cout, k, k, stride, pad, pad_mode, int(transposed), act, act_scale, output_padding,
                    algo)


def winograd_supported(desc, x_cs=None, algo=None):
    """True when `desc` can run as the Winograd variant `algo` (default: desc.algo, or F(2x2,3x3) for a direct desc)."""
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    algo = (desc.algo or ALGO_WINOGRAD) if algo is None else algo
    mask = _lib.load().t2v_conv_winograd_supported(ctypes.byref(desc), x_cs)
    return bool(mask & (2 if algo == ALGO_WINOGRAD_F4 else 1))


def polyphase_supported(desc, x_cs=None):
    """True when `desc` (a stride-2 3x3 conv or its transposed counterpart) can run as ALGO_POLYPHASE (include/t2v.h)."""
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    return bool(_lib.load().t2v_conv_polyphase_supported(ctypes.byref(desc), x_cs) & 1)


def polyphase_pays(desc, x_cs=None):
    """... and the library's own rule selects it for this shape (the faster form: both channel counts >= 256)"""
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    return bool(_lib.load().t2v_conv_polyphase_supported(ctypes.byref(desc), x_cs) & 2)


def best_conv_algo(desc, x_cs=None, cap=0):
    """ALGO_* the library would pick for `desc` (fewest GEMM rows; generator.hip uses the same rule)."""
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    return _lib.load().t2v_conv_best_algo(ctypes.byref(desc), x_cs, cap)


def conv2d_auto(x, packed_w, bias, desc, y_cs=None, stats=None, out=None):
    """conv2d or conv2d_winograd, whichever desc.algo (and the weight packing that goes with it) says."""
    if desc.algo in (ALGO_WINOGRAD, ALGO_WINOGRAD_F4, ALGO_POLYPHASE):
        assert y_cs is None or y_cs == desc.Cout
        return conv2d_winograd(x, packed_w, bias, desc, stats=stats, out=out)
    return conv2d(x, packed_w, bias, desc, y_cs=y_cs, stats=stats, out=out)


def with_algo(desc, algo):
    """copy of a conv descriptor with another algorithm"""
    return ConvDesc(desc.H, desc.W, desc.Cin, desc.Cout, desc.kH, desc.kW, desc.stride, desc.pad, desc.pad_mode,
                    desc.transposed, desc.act, desc.act_scale, desc.output_padding, algo)


GEMM_FORMS = {0: "conv_igemm_kernel<64x64 tiles, one block per tile>", 1: "conv_igemm_kernel<128x128 tiles, one block per tile>",
              2: "wino_gemm_sk_kernel<128x128 tiles on a fixed grid of 2 blocks per CU>",
              3: "wino_gemm_sk_kernel<192x64 tiles on a fixed grid of 2 blocks per CU>",
              4: "wino_gemm_sk_kernel<160x128 tiles on a fixed grid of 1 block per CU>",
              5: "wino_gemm_skr_kernel<ragged 128/96/64/32 x 128 tiles on a fixed grid of 2 blocks per CU>",
              6: "wino_gemm_sk_kernel<256x128 tiles on a fixed grid of 1 block per CU>",
              7: "wino_gemm_skt_kernel<ragged 96..192 x 128 tiles on a fixed grid of 1 block per CU>"}


def winograd_gemm_form(desc, nimg=1):
    """name of the kernel form the F(4x4,3x3) GEMM stage of `desc` takes for `nimg` images per launch"""
    return GEMM_FORMS.get(_lib.load().t2v_conv_winograd_gemm_form(ctypes.byref(desc), nimg), "?")


def winograd_tile_rows(desc):
    """GEMM rows one image contributes per F(4x4,3x3) transform position (tile count, padded)"""
    return _lib.load().t2v_conv_winograd_tile_rows(ctypes.byref(desc))


def winograd_workspace(desc, x_cs, device):
    n = _lib.load().t2v_conv_winograd_workspace_floats(ctypes.byref(desc), x_cs)
    if n == 0:
        raise RuntimeError("winograd_workspace: shape not supported")
    return torch.empty(n, dtype=torch.float32, device=device)


def conv2d_winograd(x, packed_u, bias, desc, stats=None, out=None, workspace=None, stages=7, keep_v=None):
    """3x3 stride-1 reflect-pad-1 conv through Winograd (desc.algo: ALGO_WINOGRAD F(2x2,3x3) | ALGO_WINOGRAD_F4 F(4x4,3x3)).
    stages: bit mask 1 = input transform, 2 = batched GEMM, 4 = output transform (all by default).
    keep_v = (wgrad_ws, batch, slot): the input transform lands in that slot of the Winograd-domain weight gradient's workspace
    (backward_weight_winograd_workspace) and stays there for the backward pass (conv2d_backward_weight_winograd_dy)."""
    c = context()
    _chk(x, "x")
    x_cs = x.shape[-1]
    ws = workspace if workspace is not None else winograd_workspace(desc, x_cs, x.device)
    ho, wo = conv_out_dims(desc)
    y = out if out is not None else torch.empty(ho, wo, desc.Cout, dtype=torch.float32, device=x.device)
    if keep_v is not None:
        assert stages == 7
        wg, batch, slot = keep_v
        check(c.lib.t2v_conv2d_forward_winograd_keep_v(c.handle, _stream(), ctypes.byref(desc), _p(x), x_cs, _p(packed_u), _p(bias),
                                                       _p(y), desc.Cout, _p(stats), _p(ws), _p(wg), batch, slot),
              "conv2d_forward_winograd_keep_v")
        return y
    check(c.lib.t2v_conv2d_forward_winograd_stages(c.handle, _stream(), ctypes.byref(desc), _p(x), x_cs, _p(packed_u),
                                                   _p(bias), _p(y), desc.Cout, _p(stats), _p(ws), stages),
          "conv2d_forward_winograd")
    return y


def conv_out_dims(desc):
    h, w = ctypes.c_int(), ctypes.c_int()
    check(_lib.load().t2v_conv_out_dims(ctypes.byref(desc), ctypes.byref(h), ctypes.byref(w)), "conv_out_dims")
    return h.value, w.value


def pack_conv_weight(weight, desc, x_cs=None, adjoint=False):
    """weight: torch layout ([Cout,Cin,kH,kW], or [Cin,Cout,3,3] if desc.transposed) on the device.
    adjoint=True: `desc` is the data-gradient conv of a stride-1 layer and `weight` that layer's FORWARD weight
    ([desc.Cin, desc.Cout, k, k]); flip and transpose happen inside the packing kernel."""
    c = context()
    _chk(weight, "weight")
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    n = c.lib.t2v_conv_packed_weight_floats(ctypes.byref(desc), x_cs)
    if n == 0:
        raise RuntimeError("pack_conv_weight: %s" % c.lib.t2v_last_error().decode())
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    fn = c.lib.t2v_conv_pack_weight_adjoint if adjoint else c.lib.t2v_conv_pack_weight
    check(fn(c.handle, _stream(), ctypes.byref(desc), x_cs, _p(weight), _p(packed)), "conv_pack_weight")
    return packed


def conv_stats_buffer(desc, device):
    n = _lib.load().t2v_conv_stats_floats(ctypes.byref(desc))
    return torch.empty(max(n, 1), dtype=torch.float32, device=device)


def conv2d(x, packed_w, bias, desc, y_cs=None, stats=None, out=None):
    """x: [H,W,x_cs] NHWC.  Returns y [Hout,Wout,y_cs].  Reflection padding, bias, the head
    activation and (if `stats` is given) the instance-norm partial statistics are fused."""
    c = context()
    _chk(x, "x")
    x_cs = x.shape[-1]
    ho, wo = conv_out_dims(desc)
    y_cs = round_up(desc.Cout, 4) if y_cs is None else y_cs
    y = out if out is not None else torch.empty(ho, wo, y_cs, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_conv2d_forward(c.handle, _stream(), ctypes.byref(desc), _p(x), x_cs, _p(packed_w), _p(bias),
                                   _p(y), y_cs, _p(stats)), "conv2d_forward")
    return y


def conv2d_batch(x, packed_w, bias, desc, y_cs=None, stats=None, out=None):
    """conv2d for a batch x: [B,H,W,x_cs] in ONE launch (direct algorithm) -> y [B,Hout,Wout,y_cs]; `stats` holds B
    consecutive per-image partial blocks (conv_stats_buffer(desc).numel() floats each).  Every image's result is
    conv2d's, bit for bit."""
    c = context()
    _chk(x, "x")
    B, x_cs = x.shape[0], x.shape[-1]
    ho, wo = conv_out_dims(desc)
    y_cs = round_up(desc.Cout, 4) if y_cs is None else y_cs
    y = out if out is not None else torch.empty(B, ho, wo, y_cs, dtype=torch.float32, device=x.device)
    if out is not None:
        _chk(out, "out")
    check(c.lib.t2v_conv2d_forward_batch(c.handle, _stream(), ctypes.byref(desc), B, _p(x), x_cs, _p(packed_w), _p(bias),
                                         _p(y), y_cs, _p(stats)), "conv2d_forward_batch")
    return y


def conv2d_auto_batch(x, packed_w, bias, desc, y_cs=None, stats=None, out=None):
    """conv2d_auto over a batch [B,...]: one launch for the direct algorithm, image by
    image for the Winograd forms (their batches go through the generator's own batched path).  stats: B consecutive blocks."""
    B = x.shape[0]
    if out is None:
        ho, wo = conv_out_dims(desc)
        out = torch.empty(B, ho, wo, round_up(desc.Cout, 4) if y_cs is None else y_cs, dtype=torch.float32, device=x.device)
    if desc.algo == ALGO_DIRECT and B > 1 and x.is_contiguous() and out.is_contiguous() and \
            (stats is None or stats.is_contiguous()):
        return conv2d_batch(x, packed_w, bias, desc, y_cs=y_cs, stats=stats, out=out)
    n = stats.numel() // B if stats is not None else 0
    for i in range(B):
        conv2d_auto(x[i], packed_w, bias, desc, y_cs=y_cs, stats=stats[i * n:(i + 1) * n] if stats is not None else None, out=out[i])
    return out


def instance_norm_finalize(stats, desc, eps=1e-5, out=None, running=None):
    """running = (running_mean, running_var, momentum, times): BatchNorm2d(train)'s running statistics (a batch of one) are
    moved `times` times by these statistics in the same launch (t2v_batch_norm_finalize_running)."""
    c = context()
    mr = out if out is not None else torch.empty(desc.Cout, 2, dtype=torch.float32, device=stats.device)
    if running is not None:
        rm, rv, momentum, times = running
        check(c.lib.t2v_batch_norm_finalize_running(c.handle, _stream(), ctypes.byref(desc), 1, _p(stats), eps, _p(mr), _p(rm), _p(rv),
                                                    momentum, int(times)), "batch_norm_finalize_running")
        return mr
    check(c.lib.t2v_instance_norm_finalize(c.handle, _stream(), ctypes.byref(desc), _p(stats), eps, _p(mr)),
          "instance_norm_finalize")
    return mr


def instance_norm_apply(x, mean_rstd, gamma=None, beta=None, res1=None, res2=None, relu=False, out=None):
    c = context()
    _chk(x, "x")
    C = x.shape[-1]
    npix = x.numel() // C
    y = out if out is not None else torch.empty_like(x)
    check(c.lib.t2v_instance_norm_apply(c.handle, _stream(), _p(x), _p(mean_rstd), _p(gamma), _p(beta), _p(res1),
                                        _p(res2), _p(y), npix, C, int(relu)), "instance_norm_apply")
    return y


def conv_norm_act(x, packed_w, bias, desc, gamma=None, beta=None, relu=True, res1=None, res2=None, eps=1e-5):
    """[ReflPad,] Conv, InstanceNorm(+affine), [ReLU], [+res]: one conv launch + finalize + apply."""
    stats = conv_stats_buffer(desc, x.device)
    y = conv2d(x, packed_w, bias, desc, y_cs=desc.Cout, stats=stats)
    mr = instance_norm_finalize(stats, desc, eps)
    return instance_norm_apply(y, mr, gamma, beta, res1, res2, relu, out=y)


def flow_warp_composite(raw, fw, prev, prev_c0, want_warp=False, out=None):
    """raw,fw: [H,W,4]; prev: [H,W,prev_cs] (3 channels from prev_c0).  out = raw*w + warp*(1-w)."""
    c = context()
    H, W = raw.shape[0], raw.shape[1]
    out = torch.empty_like(raw) if out is None else out
    warp = torch.empty_like(raw) if want_warp else None
    check(c.lib.t2v_flow_warp_composite(c.handle, _stream(), _p(raw), _p(fw), _p(prev), prev.shape[-1], prev_c0,
                                        _p(out), _p(warp), H, W), "flow_warp_composite")
    return (out, warp) if want_warp else out


def flow_warp(fw, img, c0=0, out=None):
    """resample(img[..., c0:c0+3], flow): fw [H,W,4] (flow_x, flow_y in pixels; channel 2 ignored) -> [H,W,4]."""
    c = context()
    H, W = fw.shape[0], fw.shape[1]
    out = torch.empty(H, W, 4, dtype=torch.float32, device=fw.device) if out is None else out
    check(c.lib.t2v_flow_warp_composite(c.handle, _stream(), None, _p(fw), _p(img), img.shape[-1], c0, None, _p(out),
                                        H, W), "flow_warp")
    return out


def flow_warp_composite_backward(d_out, d_warp, raw, fw, prev, prev_c0, want_d_prev=False):
    """Adjoint of flow_warp_composite / flow_warp.  d_out, d_warp: [H,W,4] or None (not both); raw may be None when
    d_out is.  Returns (d_raw | None, d_fw [H,W,4] = (d flow_x, d flow_y, d weight, 0), d_prev | None)."""
    c = context()
    H, W = fw.shape[0], fw.shape[1]
    d_raw = torch.empty_like(fw) if d_out is not None else None
    d_fw = torch.empty_like(fw)
    d_prev = torch.zeros_like(prev) if want_d_prev else None
    check(c.lib.t2v_flow_warp_composite_backward(c.handle, _stream(), _p(d_out), _p(d_warp), _p(raw), _p(fw), _p(prev),
                                                 prev.shape[-1], prev_c0, _p(d_raw), _p(d_fw), _p(d_prev), H, W),
          "flow_warp_composite_backward")
    return d_raw, d_fw, d_prev


def avgpool3x3s2(x):
    """AvgPool2d(3, 2, 1, count_include_pad=False) on [H,W,C]."""
    c = context()
    _chk(x, "x")
    H, W, C = x.shape
    y = torch.empty((H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_avgpool3x3s2(c.handle, _stream(), _p(x), _p(y), H, W, C), "avgpool3x3s2")
    return y


def nchw_to_nhwc(x, cs=None, out=None):
    """[C,H,W] -> [H,W,cs] (extra channels zero)."""
    c = context()
    _chk(x, "x")
    C, H, W = x.shape
    cs = round_up(C, 4) if cs is None else cs
    y = out if out is not None else torch.empty(H, W, cs, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_nchw_to_nhwc(c.handle, _stream(), _p(x), _p(y), C, H, W, cs), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, C=None):
    """[H,W,cs] -> [C,H,W]."""
    c = context()
    _chk(x, "x")
    H, W, cs = x.shape
    C = cs if C is None else C
    y = torch.empty(C, H, W, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_nhwc_to_nchw(c.handle, _stream(), _p(x), _p(y), C, H, W, cs), "nhwc_to_nchw")
    return y


def pose_u8_to_f32(src_u8, dst, c0):
    """uint8 [H,W,3] pose map -> channels [c0,c0+3) of dst [H,W,cs] as (v/255-0.5)/0.5."""
    c = context()
    H, W, _ = src_u8.shape
    check(c.lib.t2v_pose_u8_to_f32(c.handle, _stream(), _p(src_u8), _p(dst), H * W, dst.shape[-1], c0),
          "pose_u8_to_f32")
    return dst


def tensor2im_u8(x):
    """util.tensor2im on the device: uint8 of (x+1)/2*255, same layout."""
    c = context()
    _chk(x, "x")
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(c.lib.t2v_tensor2im_u8(c.handle, _stream(), _p(x), _p(y), x.numel()), "tensor2im_u8")
    return y


def copy_channels(src, src_c0, dst, dst_c0, nc):
    c = context()
    npix = src.numel() // src.shape[-1]
    check(c.lib.t2v_copy_channels(c.handle, _stream(), _p(src), src.shape[-1], src_c0, _p(dst), dst.shape[-1],
                                  dst_c0, nc, npix), "copy_channels")
    return dst


# ------------------------------------------------------------------------------------------------
# train-step pieces
# ------------------------------------------------------------------------------------------------
def batch_norm_finalize(stats, desc, batch, eps=1e-5, running=None):
    """BatchNorm2d(train) statistics over a batch: `stats` holds `batch` consecutive per-image
    partial blocks written by conv2d(..., stats=stats[b * n : (b + 1) * n]).  running: as instance_norm_finalize."""
    c = context()
    mr = torch.empty(desc.Cout, 2, dtype=torch.float32, device=stats.device)
    if running is not None:
        rm, rv, momentum, times = running
        check(c.lib.t2v_batch_norm_finalize_running(c.handle, _stream(), ctypes.byref(desc), batch, _p(stats), eps, _p(mr), _p(rm),
                                                    _p(rv), momentum, int(times)), "batch_norm_finalize_running")
        return mr
    check(c.lib.t2v_batch_norm_finalize(c.handle, _stream(), ctypes.byref(desc), batch, _p(stats), eps, _p(mr)),
          "batch_norm_finalize")
    return mr


def batch_norm_update_running(mean_rstd, running_mean, running_var, n, momentum=0.1, eps=1e-5):
    """BatchNorm2d's running statistics, updated in place from a finalize call's (mean, rstd) table over n values per
    channel ($SP/torch/nn/modules/batchnorm.py:57-64: exponential average with `momentum`, unbiased variance)."""
    c = context()
    check(c.lib.t2v_batch_norm_update_running(c.handle, _stream(), _p(mean_rstd), _p(running_mean), _p(running_var), int(n),
                                              running_mean.numel(), momentum, eps), "batch_norm_update_running")


_scratch = {}


def _reduce_scratch(device):
    s = _scratch.get(device)
    if s is None:
        s = _scratch[device] = torch.empty(2048, dtype=torch.float32, device=device)
    return s


def sum_sq_diff_const(x, c0):
    """sum((x - c0)^2) as a 1-element device tensor (LSGAN MSE numerator)."""
    c = context()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_sum_sq_diff_const(c.handle, _stream(), _p(x), float(c0), x.numel(), _p(_reduce_scratch(x.device)),
                                      _p(out)), "sum_sq_diff_const")
    return out


def sum_abs_diff(a, b):
    """sum(|a - b|) as a 1-element device tensor (L1 feature-matching numerator)."""
    c = context()
    assert a.shape == b.shape
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(c.lib.t2v_sum_abs_diff(c.handle, _stream(), _p(a), _p(b), a.numel(), _p(_reduce_scratch(a.device)), _p(out)),
          "sum_abs_diff")
    return out


def sum_abs_diff_masked(a, b, mask, c0, C):
    """sum over pixels and the channels [c0, c0+C) of mask[pix] * |a - b| ([..., cs] tensors; b / mask may be None)."""
    c = context()
    cs = a.shape[-1]
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(c.lib.t2v_sum_abs_diff_masked(c.handle, _stream(), _p(a), _p(b), _p(mask), a.numel() // cs, c0, C, cs,
                                        _p(_reduce_scratch(a.device)), _p(out)), "sum_abs_diff_masked")
    return out


def sum_abs_diff_masked_backward(a, b, mask, c0, C, scale):
    c = context()
    cs = a.shape[-1]
    da = torch.empty_like(a)
    check(c.lib.t2v_sum_abs_diff_masked_backward(c.handle, _stream(), _p(a), _p(b), _p(mask), float(scale),
                                                 a.numel() // cs, c0, C, cs, _p(da)), "sum_abs_diff_masked_backward")
    return da


def loss_terms(term_ptrs, term_ints, term_floats, chunk_term, chunk_off, term_chunk0, nterms, nchunks, chunk, partials, out):
    """all scalar loss terms of a step (values and gradient seeds) in one launch: t2v_loss_terms (include/t2v.h)"""
    c = context()
    check(c.lib.t2v_loss_terms(c.handle, _stream(), _p(term_ptrs), _p(term_ints), _p(term_floats), _p(chunk_term), _p(chunk_off),
                               _p(term_chunk0), int(nterms), int(nchunks), int(chunk), _p(partials), _p(out)), "loss_terms")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """In-place fused Adam update of one flat fp32 tensor (torch.optim.Adam.step semantics)."""
    c = context()
    check(c.lib.t2v_adam_step(c.handle, _stream(), _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                              lr, beta1, beta2, eps, int(step)), "adam_step")
    return param


def adam_step_multi(ptrs, nelem, step_size, chunk_tensor, chunk_off, chunk, beta1, beta2, eps):
    """One launch over many tensors (tables are device tensors: ptrs int64 [T,4], nelem int64 [T], step_size float32
    [T], chunk_tensor int32 [NC], chunk_off int64 [NC])."""
    c = context()
    check(c.lib.t2v_adam_step_multi(c.handle, _stream(), _p(ptrs), _p(nelem), _p(step_size), _p(chunk_tensor), _p(chunk_off),
                                    chunk_tensor.numel(), int(chunk), beta1, beta2, eps), "adam_step_multi")


def conv2d_backward_weight(x, dy, desc, accumulate_into=None):
    """Weight gradient in the PACKED layout.  x: [B,H,W,x_cs], dy: [B,Hout,Wout,dy_cs] (or 3-D, B=1)."""
    c = context()
    if x.dim() == 3:
        x, dy = x.unsqueeze(0), dy.unsqueeze(0)
    _chk(x, "x")
    _chk(dy, "dy")
    B, x_cs, dy_cs = x.shape[0], x.shape[-1], dy.shape[-1]
    n = c.lib.t2v_conv_packed_weight_floats(ctypes.byref(desc), x_cs)
    # (every real entry of the packed gradient is written -- by the kernel, its in-kernel combine or the reduce pass)
    dw = accumulate_into if accumulate_into is not None else torch.empty(n, dtype=torch.float32, device=x.device)
    nws = c.lib.t2v_conv_backward_weight_workspace_floats(ctypes.byref(desc), x_cs, B)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    check(c.lib.t2v_conv2d_backward_weight(c.handle, _stream(), ctypes.byref(desc), B, _p(x), x_cs, _p(dy), dy_cs,
                                           _p(dw), int(accumulate_into is not None), _p(ws)), "conv2d_backward_weight")
    return dw


def backward_weight_strided_supported(desc, x_cs, dy_cs):
    return bool(_lib.load().t2v_conv_backward_weight_strided_supported(ctypes.byref(desc), x_cs, dy_cs))


def conv2d_backward_weight_pair(x0, dy0, x1, dy1, desc, accumulate_into=None):
    """packed dW of the two images x0 / x1 ([H,W,x_cs] each, in buffers of their own) and their output gradients in ONE
    launch (t2v_conv2d_backward_weight_strided): what conv2d_backward_weight computes on torch.stack of them, bit for bit."""
    c = context()
    for t, n in ((x0, "x0"), (x1, "x1"), (dy0, "dy0"), (dy1, "dy1")):
        _chk(t, n)
    assert x0.shape == x1.shape and dy0.shape == dy1.shape and x0.dim() == 3
    x_cs, dy_cs = x0.shape[-1], dy0.shape[-1]
    n = c.lib.t2v_conv_packed_weight_floats(ctypes.byref(desc), x_cs)
    dw = accumulate_into if accumulate_into is not None else torch.empty(n, dtype=torch.float32, device=x0.device)
    nws = c.lib.t2v_conv_backward_weight_workspace_floats(ctypes.byref(desc), x_cs, 2)
    ws = torch.empty(max(nws, 1), dtype=torch.float32, device=x0.device)
    xs, ys = (x1.data_ptr() - x0.data_ptr()) // 4, (dy1.data_ptr() - dy0.data_ptr()) // 4
    if xs == 0 or ys == 0:
        raise ValueError("conv2d_backward_weight_pair: the two images share a buffer")
    check(c.lib.t2v_conv2d_backward_weight_strided(c.handle, _stream(), ctypes.byref(desc), 2, _p(x0), x_cs, xs, _p(dy0), dy_cs, ys,
                                                   _p(dw), int(accumulate_into is not None), _p(ws)), "conv2d_backward_weight_strided")
    return dw


def backward_weight_winograd_supported(desc, x_cs, dy_cs):
    return bool(_lib.load().t2v_conv_backward_weight_winograd_supported(ctypes.byref(desc), x_cs, dy_cs))


def conv2d_backward_weight_winograd(x, dy, desc, accumulate_into=None, out=None):
    """Weight gradient of a 3x3 stride-1 conv through the Winograd domain (F(4x4,3x3)), TORCH layout
    [Cout,Cin,3,3].  x: [B,H,W,Cin], dy: [B,Ho,Wo,Cout] (or 3-D, B=1)."""
    c = context()
    if x.dim() == 3:
        x, dy = x.unsqueeze(0), dy.unsqueeze(0)
    _chk(x, "x")
    _chk(dy, "dy")
    B, x_cs, dy_cs = x.shape[0], x.shape[-1], dy.shape[-1]
    dw = accumulate_into if accumulate_into is not None else out if out is not None else \
        torch.empty(desc.Cout, desc.Cin, 3, 3, dtype=torch.float32, device=x.device)
    nws = c.lib.t2v_conv_backward_weight_winograd_workspace_floats(ctypes.byref(desc), x_cs, B)
    if nws == 0:
        raise RuntimeError("conv2d_backward_weight_winograd: shape not supported")
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_conv2d_backward_weight_winograd(c.handle, _stream(), ctypes.byref(desc), B, _p(x), x_cs, _p(dy), dy_cs,
                                                    _p(dw), int(accumulate_into is not None), _p(ws)),
          "conv2d_backward_weight_winograd")
    return dw


def backward_weight_winograd_workspace(desc, x_cs, batch, device):
    n = _lib.load().t2v_conv_backward_weight_winograd_workspace_floats(ctypes.byref(desc), x_cs, batch)
    if n == 0:
        raise RuntimeError("backward_weight_winograd_workspace: shape not supported")
    return torch.empty(n, dtype=torch.float32, device=device)


def conv2d_backward_weight_winograd_stages(x, dy, desc, ws, batch, b0, reduce, out=None, accumulate=False):
    """Staged form: transform the images x, dy ([nb,H,W,C] / [nb,Ho,Wo,Cout]) into slots [b0, b0+nb) of `ws`
    (backward_weight_winograd_workspace(desc, x_cs, batch)); with reduce=True also run the reduction over all
    `batch` slots and return dW in torch layout, else return None."""
    c = context()
    if x.dim() == 3:
        x, dy = x.unsqueeze(0), dy.unsqueeze(0)
    _chk(x, "x")
    _chk(dy, "dy")
    dw = (out if out is not None else torch.empty(desc.Cout, desc.Cin, 3, 3, dtype=torch.float32, device=x.device)) \
        if reduce else None
    check(c.lib.t2v_conv2d_backward_weight_winograd_stages(c.handle, _stream(), ctypes.byref(desc), batch, b0, x.shape[0],
                                                           _p(x), x.shape[-1], _p(dy), dy.shape[-1], _p(dw),
                                                           int(bool(accumulate and out is not None)), _p(ws),
                                                           3 if reduce else 1),
          "conv2d_backward_weight_winograd_stages")
    return dw


def conv2d_backward_weight_winograd_dy(dy, desc, ws, batch, slot, x_cs):
    """A dy A^T of one image ([Ho,Wo,Cout] or [1,Ho,Wo,Cout]) into slot `slot` of `ws`, whose V slot the forward pass has
    filled already (conv2d_winograd(keep_v=...))."""
    c = context()
    if dy.dim() == 3:
        dy = dy.unsqueeze(0)
    _chk(dy, "dy")
    check(c.lib.t2v_conv2d_backward_weight_winograd_stages(c.handle, _stream(), ctypes.byref(desc), batch, slot, 1, None, x_cs,
                                                           _p(dy), dy.shape[-1], None, 0, _p(ws), 1),
          "conv2d_backward_weight_winograd_stages")


def conv2d_backward_weight_winograd_dy_norm(conv_out, dy, mean_rstd, gamma, beta, relu, sums, desc, ws, batch, slot, x_cs):
    """conv2d_backward_weight_winograd_dy of the gradient in FRONT of the layer's norm, formed on the fly from the gradient
    behind it (`dy`), the conv's raw output and the sums of instance_norm_backward(..., sums_only=True): one image."""
    c = context()
    _chk(conv_out, "conv_out")
    _chk(dy, "dy")
    assert conv_out.numel() == dy.numel() and conv_out.shape[-1] == desc.Cout
    check(c.lib.t2v_conv2d_backward_weight_winograd_dy_norm(c.handle, _stream(), ctypes.byref(desc), batch, slot, x_cs, _p(conv_out),
                                                            _p(dy), _p(mean_rstd), _p(gamma), _p(beta), int(relu), _p(sums), _p(ws)),
          "conv2d_backward_weight_winograd_dy_norm")


def conv2d_backward_weight_winograd_reduce(desc, ws, batch, x_cs, dy_cs, out=None, accumulate=False):
    """Reduction stage alone over the `batch` slots already transformed into `ws` -> dW in torch layout."""
    c = context()
    dw = out if out is not None else torch.empty(desc.Cout, desc.Cin, 3, 3, dtype=torch.float32, device=ws.device)
    check(c.lib.t2v_conv2d_backward_weight_winograd_stages(c.handle, _stream(), ctypes.byref(desc), batch, 0, 0, None, x_cs,
                                                           None, dy_cs, _p(dw), int(bool(accumulate and out is not None)),
                                                           _p(ws), 2),
          "conv2d_backward_weight_winograd_stages")
    return dw


def backward_data_winograd_supported(desc, x_cs, dy_cs):
    return bool(_lib.load().t2v_conv_backward_data_winograd_supported(ctypes.byref(desc), x_cs, dy_cs))


def pack_conv_weight_transposed(w, desc, x_cs):
    """U^T of a forward 3x3 layer (torch weight [Cout,Cin,3,3] on the device) for conv2d_backward_data_winograd."""
    c = context()
    n = c.lib.t2v_conv_backward_data_winograd_weight_floats(ctypes.byref(desc), x_cs)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(c.lib.t2v_conv_pack_weight_transposed(c.handle, _stream(), ctypes.byref(desc), x_cs, _p(w.contiguous()), _p(out)),
          "conv_pack_weight_transposed")
    return out


def backward_data_winograd_takes_forward_weights(desc, x_cs, dy_cs):
    """the data gradient reads the forward layer's own F(4x4) packing (pack_conv_weight) in place: no transposed copy"""
    return bool(_lib.load().t2v_conv_backward_data_winograd_takes_forward_weights(ctypes.byref(desc), x_cs, dy_cs))


def conv2d_backward_data_winograd(desc, batch, slot, wgrad_ws, x_cs, ut, out=None, forward_weights=False):
    """dx [H,W,x_cs] of image `slot` of a batch whose A dy A^T already sits in the weight gradient's workspace `wgrad_ws`
    (conv2d_backward_weight_winograd_stages): the transposed Winograd algorithm (include/t2v.h).  `ut`: the transposed
    packing (pack_conv_weight_transposed), or with forward_weights the forward layer's own F(4x4) packing."""
    c = context()
    dx = torch.empty(desc.H, desc.W, x_cs, dtype=torch.float32, device=wgrad_ws.device) if out is None else out
    n = c.lib.t2v_conv_backward_data_winograd_scratch_floats(ctypes.byref(desc), x_cs)
    scratch = torch.empty(n, dtype=torch.float32, device=wgrad_ws.device)
    fn = c.lib.t2v_conv2d_backward_data_winograd_fw if forward_weights else c.lib.t2v_conv2d_backward_data_winograd
    check(fn(c.handle, _stream(), ctypes.byref(desc), batch, slot, _p(wgrad_ws), x_cs, _p(ut), _p(scratch), _p(dx)),
          "conv2d_backward_data_winograd")
    return dx


def maxpool2x2(x):
    """MaxPool2d(2,2) on [H,W,C] or [B,H,W,C] (H even for a batch: the images are pooled as one tall image)."""
    c = context()
    _chk(x, "x")
    if x.dim() == 4:
        B, H, W, C = x.shape
        assert H % 2 == 0, "batched maxpool2x2 needs an even height"
        return maxpool2x2(x.reshape(B * H, W, C)).reshape(B, H // 2, W // 2, C)
    H, W, C = x.shape
    y = torch.empty(H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_maxpool2x2(c.handle, _stream(), _p(x), _p(y), H, W, C), "maxpool2x2")
    return y


def maxpool2x2_backward(x, dy):
    c = context()
    _chk(x, "x")
    _chk(dy, "dy")
    if x.dim() == 4:
        B, H, W, C = x.shape
        return maxpool2x2_backward(x.reshape(B * H, W, C), dy.reshape(B * (H // 2), W // 2, C)).reshape(B, H, W, C)
    H, W, C = x.shape
    dx = torch.empty_like(x)
    check(c.lib.t2v_maxpool2x2_backward(c.handle, _stream(), _p(x), _p(dy), _p(dx), H, W, C), "maxpool2x2_backward")
    return dx


def unpack_conv_weight(packed, desc, x_cs=None):
    """packed layout -> torch layout ([Cout,Cin,kH,kW] or [Cin,Cout,3,3])."""
    c = context()
    x_cs = round_up(desc.Cin, 4) if x_cs is None else x_cs
    shape = (desc.Cin, desc.Cout, desc.kH, desc.kW) if desc.transposed else (desc.Cout, desc.Cin, desc.kH, desc.kW)
    w = torch.empty(shape, dtype=torch.float32, device=packed.device)
    check(c.lib.t2v_conv_unpack_weight(c.handle, _stream(), ctypes.byref(desc), x_cs, _p(packed), _p(w)),
          "conv_unpack_weight")
    return w


def unpack_conv_weight_into(packed, desc, x_cs, out, accumulate):
    """packed layout -> torch layout written (or added, accumulate=True) into `out`: a contiguous tensor of the weight's
    torch shape, e.g. the parameter's slice of a flat gradient bucket."""
    c = context()
    check(c.lib.t2v_conv_unpack_weight_into(c.handle, _stream(), ctypes.byref(desc), x_cs, _p(packed), _p(out), int(accumulate)),
          "conv_unpack_weight_into")
    return out


def accumulate_(dst, src, overwrite=False):
    """dst = src (overwrite) or dst += src, in place on contiguous fp32 tensors of equal size."""
    c = context()
    assert dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    check(c.lib.t2v_accumulate(c.handle, _stream(), _p(dst), _p(src), dst.numel(), int(overwrite)), "accumulate")
    return dst


def unzip2_(src, dst0, dst1, overwrite=False):
    """src [C,2] -> dst0 (+)= src[:,0], dst1 (+)= src[:,1]"""
    c = context()
    check(c.lib.t2v_unzip2(c.handle, _stream(), _p(src), _p(dst0), _p(dst1), dst0.numel(), int(overwrite)), "unzip2")


def scale_(x, s):
    c = context()
    assert x.is_contiguous()
    check(c.lib.t2v_scale(c.handle, _stream(), _p(x), x.numel(), float(s)), "scale")
    return x


def zero_(x):
    c = context()
    assert x.is_contiguous()
    check(c.lib.t2v_zero(c.handle, _stream(), _p(x), x.numel() * x.element_size()), "zero")
    return x


def channel_sum(x, C=None, out=None):
    """sum over all pixels per channel of an NHWC tensor (bias gradient)."""
    c = context()
    _chk(x, "x")
    cs = x.shape[-1]
    C = cs if C is None else C
    out = torch.empty(C, dtype=torch.float32, device=x.device) if out is None else out
    scratch = torch.empty(256 * C, dtype=torch.float32, device=x.device)
    check(c.lib.t2v_channel_sum(c.handle, _stream(), _p(x), x.numel() // cs, C, cs, _p(scratch), _p(out)),
          "channel_sum")
    return out


def reflect_pad_backward(dxp, pad, out=None):
    """adjoint of ReflectionPad2d(pad): [H+2p, W+2p, C] -> [H, W, C]."""
    c = context()
    _chk(dxp, "dxp")
    Hp, Wp, C = dxp.shape
    H, W = Hp - 2 * pad, Wp - 2 * pad
    dx = out if out is not None else torch.empty(H, W, C, dtype=torch.float32, device=dxp.device)
    _chk(dx, "dx")
    check(c.lib.t2v_reflect_pad_backward(c.handle, _stream(), _p(dxp), _p(dx), H, W, C, pad), "reflect_pad_backward")
    return dx


def instance_norm_backward(x, dy, mean_rstd, gamma=None, beta=None, relu=0, out=None, affine_into=None, sums_only=False):
    """x, dy: [..., C] (all leading dims are the pixels of ONE statistics group).  Returns
    (dx, dbeta_dgamma [C,2]).  affine_into = (d_beta, d_gamma, overwrite): the two sums also land in those gradient
    tensors (written or added) inside the same launches.  sums_only: dx is None -- its consumer forms it
    (conv2d_backward_weight_winograd_dy_norm)."""
    c = context()
    _chk(x, "x")
    _chk(dy, "dy")
    C = x.shape[-1]
    npix = x.numel() // C
    dx = None if sums_only else (out if out is not None else torch.empty_like(x))
    if dx is not None:
        _chk(dx, "dx")
    sums = torch.empty(C, 2, dtype=torch.float32, device=x.device)
    scratch = torch.empty(128 * C * 2, dtype=torch.float32, device=x.device)
    if affine_into is not None:
        d_beta, d_gamma, overwrite = affine_into
        check(c.lib.t2v_instance_norm_backward_affine(c.handle, _stream(), _p(x), _p(dy), _p(mean_rstd), _p(gamma), _p(beta),
                                                      int(relu), npix, C, _p(scratch), _p(dx), _p(sums), _p(d_beta), _p(d_gamma),
                                                      int(bool(overwrite))), "instance_norm_backward_affine")
        return dx, sums
    check(c.lib.t2v_instance_norm_backward(c.handle, _stream(), _p(x), _p(dy), _p(mean_rstd), _p(gamma), _p(beta),
                                           int(relu), npix, C, _p(scratch), _p(dx), _p(sums)), "instance_norm_backward")
    return dx, sums


def act_backward(dy, y, act, slope=1.0):
    c = context()
    dpre = torch.empty_like(dy)
    check(c.lib.t2v_act_backward(c.handle, _stream(), _p(dy), _p(y), act, slope, dy.numel(), _p(dpre)), "act_backward")
    return dpre


def avgpool3x3s2_backward(dy, H, W):
    c = context()
    C = dy.shape[-1]
    dx = torch.empty(H, W, C, dtype=torch.float32, device=dy.device)
    check(c.lib.t2v_avgpool3x3s2_backward(c.handle, _stream(), _p(dy), _p(dx), H, W, C), "avgpool3x3s2_backward")
    return dx


def sum_sq_diff_const_backward(x, c0, scale):
    c = context()
    dx = torch.empty_like(x)
    check(c.lib.t2v_sum_sq_diff_const_backward(c.handle, _stream(), _p(x), float(c0), float(scale), x.numel(), _p(dx)),
          "sum_sq_diff_const_backward")
    return dx


def sum_abs_diff_backward(a, b, scale):
    c = context()
    da = torch.empty_like(a)
    check(c.lib.t2v_sum_abs_diff_backward(c.handle, _stream(), _p(a), _p(b), float(scale), a.numel(), _p(da)),
          "sum_abs_diff_backward")
    return da
