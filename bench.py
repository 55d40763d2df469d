#!/usr/bin/env python
"""bench.py -- frames/sec of the vid2vid pose->RGB generator on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns one rank per GPU itself, text2video_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one output frame of the hot path: pack the 3-pose-map window (uint8 maps already
resident in HBM), run CompositeGenerator (SURVEY config 2: ngf 128, 3 down-samplings, 9 ResNet
blocks, 512x512), shift the 2-frame FIFO and convert the frame to uint8 (tensor2im) in HBM.
The HEADLINE (`value`) is the generator WITH its flow branch and flow-warp compositor -- the variant
that contains every component north_star names (3.32 TFLOP/frame); the no-flow variant (2.57
TFLOP/frame; what --openpose_only may select upstream) is timed in the same run over the
same K steps and reported in config.variants (`--variant noflow` swaps the roles).  N>1: each rank
runs its own 64-frame-style chunk (sequence-chunk sharding; weak scaling) and the chunk
outputs are all-gathered over RCCL inside the timed region.  Weights are random-init (seeded), data
is synthetic: there is no network for checkpoints.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel (the batched-GEMM stage of the Winograd F(4x4,3x3) 1024->1024 ResnetBlock
                  conv: 19.3 GFLOP of executed fp32 MFMA work per launch at 512x512) timed with HIP events on
                  the launch stream vs the fp32 MFMA peak (157.3 TFLOP/s); "layer" = the whole conv
  "cpu_baseline": the CPU oracle (stock torch fp32) timed on this box's host cores on a bounded
                  sample of the same workload (3 frames each with all, 32 and 8 threads -- the fastest is `value`,
                  with its thread count in `cores`; SURVEY 8d asks for the 8-thread figure);
                  its frames are also compared with the timed HIP model's ("parity": max |delta| per
                  pixel, teacher-forced, tolerance 1e-3 = north_star).
  "e2e":          the drop-in test.py frame loop (rasterise pose JSONs -> H2D -> generator -> D2H ->
                  JPEG files, text2video_amd.model.run_test) on a dataset in the reference's layout, at
                  512x512, 512x680 and the reference's cropped 512x320, with the rasteriser worker count.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
SPEC_SCLK_MHZ = 2400.0         # ... at the 2.4 GHz peak engine clock (256 CUs x 256 FLOP/clk/CU x 2.4 GHz)
REF_SCLK_MHZ = 2360.0          # the clock the round-4/5 train-step targets were quoted at (<= 88 ms x 2.36 / sclk)


def gflop_per_frame(H, W, flow, ngf=128, n_down=3, n_blocks=9):
    """Algorithmic FLOPs: sum 2*k*k*Cin*Cout*Hout*Wout, transposed convs on the input grid."""
    f = 0.0
    for cin in (9, 6):
        f += 2 * 49 * cin * ngf * H * W
        for i in range(n_down):
            f += 2 * 9 * (ngf << i) * (ngf << (i + 1)) * (H >> (i + 1)) * (W >> (i + 1))
        f += (n_blocks - n_blocks // 2) * 2 * 2 * 9 * (ngf << n_down) ** 2 * (H >> n_down) * (W >> n_down)
    branch = (n_blocks // 2) * 2 * 2 * 9 * (ngf << n_down) ** 2 * (H >> n_down) * (W >> n_down)
    for i in range(n_down):
        l = n_down - i
        branch += 2 * 9 * (ngf << l) * (ngf << (l - 1)) * (H >> l) * (W >> l)
    f += branch + 2 * 49 * ngf * 3 * H * W
    if flow:
        f += branch + 2 * 49 * ngf * 3 * H * W
    return f / 1e9


def local_gflop_per_frame(H, W, flow, ngf=64, n_blocks=3):
    """CompositeLocalGenerator add-on at full resolution."""
    f = 0.0
    for cin in (9, 6):
        f += 2 * 49 * cin * ngf * H * W + 2 * 9 * ngf * 2 * ngf * (H // 2) * (W // 2)
    branch = n_blocks * 2 * 2 * 9 * (2 * ngf) ** 2 * (H // 2) * (W // 2) + 2 * 9 * 2 * ngf * ngf * (H // 2) * (W // 2)
    f += branch + 2 * 49 * ngf * 3 * H * W
    if flow:
        f += branch + 2 * 49 * ngf * 3 * H * W
    return f / 1e9


def synthetic_pose_u8(n, H, W, seed):
    """pose maps as the dataset would hand them over: uint8 HWC, black background, ~1 % coloured pixels"""
    rng = np.random.default_rng(seed)
    a = np.zeros((n, H, W, 3), np.uint8)
    m = rng.random((n, H, W)) < 0.01
    a[m] = rng.integers(0, 256, size=(int(m.sum()), 3), dtype=np.uint8)
    return a


class ClockSampler:
    """GPU core clock of THIS rank's device sampled from sysfs (pp_dpm_sclk: the active DPM level carries a '*') every
    50 ms by a background thread while a timed region runs; plus the box / device identity.  Best effort: fields are
    None where the files are missing."""

    def __init__(self, device_index, period=0.05):
        import glob
        import socket
        self.path, self.samples, self._stop, self._thr, self.period = None, [], None, None, period
        self.ident = {"host": socket.gethostname(), "gpu": None, "unique_id": None, "pci_bus_id": None}
        try:
            props = torch.cuda.get_device_properties(device_index)
            self.ident["gpu"] = props.name
            bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
            self.ident["pci_bus_id"] = bus
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
            pick = [c for c in cards if bus in os.path.realpath(os.path.dirname(c))] or cards
            if pick:
                self.path = pick[0]
                uid = os.path.join(os.path.dirname(self.path), "unique_id")
                if os.path.exists(uid):
                    self.ident["unique_id"] = open(uid).read().strip()
        except Exception:
            pass

    def _read(self):
        try:
            for line in open(self.path):
                if "*" in line:
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None

    def __enter__(self):
        if self.path is not None:
            import threading
            self._stop = threading.Event()

            def loop():
                while not self._stop.is_set():
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    self._stop.wait(self.period)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join()

    def fork(self, period=0.01):
        """a fresh sampler on the same sysfs file (another timed region of the same run: the roofline kernel loop, the
        train-step block); 10 ms period by default -- those regions last 30-500 ms"""
        c = ClockSampler.__new__(ClockSampler)
        c.path, c.samples, c._stop, c._thr, c.period, c.ident = self.path, [], None, None, period, self.ident
        return c

    def mean_mhz(self):
        return round(sum(self.samples) / len(self.samples), 1) if self.samples else None

    def summary(self):
        s = self.samples
        return dict(self.ident, sclk_mhz_mean=round(sum(s) / len(s), 1) if s else None, sclk_mhz_min=min(s) if s else None,
                    sclk_mhz_max=max(s) if s else None, sclk_samples=len(s),
                    source="sysfs pp_dpm_sclk, 50 ms period, during the headline timed region")


def warm_clocks(fn, warm_ms=80.0):
    """Run fn(i) back to back for ~warm_ms of GPU time before a kernel is timed: after a host-side gap (building inputs,
    packing weights) the core clock takes tens of ms of load to come back up -- measured with scripts/cadence_probe.py on the
    512x512 GEMM stage: 158 us per launch in a 40-launch window that follows 64 warm-up launches (9 ms), 141-145 us in the next
    windows, 139-142 us over 200 launches.  Returns the number of launches issued (so that fn's alternation continues)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(8):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) / 8, 1e-3)
    n = 8 + int(min(4000, warm_ms / per))
    for i in range(8, n):
        fn(i)
    return n


def gemm_stage_roofline(dev, sd, g0h, g0w, iters, clocks=None):
    """The dominant kernel of a frame whose global generator G0 runs on g0h x g0w, timed live with HIP events on the
    stream it is launched on (torch's current stream); returns the `roofline` object.  The kernel timed is the one the
    frames RUN: t2v_generator_forward announces its second stream to the library (overlap hint), which may then prefer a
    form that leaves that stream wave slots (256x128 tiles on one block per CU: two 512x512 images in lock-step, and one
    with T2V_OVERLAP_HINT_SINGLE=1) -- so the hint is set here too; `alone_best` is the same stage without the hint (the
    form a single-stream caller gets), when the two differ."""
    from text2video_amd import ops
    two_streams = os.environ.get("T2V_STREAMS", "") != "1"
    prev = ops.set_overlap_hint(two_streams)
    try:
        r = _gemm_stage_roofline(dev, sd, g0h, g0w, iters, clocks)
    finally:
        ops.set_overlap_hint(prev)
    if two_streams:
        alone = _gemm_stage_roofline(dev, sd, g0h, g0w, iters)
        if alone["kernel"] != r["kernel"]:
            r["alone_best"] = {"kernel": alone["kernel"].split(" as ")[0], "ms_per_launch": alone["ms_per_launch"], "frac": alone["frac"]}
    return r


def _gemm_stage_roofline(dev, sd, g0h, g0w, iters, clocks=None):
    import contextlib
    from text2video_amd import ops
    # ---- dominant kernel, timed live with HIP events on the stream it is launched on ----
    # The 1024->1024 3x3 ResnetBlock conv (28 per frame, 84 % of the algorithmic FLOPs) runs as Winograd
    # F(4x4,3x3) [F(2x2,3x3)]: input transform -> 36 [16] batched GEMMs [T x 1024] x [1024 x 1024] on the
    # implicit-GEMM kernel -> output transform.  The GEMM launch is the dominant kernel; its roofline is
    # priced on the MFMA FLOPs it EXECUTES (2*36*T*C*C = 1/4 [4/9] of the direct conv's algorithmic
    # 2*9*C*C*H*W), never on the direct-conv-equivalent figure.  Where the geometry has no Winograd path the direct kernel is timed.
    C, hb, wb = 1024, g0h // 8, g0w // 8
    direct = ops.conv_desc(hb, wb, C, C, 3, 1, 1, ops.PAD_REFLECT)
    # same selection as the generator (generator.hip enumerate_layers): F(4x4,3x3) > F(2x2,3x3) > direct
    cap = int(os.environ.get("T2V_CONV_ALGO", "0"))
    algo = ops.ALGO_DIRECT
    if cap == 0 and ops.winograd_supported(direct, C, ops.ALGO_WINOGRAD_F4):
        algo = ops.ALGO_WINOGRAD_F4
    elif cap in (0, 2) and ops.winograd_supported(direct, C, ops.ALGO_WINOGRAD):
        algo = ops.ALGO_WINOGRAD
    use_wino = algo != ops.ALGO_DIRECT
    wm = 4 if algo == ops.ALGO_WINOGRAD_F4 else 2
    npos, ntile = (wm + 2) ** 2, -(-hb // wm) * -(-wb // wm)     # (ragged maps: the edge tiles count)
    desc = ops.conv_desc(hb, wb, C, C, 3, 1, 1, ops.PAD_REFLECT, algo=algo) if use_wino else direct
    # inputs as the kernel sees them in a frame: conv1 of a ResnetBlock reads the residual stream
    # (dense, signed), conv2 reads a ReLU output (half zeros) -- alternate the two
    xs = [torch.randn(hb, wb, C, device=dev), torch.relu(torch.randn(hb, wb, C, device=dev))]
    wt = ops.pack_conv_weight(sd["model_res_img.0.conv_block.1.weight"].to(dev), desc, C)
    bias = sd["model_res_img.0.conv_block.1.bias"].to(dev)
    stats = ops.conv_stats_buffer(desc, dev)
    y = torch.empty(hb, wb, C, device=dev)
    if use_wino:
        wss = [ops.winograd_workspace(desc, C, dev) for _ in range(2)]
        for i in range(2):   # transformed inputs of both kinds, one workspace each
            ops.conv2d_winograd(xs[i], wt, bias, desc, stats=stats, out=y, workspace=wss[i], stages=1)

        def launch(i, stages=2):
            ops.conv2d_winograd(xs[i & 1], wt, bias, desc, stats=stats, out=y, workspace=wss[i & 1], stages=stages)
    else:
        def launch(i, stages=0):
            ops.conv2d(xs[i & 1], wt, bias, desc, y_cs=C, stats=stats, out=y)
    warm_clocks(launch)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    # the core clock DURING the kernel loop (sysfs, 10 ms period; the loop is repeated until the sampler has seen >= 60 ms of
    # it -- the timed figure is the last repetition's): `frac` is against the 2.4 GHz spec peak, `frac_at_sclk` against the
    # peak at the clock this box actually ran the loop at, so that a slow box and a slow kernel read differently
    ksamp = clocks.fork(0.01) if clocks is not None else None
    with (ksamp if ksamp is not None else contextlib.nullcontext()):
        for _rep in range(6):
            e0.record()
            for i in range(iters):
                launch(i)
            e1.record()
            if ksamp is None or ksamp.path is None:
                break
            torch.cuda.synchronize()
            if len(ksamp.samples) >= 6:
                break
    for i in range(iters):   # the whole conv (all three stages / the direct kernel)
        launch(i, 7)
    e2.record()
    # the same launches bracketed one by one (an event pair per launch: the kernel's own duration, without the
    # command processor's gap between two dependent launches of a queue that the back-to-back cadence above includes)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a0, a1) in enumerate(pairs):
        a0.record()
        launch(i)
        a1.record()
    torch.cuda.synchronize()
    k_ms_bracketed = sum(a0.elapsed_time(a1) for a0, a1 in pairs) / iters
    k_ms = e0.elapsed_time(e1) / iters
    conv_ms = e1.elapsed_time(e2) / iters
    conv_flop = 2.0 * 9 * C * C * hb * wb                       # algorithmic (direct-conv) FLOPs of the layer
    k_flop = 2.0 * npos * ntile * C * C if use_wino else conv_flop
    achieved = k_flop / (k_ms * 1e-3) / 1e12
    traffic = None
    # which form the GEMM stage takes here (the library's own answer: t2v_conv_winograd_gemm_form)
    form = ops.winograd_gemm_form(desc) if algo == ops.ALGO_WINOGRAD_F4 else ""
    fixed_grid = form.startswith("wino_gemm_sk")
    prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get({0: "conv_igemm_rb_hbm_bytes_per_launch",
                                                 1: "winograd_f2_gemm_hbm_bytes_per_launch",
                                                 2: ("winograd_f4_gemm_sk_256x128_hbm_bytes_per_launch" if "256x128" in form else
                                                     "winograd_f4_gemm_sk_hbm_bytes_per_launch" if fixed_grid else
                                                     "winograd_f4_gemm_hbm_bytes_per_launch")}[algo]
                                                if (hb, wb) == (64, 64) else
                                                {(64, 40): "winograd_f4_gemm_512x320_hbm_bytes_per_launch",
                                                 (64, 85): "winograd_f4_gemm_512x680_hbm_bytes_per_launch"}.get((hb, wb), "-"))
        except Exception:
            traffic = None
    kname = ((form.replace(">", ",fp32 32x32x2>") if algo == ops.ALGO_WINOGRAD_F4 else "conv_igemm_kernel<fp32 32x32x2>")
             + " as %d batched GEMMs [%d x 1024]x[1024 x 1024]: Winograd F(%dx%d,3x3) stage of the 1024->1024 3x3 "
               "ResnetBlock conv @%dx%d" % (npos, ntile, wm, wm, hb, wb)
             if use_wino else
             "conv_igemm_kernel<128x128,fp32 32x32x2,reflect,stats> 1024->1024 3x3 @%dx%d" % (hb, wb))
    sclk = ksamp.mean_mhz() if ksamp is not None else None
    roofline = {"bound": "mfma", "kernel": kname,
                "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                "sclk_mhz": sclk, "sclk_samples": len(ksamp.samples) if ksamp is not None else 0,
                "frac_at_sclk": round(achieved / (PEAK_FP32_MFMA_TFLOPS * sclk / SPEC_SCLK_MHZ), 4) if sclk else None,
                "traffic": traffic,
                "traffic_source": "profiles/pmc_summary.json (rocprofv3 --pmc passes; not measured by this run)" if traffic is not None else None,
                "ms_per_launch": round(k_ms, 4), "ms_per_launch_bracketed": round(k_ms_bracketed, 4),
                "gflop_per_launch": round(k_flop / 1e9, 2),
                "flops_counted": "executed MFMA FLOPs of the launch" if use_wino else "algorithmic conv FLOPs",
                "layer": {"algo": "winograd_f%dx%d_3x3" % (wm, wm) if use_wino else "direct", "ms_per_conv": round(conv_ms, 4),
                          "algorithmic_gflop": round(conv_flop / 1e9, 2),
                          "algorithmic_tflops": round(conv_flop / (conv_ms * 1e-3) / 1e12, 2)}}
    return roofline


def build_models(dev, flow, scales):
    """configs[1] / configs[3] generators with seeded random-init weights -> (Vid2VidModelG, [state dicts])"""
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=not flow, norm="batch")
    sd = synthetic_state_dict(spec, seed=1, flow_gain=0.1)
    nets, sds = [HipGenerator(spec, dev).load_state_dict(sd)], [sd]
    if scales == 2:
        spec1 = GeneratorSpec(ngf=64, n_blocks=3, no_flow=not flow, norm="batch", is_local=True, scale=1)
        sd1 = synthetic_state_dict(spec1, seed=2, flow_gain=0.1)
        nets.append(HipGenerator(spec1, dev).load_state_dict(sd1))
        sds.append(sd1)
    return Vid2VidModelG(nets), sds


def time_frames(model, dev, H, W, K, Wm, seed=0):
    """The headline's step (window packing from resident uint8 maps -> generator -> FIFO shift -> tensor2im) for ONE
    sequence on this process's GPU, no collectives: Wm untimed frames, K timed ones -> seconds."""
    from text2video_amd import ops
    from text2video_amd.generator import Recurrence
    poses = torch.from_numpy(synthetic_pose_u8(K + Wm + 2, H, W, seed)).to(dev)
    window = torch.zeros(H, W, 12, dtype=torch.float32, device=dev)
    frames = torch.empty(K, H, W, 4, dtype=torch.uint8, device=dev)
    st = [Recurrence()]

    def step(t, slot):
        for f in range(3):
            ops.pose_u8_to_f32(poses[t + f], window, 3 * f)
        u8 = ops.tensor2im_u8(model.inference_nhwc_batch([window], st)[0])
        if slot is not None:
            frames[slot].copy_(u8)
    for t in range(Wm):
        step(t, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        step(Wm + t, t)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured for a float4 copy)


def hires_hbm_rows(dev, iters):
    """configs[3]'s HBM-bound kernels (north_star: "HBM-bound conv tiles"; SURVEY 8d: the 7x7 stems / heads, the norm
    apply and the local enhancer's 64 / 128-channel full- and half-resolution convs), each timed alone with HIP events at
    its 1024x1024 shape: achieved rate = ALGORITHMIC bytes (every input and output element once, 4 B; weights are KBs)
    / launch time, as a fraction of the 8 TB/s HBM peak.  (The 1024-channel bottleneck of the single-scale generator is the
    MFMA-bound `gemm_stage` next to this list.)"""
    from text2video_amd import ops
    rows = []

    def timed(fn):
        warm_clocks(lambda i: fn(), 40.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(10, iters // 5)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def conv_row(name, H, W, cin, cout, k, stride, pad, pm, transposed, stats, act=0):
        desc = ops.conv_desc(H, W, cin, cout, k, stride, pad, pm, transposed, act)
        xcs, ycs = ops.round_up(cin, 4), ops.round_up(cout, 4)
        x = torch.randn(H, W, xcs, device=dev)
        w = torch.randn(*((cin, cout, k, k) if transposed else (cout, cin, k, k)), device=dev) * 0.02
        pw, b = ops.pack_conv_weight(w, desc, xcs), torch.zeros(cout, device=dev)
        sb = ops.conv_stats_buffer(desc, dev) if stats else None
        ho, wo = ops.conv_out_dims(desc)
        y = torch.empty(ho, wo, ycs, device=dev)
        ms = timed(lambda: ops.conv2d(x, pw, b, desc, y_cs=ycs, stats=sb, out=y))
        nbytes = 4.0 * (H * W * cin + ho * wo * cout)
        flop = 2.0 * k * k * cin * cout * (H * W if transposed else ho * wo)
        rows.append([name, round(ms, 3), round(nbytes / 1e6), round(nbytes / ms / 1e6), round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 3),
                     round(flop / ms / 1e9, 1)])
        del x, y, pw

    R, Z = ops.PAD_REFLECT, ops.PAD_ZERO
    # (names: generator, layer, channels @ input size; stems / down / ResnetBlock / up carry the norm-statistics epilogue)
    conv_row("G0 stem7x7 9>128@1024", 1024, 1024, 9, 128, 7, 1, 3, R, False, True)
    conv_row("G0 head7x7 128>3@1024", 1024, 1024, 128, 3, 7, 1, 3, R, False, False, ops.ACT_TANH)
    conv_row("G1 stem7x7 9>64@1024", 1024, 1024, 9, 64, 7, 1, 3, R, False, True)
    conv_row("G1 down3x3s2 64>128@1024", 1024, 1024, 64, 128, 3, 2, 1, Z, False, True)
    conv_row("G1 res3x3 128>128@512", 512, 512, 128, 128, 3, 1, 1, R, False, True)
    conv_row("G1 upT3x3s2 128>64@512", 512, 512, 128, 64, 3, 2, 1, Z, True, True)
    conv_row("G1 head7x7 64>3@1024", 1024, 1024, 64, 3, 7, 1, 3, R, False, False, ops.ACT_TANH)
    # norm apply + ReLU on the largest activation of the frame
    x = torch.randn(1024, 1024, 128, device=dev)
    y = torch.empty_like(x)
    mr = torch.stack([torch.zeros(128, device=dev), torch.ones(128, device=dev)], 1).contiguous()
    ms = timed(lambda: ops.instance_norm_apply(x, mr, None, None, relu=True, out=y))
    nbytes = 2.0 * x.numel() * 4
    rows.append(["norm apply+ReLU 128@1024", round(ms, 3), round(nbytes / 1e6), round(nbytes / ms / 1e6),
                 round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 3), 0.0])
    del x, y
    torch.cuda.empty_cache()
    return {"columns": "kernel, ms, algorithmic MB, GB/s, frac of 8000 GB/s HBM peak, algorithmic TFLOP/s (fp32 MFMA peak 157.3)",
            "rows": rows}


def hires_block(dev, K, Wm, iters, clocks=None):
    """BASELINE configs[3]: 1024x1024 frames, single-scale G0@1024^2 and the two-scale G0@512^2 + local enhancer G1@1024^2
    (16 frames), flow / no flow, + the GEMM stage of the single-scale generator timed live."""
    out = {"workload": "configs[3]: 1024x1024, %d frames after %d warm-up" % (K, Wm)}
    for scales, name in ((1, "single_scale"), (2, "two_scale")):
        blk = {}
        for flow in (True, False):
            model, sds = build_models(dev, flow, scales)
            el = time_frames(model, dev, 1024, 1024, K, Wm, seed=7)
            gf = (gflop_per_frame(512, 512, flow) + local_gflop_per_frame(1024, 1024, flow)) if scales == 2 \
                else gflop_per_frame(1024, 1024, flow)
            key = "flow" if flow else "noflow"
            blk[key + "_fps"] = round(K / el, 2)
            blk[key + "_ms"] = round(1e3 * el / K, 2)
            blk[key + "_algorithmic_tflops"] = round(K / el * gf / 1e3, 1)
            if flow and scales == 1:
                r = gemm_stage_roofline(dev, sds[0], 1024, 1024, iters, clocks)
                blk["gemm_stage"] = {k: r[k] for k in ("kernel", "ms_per_launch", "gflop_per_launch", "achieved", "frac", "sclk_mhz",
                                                       "frac_at_sclk")}
                blk["gemm_stage"]["kernel"] = r["kernel"].split(" as ")[0] + " @128x128"
            del model, sds
            torch.cuda.empty_cache()
        out[name] = blk
    out["hbm_bound"] = hires_hbm_rows(dev, iters)
    return out


def train_block(dev, dist_mod, world, rank, backend, steps, iters, ngf=128, clocks=None):
    """BASELINE configs[4], the work of ONE GPU: Vid2VidTrainer.train_step on 2 frames of 512x512 (max_frames_per_gpu 2, one
    sequence per GPU), generator WITH its flow branch + 2-scale PatchGAN D + face D, LSGAN + feature matching + flow / warp /
    weight losses against the zero reference flow, --no_vgg, fused Adam, bucketed gradient exchange (SURVEY 8d config 5;
    /root/reference/README.md:171-176).  The exchange is timed as step_with - step_without on the process group that is up
    (world > 1: the real RCCL all-reduce; one GPU: a 1-rank RCCL group, i.e. the collective's launch and device-copy cost
    without any wire time)."""
    import torch.distributed as dist
    from text2video_amd import ops
    from text2video_amd import train as T
    from text2video_amd.options import TrainOptions
    own_group = False
    if not (dist.is_available() and dist.is_initialized()):
        from text2video_amd import launch
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(launch.free_port())
        import datetime
        from text2video_amd import distributed as D
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=D.dist_timeout_s()))
        own_group = True
    os.environ["T2V_TRAIN_FORCE_DIST"] = "1"      # GradBuckets: run the collectives on a 1-rank group as well
    H = W = 512
    F = 2
    opt = TrainOptions().parse(["--name", "bench", "--dataset_mode", "pose", "--input_nc", "3", "--openpose_only", "--num_D", "2",
                                "--max_frames_per_gpu", str(F), "--n_scales_temporal", "0", "--no_first_img", "--fineSize", str(H),
                                "--no_vgg", "--add_face_disc", "--ngf", str(ngf)])
    tr = T.Vid2VidTrainer(opt, str(dev), seed=1)
    rng = np.random.default_rng(100 + rank)        # every rank its own clip
    pose = torch.zeros(F, H, W, 12, device=dev)
    pose[..., :9] = torch.from_numpy(np.where(rng.random((F, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F, H, W, 9)), -1.0)
                                     .astype(np.float32)).to(dev)
    real = torch.zeros(F, H, W, 4, device=dev)
    real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F, H, W, 3)).astype(np.float32))).to(dev)
    real_prev = torch.cat([real[1:], real[:1]], 0).contiguous()
    side = max(8, H // 32 * 8)
    boxes = [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * F
    prev = torch.zeros(1, H, W, 8, device=dev)
    prev[..., :6] = torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 6)).astype(np.float32))).to(dev)

    warm = 2 if steps >= 3 else 1

    sclk = {}
    mean_ms = {}

    def timed(n, exchange):
        import contextlib
        tr.bucketsG.exchange = tr.bucketsD.exchange = exchange
        for _ in range(warm):
            tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)
        torch.cuda.synchronize()
        dist.barrier()
        # one more untimed step BEHIND the barrier: the first step that sends its buckets off after a barrier takes 70-100 ms
        # longer than the rest (191 / 157 ms against 87-88 for every later one, per-step timings on one box), which a mean
        # over 5 timed steps showed as 93-110 ms per step in some runs and not in others; with the exchange that step's
        # collectives line the ranks up again
        tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)
        torch.cuda.synchronize()
        samp = clocks.fork(0.02) if clocks is not None else None      # this rank's core clock during the timed steps
        with (samp if samp is not None else contextlib.nullcontext()):
            t0 = time.perf_counter()
            marks = [t0]
            for _ in range(n):
                losses = tr.train_step(pose, real, boxes, prev.clone(), real_prev=real_prev)[0]
                marks.append(time.perf_counter())      # (a step ends with its one host read: the losses)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        # The figure of the block is the MEDIAN step (x n): with the exchange on, one step shortly after a barrier takes 50-100 ms
        # longer than the others (per-step marks on one box: 86.9 137.2 84.8 84.6 84.4 ms; without the exchange 84.5 83.4 83.9
        # 83.9 83.2) -- something in the process group's housekeeping, not in the step -- and a mean over 5 steps read 85 ms in
        # one run and 93-110 in the next.  The mean is reported beside it.
        per_step = sorted(b - a for a, b in zip(marks[:-1], marks[1:]))
        mean_ms[exchange] = 1e3 * el / n
        el = n * (per_step[n // 2] if n % 2 else 0.5 * (per_step[n // 2 - 1] + per_step[n // 2]))
        if samp is not None:
            sclk[exchange] = samp.mean_mhz()
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return 1e3 * el / n, losses

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):          # (the zero-reference-flow warning)
        ms_with, losses = timed(steps, True)
    nbytes, nbuckets = tr.comm_bytes, len(tr.bucketsG.bounds) + len(tr.bucketsD.bounds)
    in_sync = None
    if world > 1:     # same seed, same averaged gradients: the replicas' weights must still be equal
        chk = torch.stack([p.detach().double().sum() for p in tr.optG.params + tr.optD.params]).sum().reshape(1)
        if backend != "nccl":
            chk = chk.cpu()
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        in_sync = all(float(c) == float(allc[0]) for c in allc)
    ms_without, _ = timed(steps, False)     # (after the checksum: without the exchange the replicas drift apart)
    block = None
    if rank == 0:
        # ---- the step's heaviest kernels, live (HIP events on the launch stream), on the FLOPs they execute ----
        kclk = []      # the core clock during each kernel's loop (a sagging clock and a slow kernel read differently)

        def ev_time(fn, reps=3):
            """fastest of `reps` loops of `iters` launches, with the clock sampled during THAT loop: these rows come last in a
            long run, and a loop that meets a power-management dip read 15-20 % low now and then (0.60-0.63 against 0.74 for
            the transposed layer's weight gradient; the same launch measures 0.74 wherever its buffers lie,
            profiles/r06_wgrad_placement_probe.txt)"""
            import contextlib
            warm_clocks(lambda i: fn())
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                samp = clocks.fork(0.01) if clocks is not None else None
                with (samp if samp is not None else contextlib.nullcontext()):
                    e0.record()
                    for _ in range(iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                if best is None or ms < best[0]:
                    best = (ms, samp.mean_mhz() if samp is not None else None)
            kclk.append(best[1])
            return best[0]
        C = 1024
        desc = ops.conv_desc(64, 64, C, C, 3, 1, 1, ops.PAD_REFLECT)
        kernels = []
        if ops.backward_weight_winograd_supported(desc, C, C):
            ws = ops.backward_weight_winograd_workspace(desc, C, F, dev)
            ops.conv2d_backward_weight_winograd_stages(torch.randn(F, 64, 64, C, device=dev), torch.randn(F, 64, 64, C, device=dev),
                                                       desc, ws, F, 0, False)
            dw = torch.empty(C, C, 3, 3, device=dev)
            ms = ev_time(lambda: ops.conv2d_backward_weight_winograd_reduce(desc, ws, F, C, C, out=dw))
            gf = 2.0 * 36 * F * 256 * C * C / 1e9
            kernels.append({"kernel": "wino_wgrad_sk + dW transform, 1024->1024 conv, %d frames" % F,
                            "ms_per_launch": round(ms, 4), "gflop_per_launch": round(gf, 2),
                            "achieved": round(gf / ms, 2), "frac": round(gf / ms / PEAK_FP32_MFMA_TFLOPS, 4)})
        for name, (h, w, ci, co, st, tr_) in (("down 512->1024 3x3 s2 @128x128", (128, 128, 512, 1024, 2, False)),
                                              ("up 1024->512 convT 3x3 s2 @64x64", (64, 64, 1024, 512, 2, True))):
            d = ops.conv_desc(h, w, ci, co, 3, st, 1, ops.PAD_ZERO, tr_)
            ho, wo = ops.conv_out_dims(d)
            # as the step launches it: the clip's two frames from buffers of their own in ONE launch (train._paired_direct_wgrad)
            xs = [torch.randn(h, w, ci, device=dev) for _ in range(2)]
            dys = [torch.randn(ho, wo, co, device=dev) for _ in range(2)]
            ms = ev_time(lambda: ops.conv2d_backward_weight_pair(xs[0], dys[0], xs[1], dys[1], d))
            gf = 2 * 2.0 * 9 * ci * co * (h * w if tr_ else ho * wo) / 1e9
            kernels.append({"kernel": "conv_wgrad, 2 frames: " + name,
                            "ms_per_launch": round(ms, 4), "gflop_per_launch": round(gf, 2),
                            "achieved": round(gf / ms, 2), "frac": round(gf / ms / PEAK_FP32_MFMA_TFLOPS, 4)})
        block = {"workload": "configs[4] per GPU: 512x512, 2 frames, G (flow%s) + D (num_D 2) + face D, --no_vgg, Adam; %d GPU(s)"
                             % ("" if ngf == 128 else ", ngf %d: NOT configs[4], plumbing test only" % ngf, world),
                 "ms_per_step": round(ms_with, 2), "steps": steps, "warmup": warm + 1,
                 "statistic": "median of the timed steps (max over ranks); mean incl. the closing barrier: %.2f with / %.2f without "
                              "the exchange" % (mean_ms.get(True, 0.0), mean_ms.get(False, 0.0)),
                 # rank 0's core clock during the two timed regions, and the step scaled to the clock the round-4/5 targets
                 # were quoted at (ms x sclk / 2360: the step is GPU-bound, its kernels MFMA-bound)
                 "sclk_mhz": sclk.get(False), "sclk_mhz_with_exchange": sclk.get(True),
                 "ms_per_step_without_at_%dmhz" % REF_SCLK_MHZ:
                     round(ms_without * sclk[False] / REF_SCLK_MHZ, 2) if sclk.get(False) else None,
                 "exchange": {"group": "%d-rank %s" % (world, "rccl" if backend == "nccl" or own_group else backend),
                              "ms_per_step_with": round(ms_with, 2), "ms_per_step_without": round(ms_without, 2),
                              "ms": round(ms_with - ms_without, 2), "bytes": int(nbytes), "buckets": nbuckets,
                              "collective": "reduce_scatter+all_gather" if tr.bucketsG.rs_ag else "all_reduce(avg)",
                              "replicas_in_sync": in_sync},
                 "losses": {k: round(float(v), 3) for k, v in losses.items() if k in ("G_GAN", "G_GAN_Feat", "D", "D_f")},
                 "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                 "kernels": [dict(k, sclk_mhz=c) for k, c in zip(kernels, kclk)],
                 "kernels_how": "per row: fastest of 3 loops of %d launches, sclk_mhz sampled during that loop" % iters}
    del tr
    torch.cuda.empty_cache()
    if own_group:
        dist.destroy_process_group()
    return block


def run_e2e(model_head, model_other, head_flow, n_frames):
    """The drop-in test.py path end to end: a dataset in the layout the reference's L2 driver writes (OpenPose JSONs
    + skeleton jpgs; the committed fadg0 keypoint fixtures, cycled), rasterised by the pose-dataset worker pool ->
    uint8 H2D -> generator -> tensor2im -> pinned D2H -> JPEG files.  Full-size generator, seeded weights."""
    import shutil
    import tempfile
    from PIL import Image
    from text2video_amd.keypoints import read_keypoints
    from text2video_amd.model import run_test
    from text2video_amd.options import TestOptions
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    out, meta = [], {}
    models = {head_flow: model_head}
    if model_other is not None:
        models[not head_flow] = model_other
    # (canvas W x H of the source frames, extra flags, geometry the generator sees, which variants, sequence folders)
    one = ["tmp"]
    two = ["tmp", "tmp_smooth"]       # what the reference's L2 driver writes per utterance (text2video_audio.sh:24-31)
    cases = [((512, 512), ["--no_pose_crop", "--batch_sequences", "1"], "512x512", [True, False], one),
             ((512, 384), ["--no_pose_crop", "--batch_sequences", "1"], "512x680 (fadg0 full width)",
              [head_flow], one),
             ((512, 384), ["--batch_sequences", "1"], "512x320 (fadg0 cropped)",
              [head_flow], one),
             ((512, 512), ["--no_pose_crop", "--batch_sequences", "1"], "512x512, 2 sequences",
              [head_flow], two),
             ((512, 512), ["--no_pose_crop", "--batch_sequences", "2"], "512x512, 2 sequences in lock-step",
              [head_flow], two),
             # the reference's own run: fadg0 frames, scaleHeight 512 + central-width crop, tmp and tmp_smooth
             ((512, 384), ["--batch_sequences", "1"], "512x320, 2 sequences", [head_flow], two),
             ((512, 384), ["--batch_sequences", "2"], "512x320, 2 sequences in lock-step", [head_flow], two)]
    for canvas, extra, geom, flows, seqs in cases:
        tmp = tempfile.mkdtemp(prefix="t2v_e2e_")
        try:
            root = os.path.join(tmp, "datasets", "fadg0")
            img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), canvas))
            per_seq = n_frames // len(seqs)
            for q, seq in enumerate(seqs):
                os.makedirs(os.path.join(root, "test_openpose", seq))
                os.makedirs(os.path.join(root, "test_img", seq))
                for i in range(per_seq + 2):
                    shutil.copyfile(os.path.join(src, files[(i + 7 * q) % len(files)]),
                                    os.path.join(root, "test_openpose", seq, "%05d.json" % i))
                    img.save(os.path.join(root, "test_img", seq, "%04d.jpg" % i))
            for flow in flows:
                if flow not in models:
                    continue
                argv = ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight "
                        "--loadSize 512 --openpose_only --how_many 1200 --no_first_img --random_drop_prob 0 "
                        "--results_dir %s --checkpoints_dir %s"
                        % (root, os.path.join(tmp, "results_%d" % flow), os.path.join(tmp, "ckpt"))).split() + extra
                opt = TestOptions().parse(argv)
                m = models[flow]
                m.reset()
                import contextlib
                import io
                with contextlib.redirect_stdout(io.StringIO()):      # the loop prints "process image..." per frame
                    stats = run_test(opt, model=m, device="cuda:%d" % torch.cuda.current_device())
                from text2video_amd.pose_dataset import default_pose_workers
                workers = opt.pose_workers if opt.pose_workers is not None else default_pose_workers()
                out.append({"geometry": geom, "flow": flow, "fps": round(stats["fps_loop"], 2), "sequences": len(seqs),
                            "batch_sequences": opt.batch_sequences})
                meta["pose_workers"], meta["frames_per_run"] = workers, stats["frames"]
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return dict({"path": "vid2vid/test.py loop: rasterise (bit-exact) -> H2D -> generator -> D2H -> JPEG", "runs": out}, **meta)


UTTERANCE = "She had your dark suit in greasy wash water all year."      # configs[0] (BASELINE.json), fixture tests/golden/l2_inputs


def utterance_block(tmp, ckpt_dir, env):
    """The whole post-alignment utterance as /root/reference/text2video_audio.sh:24-44 chains it -- three processes, each
    started from the directory the script cd's into, on the configs[0] fixture (tests/golden/l2_inputs = the reference's
    Text2Video data layout: time stamps, unit tables, key poses):

        (Text2Video)  python interp_landmarks_motion_phoneme_VidTIMIT_smooth.py "$1" $2          [:31]
        (vid2vid)     python test.py --name $2 --dataroot datasets/$2 ... --random_drop_prob 0      [:42]
        (vid2vid)     python image2video_real_audio_text2video.py "$1" $2                         [:44]

    wall seconds of each (perf_counter around subprocess.run, the script's `rm -f` lines included), run twice -- and of the
    in-memory route (`python -m text2video_amd.pipeline`: one process, no JSON / skeleton-JPEG round trip, frames muxed by the
    same process).  The reference's L2 script alone takes 11.9 s on this fixture (measured in the build container)."""
    import glob
    import shutil
    import subprocess
    person = "fadg0"
    t2v, v2v = os.path.join(tmp, "Text2Video"), os.path.join(tmp, "vid2vid")
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "l2_inputs"), t2v)
    os.makedirs(os.path.join(t2v, "input_audio_real", person))
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "audio", "Shehadyour.mp3"),
                    os.path.join(t2v, "input_audio_real", person, "Shehadyour.mp3"))
    os.makedirs(v2v)
    os.symlink(ckpt_dir, os.path.join(v2v, "checkpoints"))          # test.py's default --checkpoints_dir, as the literal line needs
    flags = ("--name %s --dataroot datasets/%s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
             "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0" % (person, person)).split()
    stages = (("l2_driver_s", t2v, [os.path.join(ROOT, "Text2Video", "interp_landmarks_motion_phoneme_VidTIMIT_smooth.py"), UTTERANCE, person]),
              ("test_py_s", v2v, [os.path.join(ROOT, "vid2vid", "test.py")] + flags),
              ("mux_s", v2v, [os.path.join(ROOT, "vid2vid", "image2video_real_audio_text2video.py"), UTTERANCE, person]))

    def clean():                                                     # the script's rm -f lines (:24-28, :39-40)
        for d in ("datasets/%s/test_openpose" % person, "datasets/%s/test_img" % person, "results/%s/test_latest" % person):
            for seq in ("tmp", "tmp_smooth"):
                for f in glob.glob(os.path.join(v2v, d, seq, "*")):
                    os.remove(f)

    def chain(extra_env=None):
        out = {}
        t_all = time.perf_counter()
        clean()
        for key, cwd, argv in stages:
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable] + argv, cwd=cwd, env=dict(env, **(extra_env or {})), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True)
            out[key] = round(time.perf_counter() - t0, 3)
            if r.returncode != 0:
                return {"error": "%s: %s" % (key, r.stderr[-300:])}
        out["wall_s"] = round(time.perf_counter() - t_all, 3)
        return out

    runs = [chain(), chain()]
    if any("error" in r for r in runs):
        return {"error": [r.get("error") for r in runs]}
    frames = len(glob.glob(os.path.join(v2v, "results", person, "test_latest", "*", "fake_B_*.jpg")))
    videos = sorted(os.path.basename(f) for f in glob.glob(os.path.join(v2v, "results", person, "*.mp4")))
    # one process: L2 in memory -> frame loop -> mux (text2video_amd/pipeline.py)
    mem = []
    for _ in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "text2video_amd.pipeline", UTTERANCE, person, "--l2_root", "../Text2Video",
                            "--results_dir", os.path.join(tmp, "results_mem"), "--write_video", "--video_audio",
                            os.path.join(t2v, "input_audio_real", person, "Shehadyour.mp3")], cwd=v2v,
                           env=dict(env, PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", "")), stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            mem = {"error": r.stderr[-300:]}
            break
        mem.append(round(time.perf_counter() - t0, 3))
    # the unchanged three-process chain with the weights resident between utterances (T2V_RESIDENT=1 in the environment of
    # the unchanged script): the call that starts the server, then a warm utterance
    renv = {"T2V_RESIDENT": "1", "T2V_RESIDENT_KEY": "bench-utt-%d" % os.getpid()}
    try:
        rfirst, rwarm = chain(renv), chain(renv)
    finally:
        subprocess.run([sys.executable, os.path.join(ROOT, "vid2vid", "test.py")] + flags + ["--resident_stop"], cwd=v2v,
                       env=dict(env, **renv), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    last = runs[-1]
    return {"what": "text2video_audio.sh:24-44 on the configs[0] utterance: L2 driver | test.py | image2video, 3 processes",
            "frames": frames, "videos": len(videos), "chain_runs": runs, "chain_wall_s": last["wall_s"],
            "l2_plus_mux_s": round(last["l2_driver_s"] + last["mux_s"], 3),
            "l2_plus_mux_below_test_py": bool(last["l2_driver_s"] + last["mux_s"] < last["test_py_s"]),
            "in_memory_pipeline_wall_s": mem, "chain_resident_warm": rwarm, "chain_resident_first_wall_s": rfirst.get("wall_s"),
            "reference_l2_driver_s": 11.9}


def cold_start_block(n_maps=87, ab_envs=None, ab_reps=3, wrap=None, utterance=False):
    """The reference starts one process per utterance (text2video_audio.sh:37-44: `cd ../vid2vid; python test.py ...`): the
    wall time of exactly that command on the configs[0] utterance -- two sequences (tmp, tmp_smooth) of 87 pose maps = 2 x 85
    frames, 512x384 sources -> scaleHeight 512 + central crop = 512x320, full-size generator with its flow branch read from a
    1.46 GB latest_net_G0.pth -- measured from outside (subprocess), twice, with the split the process reports itself
    (--timing_json)."""
    import shutil
    import subprocess
    import tempfile
    from PIL import Image
    from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
    from text2video_amd.keypoints import read_keypoints
    src = os.path.join(ROOT, "tests", "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(src) if f.startswith("sa1_"))
    tmp = tempfile.mkdtemp(prefix="t2v_cold_")
    try:
        root = os.path.join(tmp, "datasets", "fadg0")
        img = Image.fromarray(read_keypoints(os.path.join(src, files[0]), (512, 384)))
        for q, (seq, stem) in enumerate((("tmp", "%04d.jpg"), ("tmp_smooth", "smooth_%04d.jpg"))):
            os.makedirs(os.path.join(root, "test_openpose", seq))
            os.makedirs(os.path.join(root, "test_img", seq))
            for i in range(n_maps):
                shutil.copyfile(os.path.join(src, files[(i + 7 * q) % len(files)]),
                                os.path.join(root, "test_openpose", seq, "%05d.json" % i))
                img.save(os.path.join(root, "test_img", seq, stem % i))
        os.makedirs(os.path.join(tmp, "ckpt", "fadg0"))
        spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=False, norm="batch")
        torch.save(synthetic_state_dict(spec, seed=1, flow_gain=0.1), os.path.join(tmp, "ckpt", "fadg0", "latest_net_G0.pth"))
        tj = os.path.join(tmp, "timing.json")
        cmd = [sys.executable, os.path.join(ROOT, "vid2vid", "test.py")] + \
            ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
             "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0 --results_dir %s --checkpoints_dir %s "
             "--timing_json %s" % (root, os.path.join(tmp, "results"), os.path.join(tmp, "ckpt"), tj)).split()
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        walls, split = [], None
        for _ in range(2):
            shutil.rmtree(os.path.join(tmp, "results"), ignore_errors=True)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=os.path.join(ROOT, "vid2vid"), env=env, stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True)
            walls.append(round(time.perf_counter() - t0, 3))
            if r.returncode != 0:
                return {"error": r.stderr[-400:]}
            split = json.load(open(tj))
        if wrap:          # the command under a wrapper (e.g. rocprofv3 --kernel-trace --stats -d DIR --): scripts/, not the bench line
            shutil.rmtree(os.path.join(tmp, "results"), ignore_errors=True)
            r = subprocess.run(list(wrap) + cmd, cwd=os.path.join(ROOT, "vid2vid"), env=dict(env, T2V_NO_FAST_EXIT="1"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True)
            return {"wrapped_rc": r.returncode, "stderr_tail": r.stderr[-300:], "cold_start": json.load(open(tj))["cold_start"]}
        ab = None
        if ab_envs:       # same-box A/B of environment variants of the plain command (scripts/; not part of the bench line)
            ab = {name: [] for name in ab_envs}
            for _ in range(ab_reps):
                for name, extra in ab_envs.items():
                    shutil.rmtree(os.path.join(tmp, "results"), ignore_errors=True)
                    t0 = time.perf_counter()
                    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "vid2vid"), env=dict(env, **extra), stdout=subprocess.DEVNULL,
                                       stderr=subprocess.PIPE, text=True)
                    w = round(time.perf_counter() - t0, 3)
                    c = json.load(open(tj))["cold_start"] if r.returncode == 0 else {}
                    ab[name].append({"wall_s": w, "loop_s": c.get("loop_s"), "create_model_s": c.get("create_model_s"),
                                     "first_step_s": c.get("first_step_s")})
            return {"ab": ab}
        cs = split["cold_start"]
        # (the line stays short -- the driver keeps a bounded tail of stdout: the full split is in --timing_json)
        for k in ("dataset_scan_s", "checkpoint_read_s", "pack_s", "mux_s"):
            cs.pop(k, None)
        if isinstance(cs.get("upload"), dict):
            cs["upload"] = {k: cs["upload"].get(k) for k in ("threads", "total_s", "mirrored")}
        if isinstance(cs.get("loop_split"), dict):
            cs["loop_split"] = {k: cs["loop_split"].get(k) for k in ("wait_pose_s", "upload_s", "enqueue_s", "finish_s")}
        # the same command with torch as the frame loop's allocator / stream provider (T2V_LEAN=0: what it was before the
        # torch-free loop of text2video_amd/leantorch.py)
        with_torch = None
        shutil.rmtree(os.path.join(tmp, "results"), ignore_errors=True)
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=os.path.join(ROOT, "vid2vid"), env=dict(env, T2V_LEAN="0"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True)
        if r.returncode == 0:
            tcs = json.load(open(tj))["cold_start"]
            with_torch = {"wall_s": round(time.perf_counter() - t0, 3), "process_to_run_test_s": tcs.get("process_to_run_test_s"),
                          "loop_s": tcs.get("loop_s")}
        # the same command as a client of the resident server (--resident: weights stay on the GPU between utterances):
        # the call that starts the server, then a warm one
        renv = dict(env, T2V_RESIDENT_KEY="bench-%d" % os.getpid())
        rwalls, rloop = [], None
        try:
            for _ in range(2):
                shutil.rmtree(os.path.join(tmp, "results"), ignore_errors=True)
                t0 = time.perf_counter()
                r = subprocess.run(cmd + ["--resident"], cwd=os.path.join(ROOT, "vid2vid"), env=renv, stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, text=True)
                rwalls.append(round(time.perf_counter() - t0, 3))
                if r.returncode != 0:
                    rwalls = None
                    break
                rloop = json.load(open(tj))["cold_start"]["loop_s"]
        finally:
            subprocess.run(cmd + ["--resident_stop"], cwd=os.path.join(ROOT, "vid2vid"), env=renv, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
        resident = None if not rwalls else {"first_call_wall_s": rwalls[0], "warm_call_wall_s": rwalls[1], "warm_loop_s": rloop,
                                            "warm_wall_over_loop": round(rwalls[1] / max(rloop, 1e-9), 2)}
        utt = utterance_block(tmp, os.path.join(tmp, "ckpt"), env) if utterance else None
        return {"utterance": utt,
                "command": "python vid2vid/test.py <text2video_audio.sh:42 flags> as a subprocess: tmp + tmp_smooth, 2 x %d frames "
                           "512x320, %.2f GB checkpoint" % (n_maps - 2, os.path.getsize(os.path.join(tmp, "ckpt", "fadg0", "latest_net_G0.pth")) / 1e9),
                "frames": split["frames"], "wall_s": walls, "split_of_last_run": cs, "with_torch": with_torch,
                "resident": resident,
                "wall_over_loop": round(walls[-1] / max(cs["loop_s"], 1e-9), 2),
                "to_last_jpeg_over_loop": (round((cs["process_to_run_test_s"] + cs["to_last_jpeg_s"]) / max(cs["loop_s"], 1e-9), 2)
                                           if cs.get("process_to_run_test_s") is not None else None)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=62)   # a 64-pose-map sequence yields 62 frames
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--variant", choices=["flow", "noflow"], default="flow",
                    help="which generator variant is the headline `value` (the other one is timed as well)")
    ap.add_argument("--flow", action="store_true", help="(kept for old command lines) same as --variant flow")
    ap.add_argument("--single-variant", action="store_true", help="time the headline variant only")
    ap.add_argument("--e2e-frames", type=int, default=160, help="frames per end-to-end test.py-path run (0 = skip)")
    ap.add_argument("--scales", type=int, default=1, choices=[1, 2],
                    help="2 = config 4's two-scale generator: G0 at H/2 x W/2 + local enhancer G1 at H x W")
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--kernel-iters", type=int, default=200)
    ap.add_argument("--no-cold-start", action="store_true", help="skip e2e.cold_start (the test.py command as a subprocess)")
    ap.add_argument("--hires-frames", type=int, default=16,
                    help="frames per 1024x1024 run of the `hires` block (configs[3]; 0 = skip; default geometry, 1 GPU only)")
    ap.add_argument("--train-ngf", type=int, default=128, help="(tests only: a narrower generator in the train_step block)")
    ap.add_argument("--train-steps", type=int, default=5,
                    help="timed optimiser steps of the `train_step` block (configs[4] per-GPU work; 0 = skip; default geometry only)")
    ap.add_argument("--batch-variants", type=lambda v: [int(x) for x in v.split(",") if x], default=[2, 4],
                    help="lock-step batch sizes timed in addition to the headline (config.variants.batch<N>_fps)")
    args = ap.parse_args()

    from text2video_amd import launch
    if args.gpus > 1 and not launch.under_launcher() and os.environ.get("T2V_DIST_BACKEND", "nccl") == "nccl" \
            and torch.cuda.device_count() < args.gpus:
        # fail before any rank starts: N ranks on fewer than N devices would sit in RCCL's communicator set-up until it times out
        print("bench.py: --gpus %d needs %d visible GPUs, this node shows %d (HIP_VISIBLE_DEVICES=%s)"
              % (args.gpus, args.gpus, torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES", "<unset>")),
              file=sys.stderr, flush=True)
        sys.exit(2)
    # plain `python bench.py --gpus N` (no torchrun environment): run the N ranks ourselves, one per GPU
    launch.fan_out_if_needed(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's world size and --gpus must agree)"
              % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    backend = os.environ.get("T2V_DIST_BACKEND", "nccl")   # "nccl" = RCCL; "gloo": several ranks on one GPU (tests only)
    local_rank = launch.local_device_index(local_rank)
    if backend == "nccl" and torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d of %d computes on device %d, but this process sees %d GPU(s) (HIP_VISIBLE_DEVICES=%s): RCCL needs "
              "one device per rank" % (rank, world, local_rank, torch.cuda.device_count(), os.environ.get("HIP_VISIBLE_DEVICES", "<unset>")),
              file=sys.stderr, flush=True)
        sys.exit(2)
    if world > 1:
        launch.pin_to_numa_node(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = D = None
    if world > 1 or os.environ.get("T2V_BENCH_FORCE_DIST") == "1":   # the env var exercises the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # the job's timeout (T2V_DIST_TIMEOUT_S, default 300 s) on the rendezvous and on every collective, then a roll call
        # that names a rank that never arrived (text2video_amd/distributed.py)
        from text2video_amd import distributed as D
        D.init_group(backend, rank, world, dev if backend == "nccl" else None)

    from text2video_amd import ops
    H, W, K, Wm = args.height, args.width, args.steps, args.warmup
    head_flow = args.variant == "flow"
    default_geometry = (H, W, args.scales) == (512, 512, 1)

    def build(flow):
        return build_models(dev, flow, args.scales)

    nposes = K + Wm + 2
    poses = torch.from_numpy(synthetic_pose_u8(nposes, H, W, seed=rank)).to(dev)   # resident in HBM
    window = torch.zeros(H, W, 12, dtype=torch.float32, device=dev)
    frames = torch.empty(K, H, W, 4, dtype=torch.uint8, device=dev)
    gathered = torch.empty(world * K, H, W, 4, dtype=torch.uint8, device=dev) if dist else None

    def all_gather_frames():
        """chunk outputs to every rank: RCCL all-gather over xGMI (gloo in single-GPU tests: staged through the host)"""
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, frames)
        else:
            host = torch.empty(gathered.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, frames.cpu())
            gathered.copy_(host)

    def timed_run(model, sampler=None, nseq=1):
        """W untimed warm-up frames, then exactly K timed steps (+ the all-gather of the chunk's frames for N>1)
        between barrier + synchronize on both sides; returns the MAX over ranks of the elapsed seconds.
        nseq > 1: every step advances `nseq` independent sequences by one frame each (lock-step batch); the frames of
        sequence 0 are the ones gathered."""
        from text2video_amd.generator import Recurrence
        states = [Recurrence() for _ in range(nseq)]
        windows = [window] + [torch.zeros_like(window) for _ in range(nseq - 1)]
        seq_poses = [poses] + [torch.from_numpy(synthetic_pose_u8(nposes, H, W, seed=1000 * (q + 1) + rank)).to(dev)
                               for q in range(nseq - 1)]

        def step(t, out_slot):
            for q in range(nseq):
                for f in range(3):                       # sliding window of tG = 3 pose maps, oldest first
                    ops.pose_u8_to_f32(seq_poses[q][t + f], windows[q], 3 * f)
            outs = model.inference_nhwc_batch(windows, states)
            u8s = [ops.tensor2im_u8(o) for o in outs]
            if out_slot is not None:
                frames[out_slot].copy_(u8s[0])

        for t in range(Wm):
            step(t, None)
        if dist:    # untimed: RCCL sets up its channels / registers the buffers on the first collective of a shape
            D.rendezvous("warm-up done")      # (a rank that died in its warm-up is named here, not timed out on below)
            all_gather_frames()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        import contextlib
        with (sampler if sampler is not None else contextlib.nullcontext()):
            t0 = time.perf_counter()
            for t in range(K):
                step(Wm + t, t)
            if dist:
                all_gather_frames()
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        if dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    model, sds = build(head_flow)
    sd = sds[0]
    spec = model.nets[0].spec
    sampler = ClockSampler(local_rank)
    elapsed = timed_run(model, sampler)            # the headline: `value`
    other = other_elapsed = None
    batch_elapsed = {}
    if not args.single_variant:
        # the same model advancing 2 / 4 independent sequences in lock-step (t2v_generator_forward_batch): aggregate rate
        # over all sequences, reported in config.variants -- `value` stays the single-sequence figure
        for nb in args.batch_variants:
            batch_elapsed[nb] = timed_run(model, None, nb)
        other, _ = build(not head_flow)
        other_elapsed = timed_run(other)

    result = None
    if rank == 0:
        fps = world * K / elapsed
        if args.scales == 2:   # G0 on the half-resolution pyramid level + the local enhancer
            def gflops(flow):
                return gflop_per_frame(H // 2, W // 2, flow) + local_gflop_per_frame(H, W, flow)
        else:
            def gflops(flow):
                return gflop_per_frame(H, W, flow)
        gf = gflops(head_flow)
        g0h, g0w = (H // 2, W // 2) if args.scales == 2 else (H, W)   # G0 runs on the half-resolution pyramid level
        roofline = gemm_stage_roofline(dev, sd, g0h, g0w, args.kernel_iters, sampler)
        # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
        cpu = None
        if args.cpu_frames > 0 and world == 1:   # reported at N=1 only (rank 0)
            from oracle.generator_ref import CompositeGenerator, Vid2VidInferenceRef
            cores = torch.get_num_threads()
            ref_net = CompositeGenerator(9, 3, 6, 128, 3, 9, spec.no_flow, "batch")
            ref_net.load_state_dict(sd, strict=False)
            ref_nets = [ref_net]
            if args.scales == 2:
                from oracle.generator_ref import CompositeLocalGenerator
                loc = CompositeLocalGenerator(9, 3, 6, 128, 3, 1, spec.no_flow, "batch")
                loc.load_state_dict(sds[1], strict=False)
                ref_nets.append(loc)
            ref = Vid2VidInferenceRef(ref_nets)
            nf = args.cpu_frames
            pf = ((poses[:nf + 3].cpu().float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)
            # the same frames through the measured HIP path, teacher-forced on the oracle's previous frames: the
            # timed model's output against the oracle's, reported next to the speed (outside the timed region)
            model.reset()
            wants = [ref.inference(pf[0:3].unsqueeze(0))]   # warm-up frame (thread pool, first-frame path)
            gots = [model.inference(pf[0:3].unsqueeze(0).to(dev))[0].cpu()]
            state1 = [p.clone() for p in ref.fake_B_prev]
            csec = 0.0
            for t in range(1, 1 + nf):
                model.load_prev(ref.fake_B_prev)
                c0 = time.perf_counter()
                wants.append(ref.inference(pf[t:t + 3].unsqueeze(0)))
                csec += time.perf_counter() - c0
                gots.append(model.inference(pf[t:t + 3].unsqueeze(0).to(dev))[0].cpu())
            parity = {"max_abs_delta_vs_oracle": float("%.3g" % max((g - w).abs().max().item() for g, w in zip(gots, wants))),
                      "frames": len(wants), "tolerance": 1e-3,
                      "how": "frames 0..%d of the sequence, previous frames taken from the oracle (teacher-forced)" % nf}
            # the same frames again with fewer threads (SURVEY 8d asks for an 8-thread figure; torch's CPU convolutions do
            # not scale to all 128 hardware threads of this host -- the fastest setting is the baseline `value`)
            by_threads = {cores: nf / csec}
            for nt in (32, 8):
                if nt >= cores:
                    continue
                torch.set_num_threads(nt)
                ref.fake_B_prev = [p.clone() for p in state1]
                ref.inference(pf[1:4].unsqueeze(0))            # warm-up at the new thread count (not timed)
                ref.fake_B_prev = [p.clone() for p in state1]
                c0 = time.perf_counter()
                for t in range(1, 1 + nf):
                    ref.inference(pf[t:t + 3].unsqueeze(0))
                by_threads[nt] = nf / (time.perf_counter() - c0)
            torch.set_num_threads(cores)
            best = max(by_threads, key=lambda k: by_threads[k])
            cpu = {"value": round(by_threads[best], 4), "unit": "frames/s", "cores": best, "kind": "port",
                   # the box: logical CPUs the host has / this process may run on (`cores` = the thread count that won)
                   "host_logical_cpus": os.cpu_count(), "host_cpus_available": len(os.sched_getaffinity(0)),
                   "sample": "%d frames %dx%d (%s) after 1 warm-up frame, torch %s CPU fp32; fastest of %s threads"
                             % (nf, H, W, "flow branch on" if head_flow else "no flow branch", torch.__version__,
                                "/".join(str(k) for k in sorted(by_threads, reverse=True))),
                   "by_threads": {str(k): round(v, 4) for k, v in sorted(by_threads.items(), reverse=True)},
                   "value_8_threads": round(by_threads[8], 4) if 8 in by_threads else None, "parity": parity}
        # ---- end to end: the drop-in test.py frame loop on a dataset in the reference's layout ----
        e2e = None
        if args.e2e_frames > 0 and world == 1 and args.scales == 1:
            e2e = run_e2e(model, other, head_flow, args.e2e_frames)
            if default_geometry and not args.no_cold_start:
                e2e["cold_start"] = cold_start_block(utterance=True)
                e2e["utterance"] = e2e["cold_start"].pop("utterance", None)
        variants = {("flow_fps" if head_flow else "noflow_fps"): round(fps, 3)}
        if other_elapsed is not None:
            variants["noflow_fps" if head_flow else "flow_fps"] = round(world * K / other_elapsed, 3)
            variants["other_ms_per_step"] = round(1e3 * other_elapsed / K, 3)
            variants["other_algorithmic_gflop_per_frame"] = round(gflops(not head_flow), 1)
        for nb, el in batch_elapsed.items():
            variants["batch%d_fps" % nb] = round(world * nb * K / el, 3)
        if batch_elapsed:
            variants["batch_note"] = "batch<N>_fps: N sequences in lock-step, aggregate"
        variants["headline"] = "flow" if head_flow else "noflow"
        variants["note"] = "both variants timed in this run over K steps; `value` = headline"
        result = {
            "metric": "frames/sec 512x512 pose->RGB (vid2vid generator)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: fadg0 openpose_only %dx%d, %d-frame synthetic pose seq per GPU, "
                                   "generator-only inference, ngf128 n_down3 n_blocks9%s, %s"
                                   % ("configs[1]" if (H, W, args.scales) == (512, 512, 1) else "configs[3]-style", H, W, K,
                                      " + local enhancer (n_scales_spatial 2)" if args.scales == 2 else "",
                                      "flow branch + flow-warp compositor ON" if head_flow else "no flow branch"),
                       "frames_per_gpu": K, "parallelism": "sequence-chunk dp%d" % world,
                       # what the process group itself reports (not what --gpus asked for)
                       "collectives": ("%s, %d ranks" % ("rccl" if backend == "nccl" else backend, dist.get_world_size()))
                                      if dist else "none (single process)",
                       "world_size": dist.get_world_size() if dist else 1,
                       "algorithmic_gflop_per_frame": round(gf, 1),
                       "algorithmic_tflops": round(fps * gf / 1e3, 2),
                       "variants": variants},
            "roofline": roofline, "cpu_baseline": cpu, "e2e": e2e, "box": sampler.summary(),
        }
        if args.hires_frames > 0 and world == 1 and default_geometry:
            result["hires"] = hires_block(dev, args.hires_frames, 4, args.kernel_iters, sampler)
    if args.train_steps > 0 and default_geometry:
        del model, other
        torch.cuda.empty_cache()
        # (RCCL prints its version banner through C stdio when a group is created: fd 1 points at stderr meanwhile, so that
        # stdout carries the one JSON line only)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            train = train_block(dev, dist, world, rank, backend, args.train_steps, args.kernel_iters, args.train_ngf, sampler)
        finally:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if rank == 0:
            result["train_step"] = train
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:   # RCCL writes banner lines through C stdio: flush them so the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result, separators=(",", ":")), flush=True)     # compact: the driver keeps a bounded tail of stdout


if __name__ == "__main__":
    from text2video_amd.distributed import fail_loudly
    fail_loudly(main)      # a rank's exception (a collective's timeout included): one line naming the rank, exit status 3
