"""GPU parity tests, whole generator: HIP path (through the C ABI) vs the CPU oracle on the same
seeded weights and pose inputs.  north_star tolerance: per-pixel |delta| <= 1e-3 (fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _pose_seq(n, H, W, seed=0):
    """synthetic pose maps: -1 background, ~2 % pixels uniformly in [-1,1]."""
    rng = np.random.default_rng(seed)
    a = -np.ones((n, 3, H, W), np.float32)
    m = rng.random((n, 1, H, W)) < 0.02
    a = np.where(m, rng.uniform(-1, 1, size=(n, 3, H, W)).astype(np.float32), a)
    return torch.from_numpy(a)


def _build(spec_kw, scales=1, seed=1, init="vid2vid", flow_gain=0.1):
    from oracle.generator_ref import CompositeGenerator, CompositeLocalGenerator, Vid2VidInferenceRef
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
    ref_nets, hip_nets = [], []
    for s in range(scales):
        if s == 0:
            spec = GeneratorSpec(**spec_kw)
            net = CompositeGenerator(spec.input_nc, 3, spec.prev_nc, spec.ngf, spec.n_downsample, spec.n_blocks,
                                     spec.no_flow, spec.norm)
        else:
            kw = dict(spec_kw)
            kw.update(ngf=spec_kw["ngf"] // (2 ** s), n_blocks=2, is_local=True, scale=s)
            spec = GeneratorSpec(**kw)
            net = CompositeLocalGenerator(spec.input_nc, 3, spec.prev_nc, spec_kw["ngf"], 2, s, spec.no_flow, spec.norm)
        sd = synthetic_state_dict(spec, seed + s, init, flow_gain)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and all(("running" in k or "num_batches" in k) for k in missing), (missing, unexpected)
        ref_nets.append(net)
        hip_nets.append(HipGenerator(spec, "cuda:0").load_state_dict(sd))
    return Vid2VidInferenceRef(ref_nets), Vid2VidModelG(hip_nets)


CASES = [
    ("flow_batchnorm", dict(ngf=32, n_downsample=3, n_blocks=3, no_flow=False, norm="batch"), 1, 64, 64),
    ("noflow_instancenorm", dict(ngf=32, n_downsample=3, n_blocks=4, no_flow=True, norm="instance"), 1, 64, 96),
    ("flow_ragged_680like", dict(ngf=32, n_downsample=3, n_blocks=2, no_flow=False, norm="batch"), 1, 64, 88),
    ("two_scale_flow", dict(ngf=64, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 2, 64, 64),
    ("two_scale_noflow", dict(ngf=64, n_downsample=2, n_blocks=2, no_flow=True, norm="instance"), 2, 64, 96),
    # bottleneck 32x16 / 16x32 (128 Winograd tiles): the ResnetBlock convs take the Winograd F(2x2,3x3) path
    ("winograd_noflow", dict(ngf=16, n_downsample=2, n_blocks=3, no_flow=True, norm="instance"), 1, 128, 64),
    ("winograd_flow", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 64, 128),
    # bottleneck 64x32 / 32x64 (128 tiles of 4x4): Winograd F(4x4,3x3)
    ("winograd_f4_noflow", dict(ngf=16, n_downsample=2, n_blocks=3, no_flow=True, norm="instance"), 1, 256, 128),
    ("winograd_f4_flow", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 128, 256),
    # odd bottleneck 32x85 (ragged 4x4 tile grid, zero-tile padding): the 512x680 fadg0 geometry scaled down
    ("winograd_ragged_680like", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 128, 340),
]


def test_winograd_is_selected_and_matches_direct():
    """t2v_generator_layer_desc reports Winograd for the ResnetBlock convs where the geometry allows, the
    direct kernel elsewhere / when conv_algo=1; both whole-frame paths agree to fp32 rounding."""
    import ctypes
    from text2video_amd import _lib
    from text2video_amd.generator import GeneratorSpec, HipGenerator, _gen_desc, layer_keys, synthetic_state_dict
    lib = _lib.load()
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=3, no_flow=True, norm="instance")
    keys = layer_keys(spec)

    def algos(H, W, conv_algo):
        gd = _gen_desc(spec, H, W, conv_algo)
        out = []
        for i in range(len(keys)):
            cd, xcs = _lib.ConvDesc(), ctypes.c_int()
            assert lib.t2v_generator_layer_desc(ctypes.byref(gd), i, ctypes.byref(cd), ctypes.byref(xcs)) == 0
            out.append(cd.algo)
        return out

    a = algos(128, 64, 0)
    rb = [i for i, (ck, nk, kind) in enumerate(keys) if ".conv_block" in str(ck)]
    assert len(rb) >= 6 and [i for i, v in enumerate(a) if v == _lib.ALGO_WINOGRAD] == rb, (a, rb)
    assert sum(algos(128, 64, 1)) == 0 and sum(algos(16, 16, 0)) == 0     # 4x4 bottleneck: padding to 128 tiles never pays
    a85 = algos(64, 4 * 85, 0)       # odd 16x85 bottleneck (the 512x680 geometry in small): ragged F(4x4) grid
    assert [i for i, v in enumerate(a85) if v == _lib.ALGO_WINOGRAD_F4] == rb
    a4 = algos(256, 128, 0)     # enough 4x4 tiles: F(4x4,3x3); conv_algo=2 caps it at F(2x2,3x3)
    assert [i for i, v in enumerate(a4) if v == _lib.ALGO_WINOGRAD_F4] == rb and sum(a4) == 2 * len(rb)
    assert [i for i, v in enumerate(algos(256, 128, 2)) if v == _lib.ALGO_WINOGRAD] == rb

    sd = synthetic_state_dict(spec, 3)
    pose = _pose_seq(3, 128, 64, seed=2)
    x = torch.zeros(128, 64, 12, device="cuda:0")
    x[..., :9] = pose.reshape(9, 128, 64).permute(1, 2, 0).to("cuda:0")
    prev = torch.rand(128, 64, 8, device="cuda:0") * 2 - 1
    prev[..., 6:] = 0
    y_w = HipGenerator(spec, "cuda:0", conv_algo=0).load_state_dict(sd).forward(x, prev)["out"]
    y_d = HipGenerator(spec, "cuda:0", conv_algo=1).load_state_dict(sd).forward(x, prev)["out"]
    err = (y_w - y_d).abs().max().item()
    assert 0 < err <= TOL_FORCED, err
    x4 = torch.cat([x, x.flip(0)], 0).repeat(1, 2, 1).contiguous()      # 256 x 128
    p4 = torch.cat([prev, prev.flip(1)], 0).repeat(1, 2, 1).contiguous()
    ys = [HipGenerator(spec, "cuda:0", conv_algo=a).load_state_dict(sd).forward(x4, p4)["out"] for a in (0, 2, 1)]
    e4, e2 = (ys[0] - ys[2]).abs().max().item(), (ys[1] - ys[2]).abs().max().item()
    assert 0 < e2 <= TOL_FORCED and 0 < e4 <= TOL_FORCED, (e4, e2)


TOL_FORCED = 2e-4  # same inputs in, fp32 summation-order differences only (observed ~1e-5)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sequence_teacher_forced(case):
    """Every frame of a sequence with the oracle's own previous frames as the recurrent input:
    isolates the kernels from the (chaotic, for random-init weights) frame recurrence."""
    name, kw, scales, H, W = case
    for flow_gain in ((0.1, 1.0) if not kw["no_flow"] else (1.0,)):
        ref, hip = _build(kw, scales, flow_gain=flow_gain)
        poses = _pose_seq(6, H, W, seed=5)
        for t in range(2, 6):
            A = poses[t - 2:t + 1].unsqueeze(0)
            first = ref.fake_B_prev is None
            if not first:
                hip.load_prev(ref.fake_B_prev)
            want = ref.inference(A)
            got, _ = hip.inference(A.to("cuda:0"))
            assert got.shape == want.shape
            if first:
                continue  # zero-image first frame: see test_first_frame_zero_prev
            err = (got.cpu() - want).abs().max().item()
            # full-gain random flow heads (+-40 px) multiply the conv rounding differences by the
            # x20 flow multiplier times the image gradient: north_star tolerance there
            tol = TOL_FORCED if (kw["no_flow"] or flow_gain < 1.0) else TOL
            assert err <= tol, "%s gain %g frame %d: max|delta|=%g" % (name, flow_gain, t, err)


@pytest.mark.parametrize("case", [c for c in CASES if c[1]["norm"] == "batch"],
                         ids=[c[0] for c in CASES if c[1]["norm"] == "batch"])
def test_first_frame_zero_prev(case):
    """--no_first_img first frame: zero previous frames, raw output only.  The prev-image encoder
    sees an all-zero image, i.e. norm layers over constant maps: with affine norm (upstream default
    --norm batch) that is well defined and must match to the north_star tolerance.  (With
    InstanceNorm(affine=False) it is 0/0 -- rounding noise renormalised to unit variance -- in the
    reference algorithm itself, so no implementation can be compared there.)  Also checks that a
    reset() (change_seq) reproduces the first frame bit for bit."""
    name, kw, scales, H, W = case
    ref, hip = _build(kw, scales)
    A = _pose_seq(3, H, W, seed=6).unsqueeze(0)
    want = ref.inference(A)
    got, _ = hip.inference(A.to("cuda:0"))
    assert (got.cpu() - want).abs().max().item() <= TOL
    hip.inference(A.to("cuda:0"))
    hip.reset()
    again, _ = hip.inference(A.to("cuda:0"))
    assert torch.equal(again, got)


def test_free_running_sequence_vs_fp64_conditioning():
    """Free-running recurrence (each path feeds on its OWN previous outputs).  A random-init
    generator is an expanding map of its previous frames, so two fp32 evaluations drift apart; the
    yardstick is an fp64 evaluation of the same network: the HIP path must stay as close to fp64
    as the fp32 CPU oracle does (x10 slack + 1e-4), and within 1e-3 while the fp32 oracle itself is
    within 1e-4 of fp64."""
    import copy
    name, kw, scales, H, W = CASES[0]
    ref, hip = _build(kw, scales)
    ref64 = copy.deepcopy(ref)
    for n in ref64.nets:
        n.double()
    poses = _pose_seq(7, H, W, seed=7)
    for t in range(2, 7):
        A = poses[t - 2:t + 1].unsqueeze(0)
        truth = ref64.inference(A.double())
        e32 = (ref.inference(A).double() - truth).abs().max().item()
        ehip = (hip.inference(A.to("cuda:0"))[0].cpu().double() - truth).abs().max().item()
        print("frame %d: |oracle32-fp64|=%.2e |hip-fp64|=%.2e" % (t, e32, ehip))
        assert ehip <= 10 * e32 + 1e-4
        if e32 <= 1e-4:
            assert ehip <= TOL


def test_free_running_64_pose_maps_vs_fp64_conditioning():
    """BASELINE configs[1]'s sequence LENGTH (64 pose maps -> 62 frames) free-running at reduced width: every path feeds
    on its own previous outputs for the whole sequence.  Yardstick as above -- an fp64 evaluation of the same network:
    the HIP path stays as close to fp64 as the fp32 CPU oracle does (x10 slack + 1e-4) on every one of the 62 frames,
    and within 1e-3 wherever the fp32 oracle itself is within 1e-4."""
    import copy
    name, kw, scales, H, W = CASES[0]
    ref, hip = _build(kw, scales)
    ref64 = copy.deepcopy(ref)
    for n in ref64.nets:
        n.double()
    poses = _pose_seq(64, H, W, seed=17)
    worst32 = worsthip = 0.0
    tight = 0
    for t in range(2, 64):
        A = poses[t - 2:t + 1].unsqueeze(0)
        truth = ref64.inference(A.double())
        e32 = (ref.inference(A).double() - truth).abs().max().item()
        ehip = (hip.inference(A.to("cuda:0"))[0].cpu().double() - truth).abs().max().item()
        worst32, worsthip = max(worst32, e32), max(worsthip, ehip)
        assert ehip <= 10 * e32 + 1e-4, "frame %d: |hip-fp64| = %g, |oracle32-fp64| = %g" % (t, ehip, e32)
        if e32 <= 1e-4:
            tight += 1
            assert ehip <= TOL
    print("62 free-running frames: max |oracle32-fp64| = %.2e, max |hip-fp64| = %.2e, %d frames with the fp32 oracle within 1e-4"
          % (worst32, worsthip, tight))
    assert tight >= 1


def test_fullsize_frame_512_noflow_matches_oracle():
    """BASELINE config 2 geometry (ngf 128, 3 down, 9 blocks, 512x512, openpose_only => no flow):
    the first frame (zero prev) and a teacher-forced second frame through the real-size network."""
    kw = dict(ngf=128, n_downsample=3, n_blocks=9, no_flow=True, norm="batch")
    ref, hip = _build(kw, 1, init="uniform_fan_in")
    poses = _pose_seq(4, 512, 512, seed=3)
    for t in (2, 3):
        A = poses[t - 2:t + 1].unsqueeze(0)
        if ref.fake_B_prev is not None:
            hip.load_prev(ref.fake_B_prev)
        want = ref.inference(A)
        got, _ = hip.inference(A.to("cuda:0"))
        err = (got.cpu() - want).abs().max().item()
        print("512x512 frame %d max|delta| = %.3g" % (t, err))
        assert err <= TOL, "frame %d: max|delta|=%g" % (t, err)
        assert want.abs().max().item() > 0.05  # the comparison is not vacuous


def test_fullsize_frame_512_flow_property():
    """Full-size flow generator: size-independent properties instead of a CPU oracle run --
    (1) out == raw*w + warp*(1-w) recomputed from the returned taps, (2) determinism."""
    from text2video_amd import ops
    from text2video_amd.generator import GeneratorSpec, HipGenerator, synthetic_state_dict
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=False, norm="batch")
    net = HipGenerator(spec, "cuda:0").load_state_dict(synthetic_state_dict(spec, 1))
    poses = _pose_seq(3, 512, 512, seed=4)
    pose = ops.nchw_to_nhwc(poses.reshape(9, 512, 512).to("cuda:0"))
    prev = torch.tanh(torch.randn(512, 512, 8, device="cuda:0"))
    prev[..., 6:] = 0
    r1 = net.forward(pose, prev, False, want=("out", "raw", "flow_w"))
    r2 = net.forward(pose, prev, False, want=("out", "raw", "flow_w"))
    assert torch.equal(r1["out"], r2["out"])
    out, warp = ops.flow_warp_composite(r1["raw"], r1["flow_w"], prev, 3, want_warp=True)
    assert torch.equal(out, r1["out"])
    w = r1["flow_w"][..., 2:3]
    assert (w >= 0).all() and (w <= 1).all() and torch.isfinite(r1["out"]).all()
    assert (r1["raw"][..., :3].abs() <= 1).all()


def test_generator_edge_geometries_and_argument_errors():
    """smallest legal frame (bottleneck 2x2), strongly non-square frames, and the C ABI's refusals."""
    import ctypes
    from text2video_amd import _lib
    from text2video_amd.generator import GeneratorSpec, HipGenerator, _gen_desc, synthetic_state_dict
    kw = dict(ngf=32, n_downsample=3, n_blocks=2, no_flow=False, norm="batch")
    for (H, W) in [(16, 16), (16, 72), (72, 16)]:
        ref, hip = _build(kw, 1)
        A = _pose_seq(3, H, W, seed=8).unsqueeze(0)
        want = ref.inference(A)
        got, _ = hip.inference(A.to("cuda:0"))
        assert (got.cpu() - want).abs().max().item() <= TOL, (H, W)
    spec = GeneratorSpec(**kw)
    net = HipGenerator(spec, "cuda:0")
    with pytest.raises(RuntimeError, match="before load_state_dict"):
        net.forward(torch.zeros(16, 16, 12, device="cuda:0"), torch.zeros(16, 16, 8, device="cuda:0"))
    net.load_state_dict(synthetic_state_dict(spec, 1))
    with pytest.raises(RuntimeError, match="multiples of 8"):      # H not divisible by 2^n_downsample
        net.forward(torch.zeros(20, 16, 12, device="cuda:0"), torch.zeros(20, 16, 8, device="cuda:0"))
    lib = _lib.load()
    gd = _gen_desc(spec, 8, 8)                                       # bottleneck 1x1: reflection pad impossible
    assert lib.t2v_generator_workspace_bytes(ctypes.byref(gd)) == 0 and b"too small" in lib.t2v_last_error()
    # too small a workspace is reported, not written past
    gd = _gen_desc(spec, 16, 16)
    net._workspace(16, 16)                                           # packs the weights for this geometry
    io = _lib.GenIO()
    x = torch.zeros(16, 16, 12, device="cuda:0")
    io.pose, io.prev, io.out = x.data_ptr(), x.data_ptr(), x.data_ptr()
    ws = torch.empty(256, dtype=torch.uint8, device="cuda:0")
    st = lib.t2v_generator_forward(net.ctx.handle, None, ctypes.byref(gd), net._layers, len(net.keys), ctypes.byref(io),
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel())
    assert st == -3 and b"workspace" in lib.t2v_last_error()


@pytest.mark.parametrize("no_flow", [True, False], ids=["noflow", "flow"])
def test_two_stream_frames_are_reproducible(no_flow):
    """Race screen for the two-stream orchestration (encoders / image+flow branches run concurrently on the
    caller's stream and the context's side stream): the same sequence twice, on the same buffers, gives the same bits,
    for the full-width bottleneck (Winograd path) as well."""
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
    spec = GeneratorSpec(ngf=32, n_downsample=3, n_blocks=4, no_flow=no_flow, norm="batch")
    sd = synthetic_state_dict(spec, 9, flow_gain=0.1)
    H, W, n = 256, 256, 8
    seq = _pose_seq(n + 2, H, W, seed=4)
    wins = []
    for t in range(n):
        w = torch.zeros(H, W, 12, device="cuda:0")
        w[..., :9] = seq[t:t + 3].reshape(9, H, W).permute(1, 2, 0).to("cuda:0")
        wins.append(w)
    model = Vid2VidModelG([HipGenerator(spec, "cuda:0").load_state_dict(sd)])
    runs = []
    for _ in range(3):
        model.reset()
        runs.append([model.inference_nhwc(w).clone() for w in wins])
        torch.cuda.synchronize()
    for t in range(n):
        assert torch.equal(runs[0][t], runs[1][t]) and torch.equal(runs[0][t], runs[2][t]), t
        assert torch.isfinite(runs[0][t]).all()



def test_frames_do_not_depend_on_the_overlap_hint_or_the_stream_count(t2v_env):
    """Full-width generator (ngf 128, 9 blocks, flow branch), 512x512: (i) the two-stream frame -- and the one whose ResnetBlock GEMM stage
    runs on 256x128 tiles with one block per CU under the overlap hint (T2V_OVERLAP_HINT_SINGLE=1) -- is BIT-identical to the frame with T2V_OVERLAP_HINT=0
    (128x128 tiles, two per CU) and to the single-stream frame (which kernel owns a tile changes, an output's K-ordered MFMA
    chain does not)."""
    from text2video_amd import ops
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=False, norm="batch")
    sd = synthetic_state_dict(spec, 1, flow_gain=0.1)
    H = W = 512
    seq = _pose_seq(5, H, W, seed=11)
    wins = []
    for t in range(3):
        w = torch.zeros(H, W, 12, device="cuda:0")
        w[..., :9] = seq[t:t + 3].reshape(9, H, W).permute(1, 2, 0).to("cuda:0")
        wins.append(w)
    model = Vid2VidModelG([HipGenerator(spec, "cuda:0").load_state_dict(sd)])
    desc = ops.conv_desc(64, 64, 1024, 1024, 3, 1, 1, ops.PAD_REFLECT, algo=ops.ALGO_WINOGRAD_F4)

    def run():
        model.reset()
        out = [model.inference_nhwc(w).clone() for w in wins]
        torch.cuda.synchronize()
        return out
    base = run()                                     # two streams, hint on
    t2v_env("T2V_OVERLAP_HINT_SINGLE", "1")          # the rule of rounds 4-5: one image's GEMM stage on the tall form too
    prev = ops.set_overlap_hint(True)
    try:
        assert "256x128" in ops.winograd_gemm_form(desc)      # (what the generator's own scope then selects)
    finally:
        ops.set_overlap_hint(prev)
    tall = run()
    t2v_env("T2V_OVERLAP_HINT_SINGLE", "0")
    t2v_env("T2V_OVERLAP_HINT", "0")
    no_hint = run()
    t2v_env("T2V_OVERLAP_HINT", "1")
    t2v_env("T2V_STREAMS", "1")
    one_stream = run()
    t2v_env("T2V_STREAMS", "0")
    for t in range(3):
        assert torch.equal(base[t], no_hint[t]) and torch.equal(base[t], one_stream[t]) and torch.equal(base[t], tall[t]), t
        assert torch.isfinite(base[t]).all() and base[t][..., :3].abs().max().item() > 0.05


def test_lazy_resblock_chain_is_bit_identical_to_the_apply_form():
    """The ResnetBlock chains run with every norm applied inside the next conv's Winograd input transform (default);
    T2V_CHAIN_LAZY=0 keeps the separate apply passes.  Same arithmetic in the same order => the same bits, for square and
    ragged tile grids, instance norm and batch norm (affine), with and without the flow branch."""
    import os
    import subprocess
    import sys
    import tempfile
    code = ("import sys, torch, numpy as np; sys.path.insert(0, %r);"
            "from text2video_amd.generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict;"
            "outs = [];\n"
            "g = torch.Generator().manual_seed(0)\n"
            "for norm, no_flow, nb in (('batch', False, 4), ('instance', True, 9), ('batch', False, 2)):\n"
            "    spec = GeneratorSpec(ngf=32, n_downsample=3, n_blocks=nb, no_flow=no_flow, norm=norm)\n"
            "    m = Vid2VidModelG([HipGenerator(spec, 'cuda:0').load_state_dict(synthetic_state_dict(spec, 9, flow_gain=0.1))])\n"
            "    for H, W in ((256, 256), (128, 344), (72, 40)):\n"
            "        m.reset()\n"
            "        for t in range(3):\n"
            "            w = torch.zeros(H, W, 12, device='cuda:0'); w[..., :9] = (torch.rand(H, W, 9, generator=g) * 2 - 1).cuda()\n"
            "            outs.append(m.inference_nhwc(w).cpu().numpy())\n"
            "np.save(sys.argv[1], np.concatenate([o.reshape(-1) for o in outs]))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("0", "1"):
            out = os.path.join(d, "o%s.npy" % mode)
            r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, T2V_CHAIN_LAZY=mode), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[mode] = np.load(out)
    assert np.isfinite(res["1"]).all() and np.abs(res["1"]).max() > 0.05
    assert np.array_equal(res["0"], res["1"])


def test_xcd_sliced_transforms_are_bit_identical_to_the_flat_thread_index(t2v_env):
    """With 1024 channels the Winograd input transforms (plain / norm + ReLU / norm + residual) and the data gradient's
    output transform hand every XCD its own 64-pair channel slices (T2V_XCD_SLICES=1, default: the tiles that share input
    pixels read them through one L2); T2V_XCD_SLICES=0 keeps the flat (tile, channel) thread index.  Only which thread
    computes an item changes: frames (two sequences in lock-step, square and ragged tile grids) and data gradients carry
    the same bits."""
    from text2video_amd import ops
    from text2video_amd.generator import GeneratorSpec, HipGenerator, Recurrence, Vid2VidModelG, synthetic_state_dict
    spec = GeneratorSpec(ngf=128, n_downsample=3, n_blocks=1, no_flow=False, norm="batch")
    net = Vid2VidModelG([HipGenerator(spec, "cuda:0").load_state_dict(synthetic_state_dict(spec, 9, flow_gain=0.1))])
    g = torch.Generator().manual_seed(0)
    wins = {}
    for H, W in ((256, 256), (264, 136)):      # 32 x 32 and 33 x 17 bottlenecks: 64 whole / 45 ragged Winograd tiles
        assert ops.best_conv_algo(ops.conv_desc(H // 8, W // 8, 1024, 1024, 3, 1, 1, ops.PAD_REFLECT)) == ops.ALGO_WINOGRAD_F4
        wins[(H, W)] = []
        for t in range(2):
            pair = []
            for q in range(2):
                w = torch.zeros(H, W, 12, device="cuda:0")
                w[..., :9] = (torch.rand(H, W, 9, generator=g) * 2 - 1).cuda()
                pair.append(w)
            wins[(H, W)].append(pair)
    C = 1024
    grads = []
    for H, W in ((16, 16), (8, 24)):
        desc = ops.with_algo(ops.conv_desc(H, W, C, C, 3, 1, 1, ops.PAD_REFLECT), ops.ALGO_WINOGRAD_F4)
        xs, dys = torch.randn(2, H, W, C, generator=g).cuda(), torch.randn(2, H, W, C, generator=g).cuda()
        wd = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
        grads.append((desc, xs, dys, ops.pack_conv_weight_transposed(wd, desc, C)))

    def run():
        outs = []
        for hw, seq in wins.items():
            recs = [Recurrence(), Recurrence()]
            for pair in seq:
                outs += [o.clone() for o in net.inference_nhwc_batch(pair, recs)]
        for desc, xs, dys, ut in grads:
            ws = ops.backward_weight_winograd_workspace(desc, C, 2, "cuda:0")
            ops.conv2d_backward_weight_winograd_stages(xs, dys, desc, ws, 2, 0, False)
            outs += [ops.conv2d_backward_data_winograd(desc, 2, b, ws, C, ut).clone() for b in range(2)]
        torch.cuda.synchronize()
        return outs
    sliced = run()
    t2v_env("T2V_XCD_SLICES", "0")
    flat = run()
    assert len(sliced) == len(flat) == 12
    for a, b in zip(sliced, flat):
        assert torch.isfinite(a).all() and a.abs().max().item() > 0.05
        assert torch.equal(a, b)


def test_alternating_frame_geometries_keep_their_packed_weights():
    """Two lanes whose sequences have different frame sizes alternate every step of the lock-step loop: the
    generator keeps the packed / Winograd-transformed weights and the arena of the last geometries, so only the FIRST frame
    of each geometry packs; the frames are those of a generator that only ever saw that geometry."""
    from text2video_amd import ops
    from text2video_amd.generator import GeneratorSpec, HipGenerator, synthetic_state_dict
    spec = GeneratorSpec(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    sd = synthetic_state_dict(spec, 3, "vid2vid", 0.1)
    geoms = [(128, 256), (128, 128)]

    def inputs(H, W, seed):
        pose = ops.nchw_to_nhwc(_pose_seq(3, H, W, seed).reshape(9, H, W).cuda())
        prev = torch.zeros(H, W, 8, device="cuda:0")
        prev[..., :6] = torch.tanh(torch.randn(H, W, 6, generator=torch.Generator().manual_seed(seed))).cuda()
        return pose, prev
    want = {}
    for g in geoms:
        alone = HipGenerator(spec, "cuda:0").load_state_dict(sd)
        want[g] = [alone.forward(*inputs(*g, seed=t), False)["out"].clone() for t in range(3)]
    hip = HipGenerator(spec, "cuda:0").load_state_dict(sd)
    packs = []
    orig = hip._pack
    hip._pack = lambda gd: (packs.append((gd.H, gd.W)), orig(gd))[1]
    for t in range(3):
        for g in geoms:
            assert torch.equal(hip.forward(*inputs(*g, seed=t), False)["out"], want[g][t]), (g, t)
    assert packs == geoms, packs
    # a third geometry evicts the oldest; coming back to it packs again, the other one is still there
    hip.forward(*inputs(64, 64, 5), False)
    hip.forward(*inputs(128, 128, 1), False)
    assert packs == geoms + [(64, 64)], packs
    hip.forward(*inputs(128, 256, 1), False)
    assert packs == geoms + [(64, 64), (128, 256)], packs
