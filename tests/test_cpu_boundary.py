"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/t2v.h declares, its
pure-host planning entry points behave, and errors surface as status codes + messages."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(lib_built):
    from text2video_amd import _lib
    header = open(os.path.join(ROOT, "include", "t2v.h")).read()
    declared = set(re.findall(r"\b(t2v_[a-z0-9_]+)\s*\(", header))
    declared -= {"t2v_status"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert getattr(lib_built, name) is not None
    assert lib_built.t2v_abi_version() == _lib.ABI_VERSION


def test_conv_planning_entry_points(lib_built):
    from text2video_amd import _lib
    d = _lib.ConvDesc(64, 64, 1024, 1024, 3, 3, 1, 1, _lib.PAD_REFLECT, 0, 0, 1.0)
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib_built.t2v_conv_out_dims(ctypes.byref(d), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (h.value, w.value) == (64, 64)
    assert lib_built.t2v_conv_packed_weight_floats(ctypes.byref(d), 1024) == 1024 * 9216
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(d)) == 32 * 1024 * 2          # 32 M-tiles of 128 pixels
    dt = _lib.ConvDesc(64, 64, 1024, 512, 3, 3, 2, 1, _lib.PAD_ZERO, 1, 0, 1.0, 1)       # transposed, output_padding 1
    assert lib_built.t2v_conv_out_dims(ctypes.byref(dt), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (h.value, w.value) == (128, 128)
    assert lib_built.t2v_conv_packed_weight_floats(ctypes.byref(dt), 1024) == 512 * 9 * 1024  # 4 phases, 9 taps
    ds = _lib.ConvDesc(512, 512, 9, 128, 7, 7, 1, 3, _lib.PAD_REFLECT, 0, 0, 1.0)        # stem, Cin 9 -> storage 12
    dd = _lib.ConvDesc(257, 257, 128, 64, 4, 4, 2, 2, _lib.PAD_ZERO, 1, 0, 1.0, 0)       # dgrad of a D conv k4 s2 p2
    assert lib_built.t2v_conv_out_dims(ctypes.byref(dd), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (h.value, w.value) == (512, 512)
    dd.H = dd.W = 129; dd.output_padding = 1                                             # odd output 257
    assert lib_built.t2v_conv_out_dims(ctypes.byref(dd), ctypes.byref(h), ctypes.byref(w)) == 0
    assert (h.value, w.value) == (257, 257)
    assert lib_built.t2v_conv_packed_weight_floats(ctypes.byref(ds), 12) == 128 * 608    # K=588 padded to 608


def test_errors_are_status_codes_with_messages(lib_built):
    from text2video_amd import _lib
    bad = _lib.ConvDesc(64, 64, 1024, 1024, 3, 3, 1, 1, _lib.PAD_REFLECT, 0, 0, 1.0)
    h, w = ctypes.c_int(), ctypes.c_int()
    bad.H = 0
    assert lib_built.t2v_conv_out_dims(ctypes.byref(bad), ctypes.byref(h), ctypes.byref(w)) == -1
    assert b"bad dims" in lib_built.t2v_last_error()
    refl = _lib.ConvDesc(2, 2, 8, 8, 7, 7, 1, 3, _lib.PAD_REFLECT, 0, 0, 1.0)
    assert lib_built.t2v_conv_out_dims(ctypes.byref(refl), ctypes.byref(h), ctypes.byref(w)) == -1
    assert b"reflection pad" in lib_built.t2v_last_error()
    tr = _lib.ConvDesc(8, 8, 8, 8, 5, 5, 2, 1, _lib.PAD_ZERO, 1, 0, 1.0, 1)
    assert lib_built.t2v_conv_out_dims(ctypes.byref(tr), ctypes.byref(h), ctypes.byref(w)) == -1
    with pytest.raises(RuntimeError, match="status -1"):
        _lib.check(-1, "unit")


def test_generator_layer_list_matches_host_mirror(lib_built):
    from text2video_amd import _lib
    from text2video_amd.generator import GeneratorSpec, _gen_desc, layer_keys, layer_shapes
    for spec, nl in [(GeneratorSpec(), 2 * (1 + 3 + 10) + 8 + 3 + 1 + 8 + 3 + 1),
                     (GeneratorSpec(no_flow=True), 2 * (1 + 3 + 10) + 8 + 3 + 1),
                     (GeneratorSpec(ngf=64, n_blocks=3, is_local=True, scale=1), 2 * 2 + 6 + 1 + 1 + 6 + 1 + 1)]:
        gd = _gen_desc(spec, 512, 512)
        assert lib_built.t2v_generator_num_layers(ctypes.byref(gd)) == nl == len(layer_keys(spec))
        assert len(layer_shapes(spec)) > 0
    # 283.0 M parameters without flow / 364.7 M with flow
    n = sum(int.__mul__(1, __import__("math").prod(s)) for _, s, _ in layer_shapes(GeneratorSpec(no_flow=True)))
    assert abs(n / 1e6 - 283.0) < 0.5
    n = sum(__import__("math").prod(s) for _, s, _ in layer_shapes(GeneratorSpec()))
    assert abs(n / 1e6 - 364.7) < 0.5
    gd = _gen_desc(GeneratorSpec(no_flow=True), 512, 512)
    ws = lib_built.t2v_generator_workspace_bytes(ctypes.byref(gd))
    assert 0.5e9 < ws < 2e9
    gd = _gen_desc(GeneratorSpec(), 500, 512)  # H not a multiple of 8
    assert lib_built.t2v_generator_workspace_bytes(ctypes.byref(gd)) == 0
    assert b"multiples of 8" in lib_built.t2v_last_error()


def test_product_path_fails_loudly_without_gpu(lib_built):
    """No CPU fallback: asking for a context on a box without a HIP device raises."""
    import torch
    from text2video_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.context()


def test_bench_flop_model_matches_baseline():
    import bench
    assert abs(bench.gflop_per_frame(512, 512, False) - 2572) < 1.0
    assert abs(bench.gflop_per_frame(512, 512, True) - 3316) < 1.0
    assert abs(bench.gflop_per_frame(1024, 1024, False) - 10287) < 2.0


def test_algorithm_choice_and_winograd_bookkeeping(lib_built):
    """Host-side planning needs no GPU: which algorithm a 3x3 layer gets, and the buffer sizes that follow."""
    from text2video_amd import _lib
    from text2video_amd._lib import ConvDesc

    def desc(H, W, C=1024, k=3, stride=1, pad=1, pad_mode=_lib.PAD_REFLECT, algo=0):
        return ConvDesc(H, W, C, C, k, k, stride, pad, pad_mode, 0, _lib.ACT_NONE, 1.0, 0, algo)

    best = lambda d, cap=0: lib_built.t2v_conv_best_algo(ctypes.byref(d), d.Cin, cap)
    assert best(desc(64, 64)) == _lib.ALGO_WINOGRAD_F4                       # 512x512 frames: 256 tiles of 4x4
    assert best(desc(64, 40)) == _lib.ALGO_WINOGRAD_F4                       # 512x320: 160 tiles, padded to 192
    assert best(desc(64, 85)) == _lib.ALGO_WINOGRAD_F4                       # 512x680: ragged 16 x 22 grid
    assert best(desc(16, 16)) == _lib.ALGO_WINOGRAD                          # 64 tiles of 2x2 pad to 128 < 9 rows/pixel
    assert best(desc(8, 8)) == _lib.ALGO_DIRECT                              # padding to 128 tiles never pays
    assert best(desc(64, 64), 1) == _lib.ALGO_DIRECT and best(desc(64, 64), 2) == _lib.ALGO_WINOGRAD
    assert best(desc(64, 64, stride=2, pad_mode=_lib.PAD_ZERO)) == _lib.ALGO_DIRECT
    assert best(desc(64, 64, pad=2, pad_mode=_lib.PAD_ZERO)) == _lib.ALGO_WINOGRAD_F4      # the data gradient's geometry
    assert best(desc(64, 64, C=24)) == _lib.ALGO_DIRECT                      # channels must be a multiple of 32
    f4 = desc(64, 64, algo=_lib.ALGO_WINOGRAD_F4)
    assert lib_built.t2v_conv_winograd_supported(ctypes.byref(f4), 1024) == 3
    assert lib_built.t2v_conv_packed_weight_floats(ctypes.byref(f4), 1024) == 36 * 1024 * 1024
    # V + M of the 256 tiles, then the hand-over scratch of the fixed-grid GEMM (1024 blocks x 4 waves x (64 x 64 + 2))
    assert lib_built.t2v_conv_winograd_workspace_floats(ctypes.byref(f4), 1024) == 36 * 256 * 2048 + 1024 * 4 * (64 * 64 + 2)
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(f4)) == 32 * 1024 * 2      # one partial per 128 output pixels
    r = desc(64, 40, algo=_lib.ALGO_WINOGRAD_F4)
    assert lib_built.t2v_conv_winograd_workspace_floats(ctypes.byref(r), 1024) == 36 * 192 * 2048 + 1024 * 4 * (64 * 64 + 2)   # 160 tiles -> 192
    assert lib_built.t2v_conv_backward_weight_winograd_supported(ctypes.byref(f4), 1024, 1024) == 1
    assert lib_built.t2v_conv_backward_weight_winograd_workspace_floats(ctypes.byref(f4), 1024, 2) == \
        36 * 2 * 256 * 2048 + 36 * 1024 * 1024 + 1024 * 4 * (64 * 64 + 2)     # V + M_dy, dU, the fixed grid's hand-over area
    assert lib_built.t2v_conv_winograd_tile_rows(ctypes.byref(f4)) == 256 and lib_built.t2v_conv_winograd_tile_rows(ctypes.byref(r)) == 192
    # activations of 2 GiB and more do not fit the kernels' 32-bit buffer offsets: the plan refuses them (no wrap)
    big = desc(2048, 2048, C=128)
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(big)) == 0 and b"too large" in lib_built.t2v_last_error()
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(desc(1024, 1024, C=128))) > 0


def test_polyphase_bookkeeping_and_generator_selection(lib_built):
    """T2V_ALGO_POLYPHASE (ABI 15; the deep stride-2 3x3 convs and their transposed counterparts as polyphase Winograd F(4,2)):
    where it applies, its buffer sizes, and which layers of the generator the library gives it to -- host-side planning only."""
    from text2video_amd import _lib
    from text2video_amd._lib import ConvDesc

    def desc(H, W, Cin, Cout, tr=0, stride=2, pad=1, pad_mode=_lib.PAD_ZERO, k=3, act=_lib.ACT_NONE):
        return ConvDesc(H, W, Cin, Cout, k, k, stride, pad, pad_mode, tr, act, 1.0, 1 if tr else 0, _lib.ALGO_POLYPHASE)

    ok = lambda d: lib_built.t2v_conv_polyphase_supported(ctypes.byref(d), d.Cin) & 1
    pays = lambda d: lib_built.t2v_conv_polyphase_supported(ctypes.byref(d), d.Cin) >> 1
    down3, up1 = desc(128, 128, 512, 1024), desc(64, 64, 1024, 512, tr=1)
    assert ok(down3) == 1 and ok(up1) == 1 and ok(desc(128, 80, 512, 1024)) == 1 and ok(desc(64, 40, 1024, 512, tr=1)) == 1
    assert ok(desc(128, 170, 512, 1024)) == 1 and ok(desc(64, 85, 1024, 512, tr=1)) == 1       # ragged tile grids (512x680 frames)
    assert ok(desc(128, 85, 512, 1024)) == 0          # a down conv needs even H, W
    assert ok(desc(128, 128, 512, 1024, stride=1)) == 0 and ok(desc(128, 128, 512, 1024, pad_mode=_lib.PAD_REFLECT)) == 0
    assert ok(desc(128, 128, 48, 1024)) == 0 and ok(desc(128, 128, 512, 192)) == 0      # Cin % 32, Cout % 128
    assert ok(desc(128, 128, 512, 1024, act=_lib.ACT_LRELU)) == 0 and ok(desc(128, 128, 512, 1024, k=4)) == 0
    assert lib_built.t2v_conv_polyphase_supported(ctypes.byref(down3), 516) == 0        # channel storage must equal Cin
    assert pays(down3) == 1 and pays(up1) == 1 and pays(desc(512, 512, 128, 256)) == 0 and pays(desc(16, 16, 512, 1024)) == 0
    # 81 positions: packed weight, V + M of the 256 tiles + the fixed-grid GEMM's hand-over scratch, one partial per 128 pixels
    assert lib_built.t2v_conv_packed_weight_floats(ctypes.byref(down3), 512) == 81 * 1024 * 512
    assert lib_built.t2v_conv_winograd_workspace_floats(ctypes.byref(down3), 512) == 81 * 256 * (512 + 1024) + 1024 * 4 * (64 * 64 + 2)
    assert lib_built.t2v_conv_winograd_workspace_floats(ctypes.byref(up1), 1024) == 81 * 256 * (1024 + 512) + 1024 * 4 * (64 * 64 + 2)
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(down3)) == (64 * 64 // 128) * 1024 * 2
    assert lib_built.t2v_conv_stats_floats(ctypes.byref(up1)) == (128 * 128 // 128) * 512 * 2
    h, w = ctypes.c_int(), ctypes.c_int()
    assert lib_built.t2v_conv_out_dims(ctypes.byref(up1), ctypes.byref(h), ctypes.byref(w)) == 0 and (h.value, w.value) == (128, 128)
    # the generator: at ngf 128 the 256->512 / 512->1024 downs and the 1024->512 / 512->256 ups, not the 128<->256 layers
    from text2video_amd.generator import GeneratorSpec, _gen_desc
    for (H, W, cap, want) in ((512, 512, 0, 8), (512, 320, 0, 8), (512, 680, 0, 8), (512, 512, 1, 0)):
        gd = _gen_desc(GeneratorSpec(ngf=128, n_downsample=3, n_blocks=9, no_flow=False, norm="batch"), H, W, cap)
        n = lib_built.t2v_generator_num_layers(ctypes.byref(gd))
        algos = []
        for i in range(n):
            cd, xcs = ConvDesc(), ctypes.c_int()
            assert lib_built.t2v_generator_layer_desc(ctypes.byref(gd), i, ctypes.byref(cd), ctypes.byref(xcs)) == 0
            if cd.algo == _lib.ALGO_POLYPHASE:
                assert cd.stride == 2 and min(cd.Cin, cd.Cout) >= 256
            algos.append(cd.algo)
        assert algos.count(_lib.ALGO_POLYPHASE) == want, (H, W, cap, algos)


def test_makefile_rebuilds_objects_when_the_winograd_constants_change():
    """csrc/Makefile: every object depends on winograd_f4_consts.h (generated by scripts/gen_winograd_consts.py); an
    incremental `make` -- what __graft_entry__.build() runs -- must rebuild after the header is regenerated."""
    import os
    import re
    mk = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "text2video_amd", "csrc", "Makefile")).read()
    rule = re.search(r"^%\.o: %\.hip (.*)$", mk, re.M).group(1).split()
    assert "winograd_f4_consts.h" in rule and "t2v_internal.h" in rule and "../../include/t2v.h" in rule
    assert "polyphase_consts.h" in rule
