"""ctypes binding of libt2v_hip.so (C ABI declared in include/t2v.h).

The product path has NO fallback: if the shared library is missing or does not export the
expected symbols, importing/using it raises.  Build it with `python __graft_entry__.py` or
`make -C text2video_amd/csrc`.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (T2V_LIBRARY: another build of the same ABI -- same-box A/B runs of a kernel change, scripts/ only)
LIB_PATH = os.environ.get("T2V_LIBRARY") or os.path.join(_HERE, "lib", "libt2v_hip.so")

T2V_OK = 0
PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_TANH, ACT_FLOW_W, ACT_LRELU = 0, 1, 2, 3
ABI_VERSION = 18
MAX_BATCH = 8     # T2V_MAX_BATCH
ALGO_DIRECT, ALGO_WINOGRAD, ALGO_WINOGRAD_F4, ALGO_POLYPHASE = 0, 1, 2, 3


class ConvDesc(Structure):
    """t2v_conv_desc (include/t2v.h)."""
    _fields_ = [("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int), ("kH", c_int), ("kW", c_int),
                ("stride", c_int), ("pad", c_int), ("pad_mode", c_int), ("transposed", c_int), ("act", c_int),
                ("act_scale", c_float), ("output_padding", c_int), ("algo", c_int)]


class GenDesc(Structure):
    """t2v_gen_desc (include/t2v.h)."""
    _fields_ = [("H", c_int), ("W", c_int), ("input_nc", c_int), ("prev_nc", c_int), ("output_nc", c_int),
                ("ngf", c_int), ("n_downsample", c_int), ("n_blocks", c_int), ("no_flow", c_int),
                ("norm_affine", c_int), ("is_local", c_int), ("flow_multiplier", c_float), ("eps", c_float),
                ("conv_algo", c_int)]


class Layer(Structure):
    """t2v_layer."""
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("gamma", c_void_p), ("beta", c_void_p)]


class GenIO(Structure):
    """t2v_gen_io."""
    _fields_ = [("pose", c_void_p), ("prev", c_void_p), ("coarse_img_feat", c_void_p),
                ("coarse_flow_feat", c_void_p), ("use_raw_only", c_int), ("out", c_void_p), ("raw", c_void_p),
                ("flow_w", c_void_p), ("img_feat", c_void_p), ("flow_feat", c_void_p)]


# name -> (restype, argtypes); every symbol include/t2v.h declares
SIGNATURES = {
    "t2v_abi_version": (c_int, []),
    "t2v_last_error": (c_char_p, []),
    "t2v_create": (c_int, [POINTER(c_void_p), c_int]),
    "t2v_destroy": (c_int, [c_void_p]),
    "t2v_reload_env": (None, []),
    "t2v_set_overlap_hint": (c_int, [c_int]),
    "t2v_check_async_errors": (c_int, []),
    "t2v_fixed_grid_enabled": (c_int, []),
    "t2v_debug_async_error": (None, [c_int]),
    "t2v_conv_out_dims": (c_int, [POINTER(ConvDesc), POINTER(c_int), POINTER(c_int)]),
    "t2v_conv_packed_weight_floats": (c_size_t, [POINTER(ConvDesc), c_int]),
    "t2v_conv_pack_weight": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "t2v_conv_pack_weight_adjoint": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "t2v_conv_stats_floats": (c_size_t, [POINTER(ConvDesc)]),
    "t2v_conv2d_forward": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p]),
    "t2v_conv2d_forward_batch": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_int, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p]),
    "t2v_conv_winograd_supported": (c_int, [POINTER(ConvDesc), c_int]),
    "t2v_conv_polyphase_supported": (c_int, [POINTER(ConvDesc), c_int]),
    "t2v_conv_best_algo": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv_winograd_workspace_floats": (c_size_t, [POINTER(ConvDesc), c_int]),
    "t2v_conv_winograd_tile_rows": (c_int, [POINTER(ConvDesc)]),
    "t2v_conv_winograd_gemm_form": (c_int, [POINTER(ConvDesc), c_int]),
    "t2v_conv2d_forward_winograd": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p,
                                            c_void_p, c_int, c_void_p, c_void_p]),
    "t2v_conv2d_forward_winograd_stages": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p,
                                                   c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "t2v_conv2d_backward_weight_winograd_dy_norm": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_int, c_void_p,
                                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "t2v_conv2d_forward_winograd_keep_v": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_void_p, c_int, c_void_p,
                                                   c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "t2v_instance_norm_finalize": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_void_p, c_float, c_void_p]),
    "t2v_batch_norm_finalize": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_float, c_void_p]),
    "t2v_batch_norm_finalize_running": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_float, c_void_p,
                                                c_void_p, c_void_p, c_float, c_int]),
    "t2v_conv_backward_weight_workspace_floats": (c_size_t, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv_backward_weight_winograd_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv_backward_weight_winograd_workspace_floats": (c_size_t, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv2d_backward_weight_winograd": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_int,
                                                    c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "t2v_conv2d_backward_weight_winograd_stages": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_int,
                                                           c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                           c_int]),
    "t2v_conv2d_backward_weight": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_int, c_void_p,
                                           c_int, c_void_p, c_int, c_void_p]),
    "t2v_conv_backward_weight_strided_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv2d_backward_weight_strided": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_int, c_long, c_void_p,
                                                   c_int, c_long, c_void_p, c_int, c_void_p]),
    "t2v_conv_unpack_weight": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "t2v_conv_backward_data_winograd_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv_backward_data_winograd_weight_floats": (c_size_t, [POINTER(ConvDesc), c_int]),
    "t2v_conv_backward_data_winograd_scratch_floats": (c_size_t, [POINTER(ConvDesc), c_int]),
    "t2v_conv_pack_weight_transposed": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "t2v_conv2d_backward_data_winograd": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p, c_int, c_void_p,
                                                  c_void_p, c_void_p]),
    "t2v_conv_backward_data_winograd_takes_forward_weights": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "t2v_conv2d_backward_data_winograd_fw": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p, c_int, c_void_p,
                                                     c_void_p, c_void_p]),
    "t2v_conv_unpack_weight_into": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_int]),
    "t2v_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int]),
    "t2v_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float]),
    "t2v_zero": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "t2v_unzip2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "t2v_channel_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "t2v_reflect_pad_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "t2v_instance_norm_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "t2v_instance_norm_backward_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "t2v_act_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_long, c_void_p]),
    "t2v_avgpool3x3s2_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "t2v_sum_sq_diff_const_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_long, c_void_p]),
    "t2v_sum_abs_diff_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_long, c_void_p]),
    "t2v_sum_sq_diff_const": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_long, c_void_p, c_void_p]),
    "t2v_sum_abs_diff": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "t2v_sum_abs_diff_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int,
                                        c_void_p, c_void_p]),
    "t2v_sum_abs_diff_masked_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_long, c_int,
                                                 c_int, c_int, c_void_p]),
    "t2v_loss_terms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               c_int, c_void_p, c_void_p]),
    "t2v_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_double, c_double,
                              c_double, c_double, c_int]),
    "t2v_adam_step_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_double, c_double, c_double]),
    "t2v_batch_norm_update_running": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_float,
                                              c_float]),
    "t2v_instance_norm_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_long, c_int, c_int]),
    "t2v_flow_warp_composite": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_int, c_int]),
    "t2v_flow_warp_composite_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "t2v_avgpool3x3s2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "t2v_maxpool2x2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "t2v_maxpool2x2_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "t2v_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "t2v_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "t2v_pose_u8_to_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int]),
    "t2v_tensor2im_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long]),
    "t2v_copy_channels": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                  c_long]),
    "t2v_generator_num_layers": (c_int, [POINTER(GenDesc)]),
    "t2v_generator_layer_desc": (c_int, [POINTER(GenDesc), c_int, POINTER(ConvDesc), POINTER(c_int)]),
    "t2v_generator_workspace_bytes": (c_size_t, [POINTER(GenDesc)]),
    "t2v_generator_forward": (c_int, [c_void_p, c_void_p, POINTER(GenDesc), POINTER(Layer), c_int, POINTER(GenIO),
                                      c_void_p, c_size_t]),
    "t2v_generator_workspace_bytes_batch": (c_size_t, [POINTER(GenDesc), c_int]),
    "t2v_generator_forward_batch": (c_int, [c_void_p, c_void_p, POINTER(GenDesc), POINTER(Layer), c_int, POINTER(GenIO),
                                            c_int, c_void_p, c_size_t]),
    # host plumbing (ABI 14): what text2video_amd/leantorch.py allocates, copies and synchronises with
    "t2v_device_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "t2v_device_free": (c_int, [c_void_p, c_void_p]),
    "t2v_host_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "t2v_host_free": (c_int, [c_void_p, c_void_p]),
    "t2v_memcpy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "t2v_stream_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "t2v_stream_destroy": (c_int, [c_void_p, c_void_p]),
    "t2v_stream_synchronize": (c_int, [c_void_p, c_void_p]),
    "t2v_event_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "t2v_event_record": (c_int, [c_void_p, c_void_p, c_void_p]),
    "t2v_event_synchronize": (c_int, [c_void_p, c_void_p]),
    "t2v_event_destroy": (c_int, [c_void_p, c_void_p]),
    "t2v_device_synchronize": (c_int, [c_void_p]),
}
COPY_H2D, COPY_D2H, COPY_D2D = 1, 2, 3

_lib = None
_load_lock = __import__("threading").RLock()


def load():
    """Load libt2v_hip.so and bind every declared symbol.  Raises if anything is missing."""
    if _lib is not None:
        return _lib
    with _load_lock:      # (vid2vid/test.py brings the runtime up on a thread while the main thread still imports)
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libt2v_hip.so not found at %s: the HIP extension is required (no CPU fallback). "
            "Build it with `python __graft_entry__.py` or `make -C text2video_amd/csrc`." % LIB_PATH)
    # torch must initialise first: it bundles its own libamdhip64.so (same soname as /opt/rocm's).
    # Loaded in this order the one HIP runtime of the process is torch's and device pointers /
    # streams are shared; the other order would put two HIP runtimes in one process.
    # (The torch-free frame loop of vid2vid/test.py -- _xp.LEAN -- never imports torch: the library then brings the
    # ROCm installation's runtime in through its own RUNPATH.)
    from ._xp import LEAN
    if not LEAN:
        import torch  # noqa: F401
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    elif os.environ.get("T2V_HIP_RUNTIME", "torch") != "system":
        # torch-free process, same HIP runtime: the copy PyTorch bundles (located without importing torch) -- the one every
        # measurement and test of this tree ran on; T2V_HIP_RUNTIME=system takes the ROCm installation's instead
        import importlib.util
        spec = importlib.util.find_spec("torch")
        hip_rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.t2v_abi_version() != ABI_VERSION:
        raise RuntimeError("libt2v_hip.so ABI %d != binding ABI %d" % (lib.t2v_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, what=""):
    if status != T2V_OK:
        msg = load().t2v_last_error()
        raise RuntimeError("t2v %s failed (status %d): %s" % (what, status, msg.decode() if msg else "?"))


class Context:
    """Owns a t2v_ctx bound to one device."""

    def __init__(self, device=0):
        self.lib = load()
        self._h = c_void_p()
        check(self.lib.t2v_create(ctypes.byref(self._h), int(device)), "t2v_create")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self.lib.t2v_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
