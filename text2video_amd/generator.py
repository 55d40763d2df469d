"""Host side of the vid2vid generators on MI355X: checkpoint -> packed weights -> HIP forward.

Mirrors the reference-side objects for the hot path:
  * `HipGenerator`   ~ networks.CompositeGenerator / CompositeLocalGenerator (netG0 / netG1)
  * `Vid2VidModelG`  ~ models/vid2vid_model_G.py: `inference(A, B, inst)` with the 2-deep FIFO of
                       generated frames, `--no_first_img` first-frame rule and the spatial pyramid.
State-dict key names are upstream's (nn.Sequential indices), so a real
checkpoints/<name>/latest_net_G0.pth loads unchanged (SURVEY section 5 "Checkpoint / resume").
All device math goes through libt2v_hip.so; torch only owns memory and the stream.
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import os

from ._xp import torch     # the real torch, or leantorch under vid2vid/test.py's torch-free frame loop

from . import _lib, ops
from ._lib import GenDesc, GenIO, Layer, check


@dataclass
class GeneratorSpec:
    input_nc: int = 9          # opt.input_nc(3) * n_frames_G(3)
    prev_nc: int = 6           # (n_frames_G-1) * output_nc
    output_nc: int = 3
    ngf: int = 128
    n_downsample: int = 3      # --n_downsample_G
    n_blocks: int = 9          # --n_blocks (global) / --n_blocks_local (local)
    no_flow: bool = False      # --no_flow, implied by --openpose_only
    norm: str = "batch"        # --norm {batch: BatchNorm2d(affine) in train mode == IN+affine | instance}
    is_local: bool = False
    scale: int = 0             # spatial scale index s; flow multiplier 20*2^s

    @property
    def n_down(self):
        return 1 if self.is_local else self.n_downsample


def layer_keys(spec):
    """Canonical layer order of t2v_generator_forward -> [(conv_key, norm_key|None, kind)], kind in
    {'conv','convT','head','flow_w'}.  Key names per SURVEY App. A.1 (last paragraph) / A.2."""
    out = []
    n = spec.n_down

    def enc(prefix):
        out.append((prefix + ".1", prefix + ".2", "conv"))
        for i in range(n):
            out.append((prefix + ".%d" % (4 + 3 * i), prefix + ".%d" % (5 + 3 * i), "conv"))

    def rbs(prefix, first, count):
        for j in range(count):
            base = prefix + ".%d.conv_block" % (first + j)
            out.append((base + ".1", base + ".2", "conv"))
            out.append((base + ".5", base + ".6", "conv"))

    def ups(prefix, first):
        for i in range(n):
            out.append((prefix + ".%d" % (first + 3 * i), prefix + ".%d" % (first + 3 * i + 1), "convT"))

    if not spec.is_local:
        nb_enc = spec.n_blocks - spec.n_blocks // 2
        nb_res = spec.n_blocks // 2
        enc("model_down_seg"); rbs("model_down_seg", 4 + 3 * n, nb_enc)
        enc("model_down_img"); rbs("model_down_img", 4 + 3 * n, nb_enc)
        rbs("model_res_img", 0, nb_res); ups("model_up_img", 0)
        out.append(("model_final_img.1", None, "head"))
        if not spec.no_flow:
            rbs("model_res_flow", 0, nb_res); ups("model_up_flow", 0)
            out.append((("model_final_flow.1", "model_final_w.1"), None, "flow_w"))
    else:
        enc("model_down_seg")
        enc("model_down_img")
        rbs("model_up_img", 0, spec.n_blocks); ups("model_up_img", spec.n_blocks)
        out.append(("model_final_img.1", None, "head"))
        if not spec.no_flow:
            rbs("model_up_flow", 0, spec.n_blocks); ups("model_up_flow", spec.n_blocks)
            out.append((("model_final_flow.1", "model_final_w.1"), None, "flow_w"))
    return out


def _gen_desc(spec, H, W, conv_algo=0):
    return GenDesc(H, W, spec.input_nc, spec.prev_nc, spec.output_nc, spec.ngf, spec.n_downsample, spec.n_blocks,
                   int(spec.no_flow), int(spec.norm == "batch"), int(spec.is_local), 20.0 * (2 ** spec.scale), 1e-5,
                   conv_algo)


def layer_shapes(spec):
    """[(state-dict key, torch-layout shape, role)] for this generator; role in
    {'conv_w','convT_w','conv_b','norm_w','norm_b'}.  Shapes do not depend on H, W."""
    lib = _lib.load()
    gd = _gen_desc(spec, 64, 64)
    keys = layer_keys(spec)
    if lib.t2v_generator_num_layers(ctypes.byref(gd)) != len(keys):
        raise RuntimeError("layer list mismatch between generator.py and libt2v_hip.so: %s"
                           % lib.t2v_last_error().decode())
    out = []
    for i, (ck, nk, kind) in enumerate(keys):
        cd, xcs = _lib.ConvDesc(), ctypes.c_int()
        check(lib.t2v_generator_layer_desc(ctypes.byref(gd), i, ctypes.byref(cd), ctypes.byref(xcs)), "layer_desc")
        if kind == "flow_w":
            out += [(ck[0] + ".weight", (2, cd.Cin, cd.kH, cd.kW), "conv_w"), (ck[0] + ".bias", (2,), "conv_b"),
                    (ck[1] + ".weight", (1, cd.Cin, cd.kH, cd.kW), "conv_w"), (ck[1] + ".bias", (1,), "conv_b")]
        else:
            wshape = (cd.Cin, cd.Cout, 3, 3) if kind == "convT" else (cd.Cout, cd.Cin, cd.kH, cd.kW)
            out += [(ck + ".weight", wshape, "convT_w" if kind == "convT" else "conv_w"), (ck + ".bias", (cd.Cout,), "conv_b")]
        if nk is not None and spec.norm == "batch":
            out += [(nk + ".weight", (cd.Cout,), "norm_w"), (nk + ".bias", (cd.Cout,), "norm_b")]
    return out


def default_init_bound(weight_shape, transposed=False):
    """stdv of torch-0.4.1's `_ConvNd.reset_parameters` ($SP/torch/nn/modules/conv.py:40-47) for a weight of this
    shape: 1/sqrt(in_channels * kH * kW); a ConvTranspose2d weight is [in, out, kH, kW]."""
    cin = weight_shape[0] if transposed else weight_shape[1]
    return 1.0 / float(np.sqrt(cin * weight_shape[2] * weight_shape[3]))


def synthetic_state_dict(spec, seed=1, init="uniform_fan_in", flow_gain=1.0):
    """Deterministic random-init weights (numpy RNG, platform independent) in upstream key names.

    init='uniform_fan_in': U(-stdv, +stdv) for conv weight AND bias = torch-0.4.1 default
        ($SP/torch/nn/modules/conv.py:40-47: stdv = 1/sqrt(in_channels*kH*kW), where in_channels is weight.size(1)
        of a Conv2d and weight.size(0) of a ConvTranspose2d; pinned against the vendored file by
        tests/golden/make_transforms_golden.py -> transforms.npz `conv_init`) -- the BASELINE.md / SURVEY 8(d)
        config-2 weights.
    init='vid2vid': vid2vid's weights_init, N(0,0.02) conv weights [RECALL]; it leaves the conv biases at the
        default init above.
    Norm affine params (norm='batch'): gamma ~ N(1,0.02) (weights_init), beta ~ N(0,0.1) so the
    affine path is exercised.
    flow_gain scales model_final_flow's weight and bias: a random-init flow head times the x20
    flow multiplier yields +-40 px flows, which turns 1e-5 rounding differences into >1e-3 pixel
    differences once they compound through the frame recurrence; trained networks predict flows
    of a few pixels, which flow_gain=0.1 emulates (test conditioning only).
    """
    rng = np.random.default_rng(seed)
    sd = {}
    bound = 1.0
    for key, shape, role in layer_shapes(spec):
        if role in ("conv_w", "convT_w"):
            bound = default_init_bound(shape, role == "convT_w")
            a = rng.normal(0.0, 0.02, size=shape) if init == "vid2vid" else rng.uniform(-bound, bound, size=shape)
        elif role == "conv_b":
            a = rng.uniform(-bound, bound, size=shape)
        elif role == "norm_w":
            a = rng.normal(1.0, 0.02, size=shape)
        else:
            a = rng.normal(0.0, 0.1, size=shape)
        if key.startswith("model_final_flow."):
            a = a * flow_gain
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def upload_tensors(tensors, device):
    """{name: fp32 tensor} -> the same tensors on `device`.  (Round 4 measured a hand-made staging path for the 1.5 GB of a
    real checkpoint -- a ring of three pinned 64 MiB buffers filled from the memory-mapped file while the previous buffer's
    DMA runs -- against the runtime's own pageable copy: 0.98 s vs 0.40 s on the MI355X box.  Faulting the mapped pages in
    from one Python thread is the bottleneck, which the runtime's copy path avoids; the plain copy stays.  Several host
    threads copying their shares on streams of their own change nothing either: 0.27-0.31 s for 1.46 GB with 1, 2, 4, 8, 16
    threads -- the runtime serialises pageable copies.  Registering the mapped file as pinned memory first (hipHostRegister,
    scripts/upload_register_probe.py) moves the time instead of saving it: 0.22-0.25 s to register 1.46 GB + 0.034 s of copies
    at 43 GB/s = the plain copy's 0.28-0.30 s.  What does work is not touching the mapping at all: leantorch.upload_many
    reads the file span with os.preadv on four threads straight into page-locked chunks -- 0.066 s; it knows the tensors'
    file offsets because it read the container itself, which torch.load does not tell.)"""
    dev = torch.device(device)
    if hasattr(torch, "upload_many"):      # leantorch: a checkpoint's tensors go up as one file span, read by several threads
        return torch.upload_many(tensors, dev)
    return {k: v.detach().to(dev, torch.float32).contiguous() for k, v in tensors.items()}


class HipGenerator:
    """One generator scale resident on the GPU: packed weights + workspace + forward().

    Weights are packed lazily per frame geometry: the ResnetBlock convs use Winograd-transformed
    weights wherever the geometry supports it (t2v_generator_layer_desc reports the algorithm)."""

    def __init__(self, spec, device="cuda", conv_algo=None):
        import os
        self.spec = spec
        self.device = torch.device(device)
        self.ctx = ops.context(self.device)
        self.lib = self.ctx.lib
        self.keys = layer_keys(spec)
        self.conv_algo = int(os.environ.get("T2V_CONV_ALGO", "0")) if conv_algo is None else conv_algo
        self._raw = None       # upstream-named fp32 tensors on the device
        self._packed = None    # keeps the packed tensors of the current geometry alive
        self._layers = None
        self._ws = None
        self._ws_hw = None
        self._ws_batch = 0
        # packed weights + arena of the geometries seen last (two lanes with different frame sizes alternate every
        # step -- re-packing the Winograd filter transforms each time would cost a weight pack per frame).  Two entries: what
        # one lock-step pair of sequences can need; each holds ~5.5 GB of transformed weights for the full-size generator.
        self._geoms = {}
        self.max_cached_geometries = 2

    # -- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: upstream-named tensors (CPU or GPU)."""
        need = []
        for ck, nk, kind in self.keys:
            cks = ck if isinstance(ck, tuple) else (ck,)
            for k in cks:
                need += [k + ".weight", k + ".bias"]
            if nk is not None and self.spec.norm == "batch":
                need += [nk + ".weight", nk + ".bias"]
        with torch.cuda.device(self.device):
            self._raw = upload_tensors({k: sd[k] for k in need}, self.device)
        self._layers = None
        self._ws_hw = None
        self._geoms = {}
        return self

    def _pack(self, gd):
        import time
        t0 = time.perf_counter()
        n = len(self.keys)
        arr = (Layer * n)()
        keep = []
        for i, (ck, nk, kind) in enumerate(self.keys):
            cd, xcs = _lib.ConvDesc(), ctypes.c_int()
            check(self.lib.t2v_generator_layer_desc(ctypes.byref(gd), i, ctypes.byref(cd), ctypes.byref(xcs)),
                  "layer_desc")
            if kind == "flow_w":
                w = torch.cat([self._raw[ck[0] + ".weight"], self._raw[ck[1] + ".weight"]], 0)
                b = torch.cat([self._raw[ck[0] + ".bias"], self._raw[ck[1] + ".bias"]], 0)
            else:
                w, b = self._raw[ck + ".weight"], self._raw[ck + ".bias"]
            packed = ops.pack_conv_weight(w, cd, xcs.value)
            keep += [packed, b]
            arr[i].w = packed.data_ptr()
            arr[i].bias = b.data_ptr()
            if nk is not None and self.spec.norm == "batch":
                arr[i].gamma = self._raw[nk + ".weight"].data_ptr()
                arr[i].beta = self._raw[nk + ".bias"].data_ptr()
        torch.cuda.current_stream().synchronize()
        self._packed, self._layers = keep, arr
        self.pack_seconds = getattr(self, "pack_seconds", 0.0) + time.perf_counter() - t0

    # -- forward ---------------------------------------------------------------------------------
    def _workspace(self, H, W, batch=1):
        if self._ws_hw != (H, W):
            if self._ws_hw is not None:       # park the geometry in use
                self._geoms[self._ws_hw] = (self._packed, self._layers, self._gd, self._ws, self._ws_batch)
                while len(self._geoms) > self.max_cached_geometries:
                    self._geoms.pop(next(iter(self._geoms)))        # the oldest
            ent = self._geoms.pop((H, W), None)
            if ent is not None:
                self._packed, self._layers, self._gd, self._ws, self._ws_batch = ent
            else:
                gd = _gen_desc(self.spec, H, W, self.conv_algo)
                if self.lib.t2v_generator_workspace_bytes(ctypes.byref(gd)) == 0:
                    raise RuntimeError("generator: %s" % self.lib.t2v_last_error().decode())
                self._pack(gd)
                self._ws, self._ws_batch = None, 0
                self._gd = gd
            self._ws_hw = (H, W)
        if self._ws is None or self._ws_batch < batch:     # one arena, sized for the largest batch seen
            nbytes = self.lib.t2v_generator_workspace_bytes_batch(ctypes.byref(self._gd), batch)
            if nbytes == 0:
                raise RuntimeError("generator: %s" % self.lib.t2v_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def forward(self, pose, prev, use_raw_only=False, coarse_img_feat=None, coarse_flow_feat=None,
                want=("out",)):
        """pose [H,W,round_up4(input_nc)], prev [H,W,round_up4(prev_nc)] NHWC fp32 on the device.
        Returns dict of NHWC tensors for the names in `want` ⊆ {out, raw, flow_w, img_feat, flow_feat}."""
        return self.forward_batch([pose], [prev], [use_raw_only], [coarse_img_feat], [coarse_flow_feat], want)[0]

    def forward_batch(self, poses, prevs, use_raw_only, coarse_img_feat=None, coarse_flow_feat=None, want=("out",)):
        """N independent images (one frame each of N sequences advanced in lock-step) through one pass of the layer
        list: t2v_generator_forward_batch.  All arguments are lists of length N (same geometry); returns a list of N
        dicts as forward() does.  N = 1 is t2v_generator_forward."""
        if self._raw is None:
            raise RuntimeError("HipGenerator.forward before load_state_dict")
        N = len(poses)
        if not 1 <= N <= _lib.MAX_BATCH:
            raise ValueError("forward_batch: %d images (1..%d)" % (N, _lib.MAX_BATCH))
        H, W = poses[0].shape[0], poses[0].shape[1]
        if any(tuple(p.shape[:2]) != (H, W) for p in poses):
            raise ValueError("forward_batch: one frame geometry per batch")
        coarse_img_feat = coarse_img_feat or [None] * N
        coarse_flow_feat = coarse_flow_feat or [None] * N
        ws = self._workspace(H, W, N)
        ios = (GenIO * N)()
        results = []
        for i in range(N):
            res = {}

            def buf(name, c, res=res):
                if name in want:
                    res[name] = torch.empty(H, W, c, dtype=torch.float32, device=self.device)
                    return res[name].data_ptr()
                return None

            io = ios[i]
            io.pose, io.prev = poses[i].data_ptr(), prevs[i].data_ptr()
            io.coarse_img_feat = coarse_img_feat[i].data_ptr() if coarse_img_feat[i] is not None else None
            io.coarse_flow_feat = coarse_flow_feat[i].data_ptr() if coarse_flow_feat[i] is not None else None
            io.use_raw_only = int(use_raw_only[i])
            res["out"] = torch.empty(H, W, 4, dtype=torch.float32, device=self.device)
            io.out = res["out"].data_ptr()
            io.raw = buf("raw", 4)
            io.flow_w = buf("flow_w", 4) if not self.spec.no_flow else None
            io.img_feat = buf("img_feat", self.spec.ngf)
            io.flow_feat = buf("flow_feat", self.spec.ngf) if not self.spec.no_flow else None
            results.append(res)
        check(self.lib.t2v_generator_forward_batch(self.ctx.handle, ops._stream(), ctypes.byref(self._gd), self._layers,
                                                   len(self.keys), ios, N, ctypes.c_void_p(ws.data_ptr()), ws.numel()),
              "generator_forward")
        return results


class Recurrence:
    """The state one sequence carries from frame to frame: per spatial scale (finest first) the NHWC [H,W,8] FIFO of
    the tG-1 previous generated frames (`fake_B_prev` upstream), None before the first frame."""

    def __init__(self):
        self.prev = None
        self._spare = None     # ping-pong partner of every FIFO buffer (channels >= prev_nc stay zero)

    def reset(self):
        self.prev = None
        self._spare = None


class Vid2VidModelG:
    """`Vid2VidModelG.inference` on the HIP path.

    nets: [HipGenerator scale 0 (coarsest, global), scale 1 (local), ...]
    One model can advance several independent sequences in lock-step (`inference_nhwc_batch`): each has its own
    Recurrence; `self.state` is the one the single-sequence entry points use.
    """

    def __init__(self, nets, n_frames_G=3, output_nc=3, no_first_img=True):
        self.nets = nets
        self.n_scales = len(nets)
        self.tG = n_frames_G
        self.output_nc = output_nc
        self.no_first_img = no_first_img
        self.device = nets[0].device
        self.state = Recurrence()

    @property
    def prev(self):
        """per scale: NHWC [H,W,8] FIFO of the tG-1 previous outputs (None before the first frame)"""
        return self.state.prev

    @prev.setter
    def prev(self, value):
        self.state.prev = value
        self.state._spare = None

    def reset(self):
        """`model.fake_B_prev = None` on data['change_seq']."""
        self.state.reset()

    def load_prev(self, fake_B_prev, state=None):
        """Set the FIFO from reference-layout tensors: list (finest first) of [tG-1, 3, h, w]."""
        pcs = ops.round_up(self.nets[0].spec.prev_nc, 4)
        st = state or self.state
        st.prev = [ops.nchw_to_nhwc(p.reshape(-1, p.shape[-2], p.shape[-1]).to(self.device).contiguous(), pcs)
                   for p in fake_B_prev]
        st._spare = None

    @torch.no_grad()
    def inference_nhwc_batch(self, poses, states):
        """One frame of each of N independent sequences: poses[i] is sequence i's [H,W,round_up4(3*tG)] fp32 NHWC window
        (oldest frame first), states[i] its Recurrence (updated in place).  Returns the N [H,W,4] frames (RGB0).
        Frame i is the frame inference_nhwc computes for sequence i alone."""
        N = len(poses)
        firsts = [st.prev is None for st in states]
        if any(firsts) and not self.no_first_img:
            raise NotImplementedError("first-frame generator: the reference always passes --no_first_img")
        pcs = ops.round_up(self.nets[0].spec.prev_nc, 4)
        pyr = []        # per sequence: spatial pyramid, index 0 = finest
        for i in range(N):
            ps = [poses[i]]
            for _ in range(1, self.n_scales):
                ps.append(ops.avgpool3x3s2(ps[-1]))
            pyr.append(ps)
            if firsts[i]:
                states[i].prev = [torch.zeros(p.shape[0], p.shape[1], pcs, dtype=torch.float32, device=self.device)
                                  for p in ps]
            if states[i]._spare is None:
                states[i]._spare = [torch.zeros_like(p) for p in states[i].prev]
        raw_only = [self.no_first_img and f for f in firsts]
        img_feat, flow_feat = [None] * N, [None] * N
        outs = None
        for s in range(self.n_scales):
            si = self.n_scales - 1 - s
            net = self.nets[s]
            want = ("out",) if s == self.n_scales - 1 else \
                (("out", "img_feat") if net.spec.no_flow else ("out", "img_feat", "flow_feat"))
            rs = net.forward_batch([pyr[i][si] for i in range(N)], [states[i].prev[si] for i in range(N)], raw_only,
                                   img_feat, flow_feat, want)
            outs = [r["out"] for r in rs]
            img_feat, flow_feat = [r.get("img_feat") for r in rs], [r.get("flow_feat") for r in rs]
            # fake_B_prev = cat(fake_B_prev[1:], fake_B): shift the FIFO into its ping-pong partner
            nc = self.output_nc
            for i in range(N):
                st = states[i]
                newp = st._spare[si]
                for f in range(self.tG - 2):
                    ops.copy_channels(st.prev[si], (f + 1) * nc, newp, f * nc, nc)
                ops.copy_channels(outs[i], 0, newp, (self.tG - 2) * nc, nc)
                st.prev[si], st._spare[si] = newp, st.prev[si]
        return outs

    def lockstep_pays(self, H, W):
        """Whether advancing several sequences of H x W frames through ONE batched call is worth it (run_test asks per
        geometry).  Measured end to end with the flow branch, two sequences (profiles/r04_ab_lockstep.txt): 512x320 139.3 vs
        135.2 fps, 512x680 66.3 vs 65.2, 512x512 88.3 vs 86.0 -- the batch wins 2-3 % everywhere since the ragged GEMM of a
        batch-2 launch runs one block per CU; before that, two single calls on 160x128 tiles were as fast at
        512x320.  (A geometry rule would go here; today the batch wins everywhere measured.)"""
        return True

    def inference_nhwc(self, pose):
        """pose: [H,W,round_up4(3*tG)] fp32 NHWC window (oldest frame first).  Returns [H,W,4] (RGB0)."""
        return self.inference_nhwc_batch([pose], [self.state])[0]

    @torch.no_grad()
    def inference(self, A, B=None, inst=None):
        """Reference signature.  A: [1, tG, 3, H, W] fp32 device tensor in [-1,1].
        Returns (fake_B [1,3,H,W], real_A last frame [3,H,W]) like upstream."""
        _, tG, nc, H, W = A.shape
        pose = ops.nchw_to_nhwc(A.reshape(tG * nc, H, W).contiguous())
        out = self.inference_nhwc(pose)
        fake_B = ops.nhwc_to_nchw(out, self.output_nc).unsqueeze(0)
        return fake_B, A[0, -1]
