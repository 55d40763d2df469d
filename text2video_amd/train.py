"""Train step of the vid2vid pose model on the HIP path.

torch.autograd is used as the TAPE only: every differentiable op below is a `torch.autograd.Function`
whose forward and backward are calls into libt2v_hip.so (implicit-GEMM conv, fused norm statistics,
weight-gradient kernel, norm backward, ...).  Tensors are NHWC fp32 on the device; parameters are
torch-layout fp32 `nn.Parameter`s (upstream state-dict names) that are re-packed for the kernels each
step.  Channel concat / crop / detach / scalar loss arithmetic stay torch-native plumbing.

Scope (what the reference's training command README.md:171-176 exercises): generator WITH its flow branch
(flow / weight heads, flow-warp compositor: every loss on the blended frame back-propagates through
raw*w + warp*(1-w) into flow, weight and raw -- `_WarpComposite`), or without it (--no_flow), multiscale image
discriminator (--num_D 2), face discriminator (--add_face_disc), temporal discriminators, LSGAN +
feature-matching losses, the VGG19 perceptual loss (frozen torchvision weights supplied with --vgg_weights), the
flow / warp / weight losses against a SUPPLIED reference flow (zero flow + its confidence mask by default:
FlowNet2, which produces it upstream, is not in the reference tree -- SURVEY 8d config 5), Adam(lr 2e-4,
beta1 0.5), data-parallel gradient all-reduce.
"""
import contextlib
import os
import time

import torch

from . import ops
from .backward import ConvDataGrad
from .generator import GeneratorSpec, layer_keys, synthetic_state_dict  # noqa: F401


# ------------------------------------------------------------------------------------------------
# differentiable building blocks
# ------------------------------------------------------------------------------------------------
def cached_pack(w, key, make):
    """Packed (or Winograd-transformed) layout of parameter `w`, kept until the optimiser next writes it: a step
    runs every generator layer once per frame and every discriminator layer on real and fake inputs, all with the
    same weights.  The cache lives on the parameter object itself (freed with it; autograd hands the same object to
    backward).  T2V_TRAIN_PACK_CACHE=0 re-packs per call (saves the memory of the transformed copies)."""
    if os.environ.get("T2V_TRAIN_PACK_CACHE", "1") == "0":
        return make()
    w = getattr(w, "_t2v_owner", w)    # a detached view of a parameter (frozen passes) shares its parameter's cache
    ent = getattr(w, "_t2v_packs", None)
    if ent is None or ent[0] != w._version:
        ent = (w._version, {}, {}, set())       # version, copies, how each was made, which were used since the last write
        w._t2v_packs = ent
        w._t2v_pack_event = None
        w._t2v_pack_made = {}
    elif getattr(w, "_t2v_pack_event", None) is not None:
        # packed ahead on the side stream (prefetch_packs): a use waits for that batch (on whichever stream it runs; a wait
        # for an event that has completed costs nothing)
        torch.cuda.current_stream().wait_event(w._t2v_pack_event)
    if key not in ent[1]:
        ent[1][key] = make()
        ent[2][key] = make
        if w.is_cuda:      # made on the spot (a first step, T2V_PACK_PREFETCH=0): by whom, and when it is complete
            cur = torch.cuda.current_stream()
            made = getattr(w, "_t2v_pack_made", None)
            if made is None:
                made = w._t2v_pack_made = {}
            made[key] = (cur, cur.record_event())
    elif w.is_cuda:
        # the discriminators' layers are used from two streams in one step (their own backward pass on the trainer's
        # `_d_stream`, the generator's pass through them on the current one): a copy made on the other stream is waited for
        m = (getattr(w, "_t2v_pack_made", None) or {}).get(key)
        if m is not None and m[0] != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_event(m[1])
    ent[3].add(key)
    return ent[1][key]


def invalidate_packs(p):
    """the optimiser has written `p` (through its raw pointer: no version bump): drop the packed copies, remember how
    they were made"""
    ent = getattr(p, "_t2v_packs", None)
    # (only the copies the last step actually used: a geometry or mode that is gone is not made again for ever)
    p._t2v_repack = {k: ent[2][k] for k in ent[3]} if (ent is not None and ent[3]) else None
    p._t2v_packs = None
    p._t2v_pack_event = None
    p._t2v_pack_made = None


def prefetch_packs(params):
    """After the optimiser step: every packed / Winograd-transformed / transposed copy the last step used is made again
    from the new weights NOW, on the side stream -- ~150 small memory-bound kernels per step (36 filter transforms of
    151 MB each among them) that depend on nothing but the weights and would otherwise sit in front of each layer's
    first use on the critical path; here they run under the next step's first convs.  T2V_PACK_PREFETCH=0: lazily."""
    todo = [(p, p._t2v_repack) for p in params if getattr(p, "_t2v_repack", None)]
    if not todo or os.environ.get("T2V_PACK_PREFETCH", "1") == "0" or not todo[0][0].is_cuda:
        for p, _ in todo:
            p._t2v_repack = None
        return
    with wgrad_fork(gemm=False):
        for p, makers in todo:
            p._t2v_packs = (p._version, {k: mk() for k, mk in makers.items()}, dict(makers), set())
            p._t2v_repack = None
        ev = torch.cuda.current_stream().record_event()
    for p, _ in todo:
        p._t2v_pack_event = ev


# ------------------------------------------------------------------------------------------------
# gradient slots: every trainable parameter owns a slice ("slot") of a persistent flat exchange bucket
# (GradBuckets below).  Inside a train step the backward nodes DELIVER their weight / bias / affine gradients
# straight into the slot -- the first node of a step writes, later nodes of the same parameter add (kernels with
# an accumulate flag) -- and return None to autograd: no AccumulateGrad adds, no torch.cat / copy-back around the
# collective, and a bucket's all-reduce starts the moment its last gradient has landed.
# ------------------------------------------------------------------------------------------------
class GradSlot:
    __slots__ = ("owner", "bucket", "idx", "view", "expect", "got", "_filled", "zero_valid")

    def __init__(self, owner, bucket, idx, view):
        self.owner, self.bucket, self.idx, self.view = owner, bucket, idx, view
        self.expect = self.got = 0
        self._filled = False      # the slot holds this step's (partial) gradient
        # the slot's memory holds exact zeros put there by deliver(zero=True) and nothing has written it since: the ~100
        # biases in front of norm layers get that delivery every step -- one 4 KB memset each, once, instead of per step
        # (an exchange averages zeros with the other ranks' zeros: the same layers are zero-delivered on every rank)
        self.zero_valid = False

    @property
    def filled(self):
        return self._filled

    @filled.setter
    def filled(self, v):
        self._filled = bool(v)
        if v:                     # every writer of the slot's memory marks it filled: the zeros are gone
            self.zero_valid = False


def grad_slot(p):
    """the slot of parameter `p` if its buckets are collecting (inside a train step), else None"""
    sl = getattr(p, "_t2v_gslot", None)
    return sl if (sl is not None and sl.owner.collecting) else None


def expect_gradient(p):
    """forward side: one more backward node will deliver a gradient for `p` in this step"""
    sl = grad_slot(p) if p is not None and p.requires_grad else None
    if sl is not None:
        sl.expect += 1


def deliver(sl, tensor=None, zero=False):
    """backward side: add `tensor` (or nothing / an exact zero) to slot `sl` and count the node as done"""
    if tensor is not None:
        _acc(sl.view, tensor, not sl.filled)
        sl.filled = True
    elif zero and not sl.filled:
        if not sl.zero_valid:
            _zero(sl.view)
        sl._filled, sl.zero_valid = True, True
    sl.owner.node_done(sl)


_WG_BATCH = [False]
_DW_PAIR = [False]     # inside a train-step scope: a direct-kernel layer used by exactly two frames reduces both in one launch


def _paired_direct_wgrad(w, x, dc, fdesc, slot):
    """Direct (non-Winograd) weight gradient of a layer that ran several times in this step's graph (counted in forward:
    `w._t2v_dw_uses`): the backward nodes leave their (x, dY) on the weight, the LAST one to arrive launches ONE reduction over
    all of them -- the kernel pays a fixed ~8 stage times per block, so one long reduction beats several short ones:
      * two uses of one image each (the generator's stride-2 / transposed layers on the clip's two frames):
        ops.conv2d_backward_weight_pair -- two base pointers, no copy; blocks 36 -> 73 stages, 0.64 -> 0.73 of peak;
      * anything else (the discriminators' layers: real, fake and raw pass of two frames each -- round 5): the operands are
        concatenated (a few MB per layer) and reduced as one batch: 3 launches of ~180 us at 0.39 of peak -> one.
    Returns (taken, dW or None): not taken -> the caller reduces this node alone."""
    uses = getattr(w, "_t2v_dw_uses", 0)
    if not _DW_PAIR[0] or uses < 2:
        return False, None
    st = getattr(w, "_t2v_dw_stash", None)
    if st is None:
        st = w._t2v_dw_stash = [fdesc]
    st.append((x, dc))
    if len(st) - 1 < uses:
        if slot is not None:
            slot.owner.node_done(slot)
        return True, None
    parts = st[1:]
    w._t2v_dw_stash, w._t2v_dw_uses = None, 0
    return True, _reduce_stashed(parts, fdesc, slot)


_WGRAD_MAX_BYTES = 0x7fff0000        # the weight-gradient kernel addresses its operands with 32-bit byte offsets (capi.hip)


def direct_weight_gradient(x, dc, fdesc):
    """packed dW of a batch [B,...] in as few launches as the kernel's 2 GiB operand limit allows: ONE for every shape of the
    512 / 1024 configs; a batch whose x or dY (or padded x) would pass the limit -- many frames per GPU at 2048x1024 -- is
    reduced in chunks of images, accumulating (three passes in one batch are 3x the single pass)."""
    B = x.shape[0]
    padded = (x.shape[1] + 8) * (x.shape[2] + 8) * x.shape[3] if x.dim() == 4 else x[0].numel()     # (the kernel pads x by <= 3 px)
    per = 4 * max(padded, dc[0].numel(), 1)
    step = max(1, min(B, (_WGRAD_MAX_BYTES - 1) // per))
    dwp = None
    for i in range(0, B, step):
        dwp = ops.conv2d_backward_weight(x[i:i + step], dc[i:i + step], fdesc, accumulate_into=dwp)
    return dwp


def _reduce_stashed(parts, fdesc, slot, from_node=True):
    """one weight-gradient launch over the stashed (x, dY) of a layer; into the bucket slot (-> None) or returned.
    from_node: called by the backward node that brought the last operands (side stream, the node counts as delivered);
    False: the flush after the backward pass (current stream; every node has been counted already)"""
    flat = [t for pr in parts for t in pr]
    xcs = parts[0][0].shape[-1]
    with (wgrad_fork(*flat) if (from_node and slot is not None and wgrad_stream_on(parts[0][0])) else contextlib.nullcontext()):
        if len(parts) == 2 and all(pr[0].shape[0] == 1 for pr in parts) and \
                ops.backward_weight_strided_supported(fdesc, xcs, parts[0][1].shape[-1]):
            (x0, dc0), (x1, dc1) = parts
            dwp = ops.conv2d_backward_weight_pair(x0[0], dc0[0], x1[0], dc1[0], fdesc)
        elif len(parts) == 1:
            dwp = direct_weight_gradient(parts[0][0], parts[0][1], fdesc)
        else:
            dwp = direct_weight_gradient(torch.cat([pr[0] for pr in parts]), torch.cat([pr[1] for pr in parts]), fdesc)
        if slot is not None:
            ops.unpack_conv_weight_into(dwp, fdesc, xcs, slot.view, slot.filled)
    if slot is not None:
        slot.filled = True
        if from_node:
            slot.owner.node_done(slot)
        return None
    return ops.unpack_conv_weight(dwp, fdesc, xcs)


@contextlib.contextmanager
def batched_weight_gradients(params):
    """Scope of one train step (forward and backward inside): layers that run once per frame reduce their
    Winograd-domain weight gradients over all frames at once.  The per-weight counters are cleared on entry, so a
    graph that was built but never back-propagated cannot leave a stale count behind.  T2V_WGRAD_BATCH=0: off."""
    for p in params:
        p._t2v_wg_images, p._t2v_wg_state = 0, None
        p._t2v_wg_seen = 0
        p._t2v_dw_uses, p._t2v_dw_stash = 0, None
    _WG_BATCH[0] = os.environ.get("T2V_WGRAD_BATCH", "1") != "0"
    _DW_PAIR[0] = os.environ.get("T2V_WGRAD_PAIR", "1") != "0"
    _WG_SIDE["gemm"] = False
    if params and params[0].is_cuda:
        side_gemm_hint()
    try:
        yield
    except BaseException:
        _WG_BATCH[0] = False
        _DW_PAIR[0] = False
        _WG_SIDE["gemm"] = False
        # an exception inside the step (OOM, an asynchronous error): let go of what the nodes parked on the weights -- operand
        # lists, kept-V workspaces: GPU memory a retry needs
        for p in params:
            p._t2v_dw_stash, p._t2v_dw_uses, p._t2v_wg_state, p._t2v_wg_images, p._t2v_wg_seen = None, 0, None, 0, 0
        raise
    _WG_BATCH[0] = False
    _DW_PAIR[0] = False
    _WG_SIDE["gemm"] = False
    if params and params[0].is_cuda:
        side_gemm_hint()
    # a backward pass inside this scope that did not end with flush_pending_weight_gradients() would silently lose the
    # gradients still parked on the weights (the first half of a pair, transformed slots waiting for their reduction)
    left = [i for i, p in enumerate(params) if getattr(p, "_t2v_dw_stash", None) is not None
            or (getattr(p, "_t2v_wg_state", None) is not None and p._t2v_wg_state[1] > 0)]
    for p in params:
        p._t2v_dw_stash, p._t2v_dw_uses, p._t2v_wg_state, p._t2v_wg_images = None, 0, None, 0
        # what the layer saw in this step is what the next one prepares for (_keep_v_slot)
        p._t2v_wg_expect = getattr(p, "_t2v_wg_seen", 0)
    if left:
        raise RuntimeError("batched_weight_gradients: %d parameter(s) (first: #%d) left the scope with an unreduced weight "
                           "gradient -- call flush_pending_weight_gradients(params, grads) after the backward pass" % (len(left), left[0]))


# ---- weight gradients on a second stream ---------------------------------------------------------------------------
# Nothing in the backward pass waits for a weight gradient: the chain that has to run in order is dy -> norm backward ->
# data gradient -> the next layer's dy.  With gradients delivered straight into bucket slots (GradBuckets) the weight-
# gradient kernels of a node can therefore run on a side stream, next to the data-gradient kernels of the following
# layers: the tails of one launch (4.5-round grids, split-K partial combines, transforms of 36 small matrices) are
# filled by the other stream's blocks, as in the two-stream inference frames.  The side stream waits for the node's dc;
# the main stream waits for the side stream before anything reads a slot (bucket collectives, absorb, finish).
# T2V_WGRAD_STREAM=0: everything on one stream.
# From the step's first weight gradient on the side stream to the end of the step BOTH streams carry fixed-grid GEMMs (data
# gradient | Winograd-domain weight gradient).  A two-per-CU grid keeps every CU full until its last block leaves: the other
# stream's launch -- and the bandwidth-bound kernels between two of the main stream's GEMMs -- queue behind it, the step was
# the sum of its kernels (585 us per ResnetBlock layer of the second frame's backward pass against 591 us of kernels).  With
# the overlap hint 2 (ops.set_overlap_hint) those kernels launch ONE block per CU and are resident side by side: 513 us per
# layer, the same bits (profiles/r06_train_two_queues_{two,one}_per_cu.txt).  The hint is per thread and the backward nodes run on the
# autograd engine's thread: every node (and the flush on the calling thread) sets it from the shared flag.
# T2V_TRAIN_SK_HINT=0: two blocks per CU throughout.
_WG_SIDE = {"stream": None, "pending": False, "gemm": False}


def wgrad_stream_on(t):
    return t.is_cuda and not _WG_SIDE.get("inline") and os.environ.get("T2V_WGRAD_STREAM", "1") != "0"


def side_gemm_hint():
    """this thread's launches from here on: one block per CU for the fixed-grid GEMMs while the side stream has GEMMs of its
    own in this step, the library's default otherwise"""
    ops.set_overlap_hint(2 if _WG_SIDE["gemm"] else 0)


@contextlib.contextmanager
def wgrad_fork(*tensors, gemm=True):
    """kernels launched in this scope run on the side stream, after everything the current stream holds so far; the
    tensors named (inputs allocated on the current stream) stay allocated until the side stream is done with them.
    gemm: the scope launches weight-gradient GEMMs (not the repacking of weights after the optimiser step)"""
    if _WG_SIDE["stream"] is None:
        _WG_SIDE["stream"] = torch.cuda.Stream()
    side = _WG_SIDE["stream"]
    side.wait_stream(torch.cuda.current_stream())
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _WG_SIDE["pending"] = True
    if gemm and not _WG_SIDE["gemm"] and os.environ.get("T2V_TRAIN_SK_HINT", "1") != "0":
        _WG_SIDE["gemm"] = True
        side_gemm_hint()
    with torch.cuda.stream(side):
        yield


def wgrad_join():
    """the current stream waits for the weight gradients in flight on the side stream"""
    if _WG_SIDE["pending"]:
        torch.cuda.current_stream().wait_stream(_WG_SIDE["stream"])
        _WG_SIDE["pending"] = False


def _keep_v_slot(w, x, ddesc, xcs, ycs):
    """Forward side of a batched Winograd-domain weight gradient (use number `w._t2v_wg_images` of this step, not counted
    yet): from a layer's second step on, the workspace of its weight gradient exists BEFORE the forward pass -- sized for
    the images the previous step counted (`_t2v_wg_expect`) -- and the forward conv writes its input transform V straight
    into a slot of it: backward transforms dy only (72 input transforms of 13.5 us fewer per step of the 512x512 config).
    The slots are handed out from the top, so that the backward nodes -- which arrive in reverse -- fill them in the order
    the un-kept path would: the same reduction order, the same bits.  Returns (workspace, slots, slot) or None: not kept
    (first step, a batch, more uses than expected, T2V_WGRAD_KEEP_V=0)."""
    expect = getattr(w, "_t2v_wg_expect", 0)
    idx = getattr(w, "_t2v_wg_images", 0)
    if x.shape[0] != 1 or expect < 1 or idx >= expect or xcs != ddesc.Cin or os.environ.get("T2V_WGRAD_KEEP_V", "1") == "0":
        return None
    st = getattr(w, "_t2v_wg_state", None)
    if st is None:
        if idx != 0:
            return None       # an earlier use of this step went without: stay on that path
        st = [ops.backward_weight_winograd_workspace(ddesc, xcs, expect, x.device), 0, _desc_key(ddesc, xcs), ddesc, xcs, ycs,
              expect, [False] * expect]
        w._t2v_wg_state = st
    if len(st) < 8 or st[2] != _desc_key(ddesc, xcs):
        return None
    return st[0], expect, expect - 1 - idx


def _kept_winograd_wgrad(w, dc, fdesc, kept, slot=None, info=None, lazy_dc=None):
    """backward side of _keep_v_slot: A dy A^T of this node's image into ITS slot; the node that completes the set -- every
    use the forward pass counted has come back -- zeroes the slots nobody filled (a step with fewer uses than expected)
    and runs the one reduction.  lazy_dc = (conv output, gradient behind the norm, mean_rstd, gamma, beta, relu, sums)
    instead of dc: the gradient in front of the norm is formed inside the transform."""
    ws, total, k = kept
    st = w._t2v_wg_state
    assert st is not None and st[0] is ws and not st[7][k]
    if info is not None:
        info[:] = [ws, total, k]
    if lazy_dc is not None:
        ops.conv2d_backward_weight_winograd_dy_norm(*lazy_dc, fdesc, ws, total, k, st[4])
    else:
        ops.conv2d_backward_weight_winograd_dy(dc, fdesc, ws, total, k, st[4])
    st[7][k] = True
    st[1] += 1
    last = st[1] == min(getattr(w, "_t2v_wg_images", 0), total)
    dw = None
    if last:
        _zero_unfilled_slots(st)
        if slot is not None:
            with (wgrad_fork(ws) if wgrad_stream_on(ws) else contextlib.nullcontext()):
                ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, st[4], st[5], out=slot.view, accumulate=slot.filled)
            slot.filled = True
        else:
            dw = ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, st[4], st[5])
        w._t2v_wg_state, w._t2v_wg_images = None, 0
    if slot is not None:
        slot.owner.node_done(slot)
    return dw


def _zero_unfilled_slots(st):
    """slots of a kept-V workspace whose A dy A^T never came (and whose V may never have been written): zero both -- a zero
    image contributes nothing, whatever bit pattern the allocation held would"""
    ws, fdesc, xcs, total, filled = st[0], st[3], st[4], st[6], st[7]
    if all(filled):
        return
    tp = ops.winograd_tile_rows(fdesc)
    nv = 36 * total * tp * xcs
    v = ws[:nv].view(36, total, tp * xcs)
    md = ws[nv:nv + 36 * total * tp * fdesc.Cout].view(36, total, tp * fdesc.Cout)
    for k, f in enumerate(filled):
        if not f:
            v[:, k].zero_()
            md[:, k].zero_()


def _batched_winograd_wgrad(w, x, dc, fdesc, slot=None, info=None):
    """Weight gradient of one use of a layer whose forward counted `w._t2v_wg_images` images in this graph: the
    images are transformed into their slots of a workspace kept on the weight; the node that brings the last ones
    runs the single reduction over all of them and returns dW, the earlier ones return None (a zero gradient --
    autograd sums the nodes' results).  One K = images x tiles reduction instead of one short one per frame."""
    total = getattr(w, "_t2v_wg_images", 0)
    st0 = getattr(w, "_t2v_wg_state", None)
    if st0 is not None and len(st0) >= 8:
        total = 0     # this step's workspace belongs to the uses that kept their V (_keep_v_slot); this one did not
    if total < x.shape[0]:     # no count on this object: reduce on the spot
        if slot is not None:
            with (wgrad_fork(x, dc) if wgrad_stream_on(x) else contextlib.nullcontext()):
                ops.conv2d_backward_weight_winograd(x, dc, fdesc, accumulate_into=slot.view if slot.filled else None,
                                                    out=slot.view)
            slot.filled = True
            slot.owner.node_done(slot)
            return None
        return ops.conv2d_backward_weight_winograd(x, dc, fdesc)
    st = getattr(w, "_t2v_wg_state", None)
    if st is None:
        st = [ops.backward_weight_winograd_workspace(fdesc, x.shape[-1], total, x.device), 0, _desc_key(fdesc, x.shape[-1]),
              fdesc, x.shape[-1], dc.shape[-1]]
        w._t2v_wg_state = st
    assert st[2] == _desc_key(fdesc, x.shape[-1]), "one layer, two geometries in one step: set T2V_WGRAD_BATCH=0"
    ws, done = st[0], st[1]
    if info is not None:      # where this node's A dy A^T sits: the data gradient reads it too (transposed algorithm)
        info[:] = [ws, total, done]
    last = done + x.shape[0] == total
    if slot is not None:     # the reduction writes (adds to) the parameter's bucket slot; every node counts as delivered
        if wgrad_stream_on(x):
            # transforms here (the data gradient of this node reads A dy A^T), the reduction over all slots on the side
            ops.conv2d_backward_weight_winograd_stages(x, dc, fdesc, ws, total, done, False)
            if last:
                with wgrad_fork(ws):
                    ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, x.shape[-1], dc.shape[-1], out=slot.view,
                                                               accumulate=slot.filled)
        else:
            ops.conv2d_backward_weight_winograd_stages(x, dc, fdesc, ws, total, done, last, out=slot.view,
                                                       accumulate=slot.filled)
        dw = None
        if last:
            slot.filled = True
    else:
        dw = ops.conv2d_backward_weight_winograd_stages(x, dc, fdesc, ws, total, done, last)
    if last:
        w._t2v_wg_state, w._t2v_wg_images = None, 0
    else:
        st[1] = done + x.shape[0]
    if slot is not None:
        slot.owner.node_done(slot)
    return dw


def flush_pending_weight_gradients(params, grads):
    """After the backward pass: a layer whose forward counted more images than its backward nodes delivered (part of
    the graph fed no loss -- e.g. the flow head of a raw-only first frame when no flow loss is on) has transformed
    slots waiting for a reduction that no node will run.  Reduce what is there: the slots that were never filled are
    zeroed (a zero image contributes nothing) and the reduction runs over the whole workspace.  Returns `grads` with
    those gradients filled in."""
    out = list(grads)
    wgrad_join()
    if params and params[0].is_cuda:
        side_gemm_hint()       # (this thread: the backward nodes set theirs on the autograd engine's)
    for i, p in enumerate(params):
        half = getattr(p, "_t2v_dw_stash", None)
        if half is not None:        # a layer whose remaining uses never came back: reduce the ones that did
            dw = _reduce_stashed(half[1:], half[0], grad_slot(p), from_node=False)
            if dw is not None:
                out[i] = dw if out[i] is None else out[i] + dw
        p._t2v_dw_stash, p._t2v_dw_uses = None, 0
        st = getattr(p, "_t2v_wg_state", None)
        if st is None:
            continue
        if len(st) >= 8:      # kept V (_keep_v_slot): the slots that did come back, the others zeroed
            if st[1] > 0:
                _zero_unfilled_slots(st)
                ws, fdesc, xcs, dycs, total = st[0], st[3], st[4], st[5], st[6]
                sl = grad_slot(p)
                if sl is not None:
                    ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, xcs, dycs, out=sl.view, accumulate=sl.filled)
                    sl.filled = True
                else:
                    dw = ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, xcs, dycs)
                    out[i] = dw if out[i] is None else out[i] + dw
            p._t2v_wg_state, p._t2v_wg_images = None, 0
            continue
        ws, done, _, fdesc, xcs, dycs = st
        total = p._t2v_wg_images
        if done > 0:
            tp = ops.winograd_tile_rows(fdesc)      # slot pitch of the batch-wide tile lists
            nv = 36 * total * tp * xcs
            ws[:nv].view(36, total, tp * xcs)[:, done:].zero_()
            # (their A dy A^T slots too: 0 x whatever the allocation held is not 0 for a NaN / Inf bit pattern)
            ws[nv:nv + 36 * total * tp * fdesc.Cout].view(36, total, tp * fdesc.Cout)[:, done:].zero_()
            sl = grad_slot(p)
            if sl is not None:
                ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, xcs, dycs, out=sl.view, accumulate=sl.filled)
                sl.filled = True
            else:
                dw = ops.conv2d_backward_weight_winograd_reduce(fdesc, ws, total, xcs, dycs)
                out[i] = dw if out[i] is None else out[i] + dw
        p._t2v_wg_state, p._t2v_wg_images = None, 0
    return out


BN_MOMENTUM = 0.1     # nn.BatchNorm2d default ($SP/torch/nn/modules/batchnorm.py:16)
_BN_UPDATES = [1]     # running-statistics updates per forward (a forward that stands for two upstream forwards: 2)
_NO_PARAM_GRAD = set()   # ids of parameters whose gradients the backward pass under way must not produce
_NO_INPUT_DX = [False]   # the backward pass under way wants parameter gradients only: input layers skip their data gradient


@contextlib.contextmanager
def input_gradients_off():
    """Scope of a backward pass that asks for PARAMETER gradients of a loss network only (D's loss): the network's input
    layers (conv_block(..., need_dx=2)) skip the data-gradient conv into their input -- with one shared forward on the
    attached fake frames that input requires grad, but only G's backward pass (param_gradients_off) ever reads it."""
    old = _NO_INPUT_DX[0]
    _NO_INPUT_DX[0] = True
    try:
        yield
    finally:
        _NO_INPUT_DX[0] = old


@contextlib.contextmanager
def param_gradients_off(params):
    """Scope of a backward pass that runs THROUGH layers whose parameters belong to another loss: their nodes only
    propagate the data gradient (no weight-gradient kernels, nothing delivered to their gradient slots)."""
    ids = {id(getattr(p, "_t2v_owner", p)) for p in params}
    _NO_PARAM_GRAD.update(ids)
    try:
        yield
    finally:
        _NO_PARAM_GRAD.difference_update(ids)


@contextlib.contextmanager
def bn_updates(n):
    old = _BN_UPDATES[0]
    _BN_UPDATES[0] = n
    try:
        yield
    finally:
        _BN_UPDATES[0] = old


def running_stats(gamma, create=True):
    """[running_mean, running_var, num_batches_tracked] of the BatchNorm2d whose weight is `gamma` (kept on the parameter
    object; created at torch's initial values 0 / 1 / 0)."""
    owner = getattr(gamma, "_t2v_owner", gamma)
    rs = getattr(owner, "_t2v_running", None)
    if rs is None and create:
        rs = [torch.zeros(owner.numel(), dtype=torch.float32, device=owner.device),
              torch.ones(owner.numel(), dtype=torch.float32, device=owner.device), 0]
        owner._t2v_running = rs
    return rs


def running_update_args(gamma, n, times=None):
    """Every training-mode forward of a BatchNorm2d also moves its running statistics
    ($SP/torch/nn/modules/batchnorm.py:57-64, momentum 0.1, unbiased variance) -- never read on this path (upstream keeps
    the generator in train mode at test time) but part of a faithful checkpoint.  Returns (running_mean,
    running_var, momentum, times) for the finalize call, which moves them in its own launch, or None (no affine norm,
    fewer than two values per channel); counts the updates on the statistics' step counter."""
    if gamma is None or n < 2:
        return None
    times = _BN_UPDATES[0] if times is None else int(times)
    rs = running_stats(gamma)
    rs[2] += times
    return (rs[0], rs[1], BN_MOMENTUM, times)


_ZEROS = {}


def _zeros(n, device):
    """shared read-only zero vector (the exactly-zero bias gradients in front of norm layers)"""
    z = _ZEROS.get((n, device))
    if z is None:
        z = _ZEROS[(n, device)] = torch.zeros(n, dtype=torch.float32, device=device)
        if z.is_cuda:
            torch.cuda.current_stream().synchronize()      # (once per size: the vector is read from more than one stream)
    return z


def _desc_key(d, xcs):
    return (d.H, d.W, d.Cin, d.Cout, d.kH, d.stride, d.pad, d.pad_mode, d.transposed, d.output_padding, d.algo, xcs)


class _ConvBlock(torch.autograd.Function):
    """y = act(norm(conv(x) + b)) + res  on a batch [B,H,W,cs].

    norm: None | 'instance' (statistics per image) | 'batch' (statistics over the batch, BatchNorm2d in
    train mode); relu: 0 none / 1 ReLU / 2 LeakyReLU(0.2) applied after the norm; without a norm the
    activation `act` (ACT_NONE / ACT_TANH / ACT_LRELU) is fused into the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, res, desc, norm, relu, act, need_dx, slope=0.2, groups=1, bn_times=None, pt_skip=0):
        # groups > 1 (norm 'batch'): the batch is `groups` independent passes of B / groups images each -- the discriminators'
        # real / fake / raw passes in ONE node: one conv launch, one data-gradient launch, one weight-gradient launch over
        # all of them; the BatchNorm statistics (and their running averages, moved bn_times[g] times) stay per pass.
        # pt_skip: leading images nobody differentiates in a pass-through backward (param_gradients_off: the generator's
        # loss reaches the fake / raw passes only) -- that backward runs on the images [pt_skip:] alone.
        B = x.shape[0]
        assert B % groups == 0, "conv_block: batch %d is not %d equal passes" % (B, groups)
        dev = x.device
        xcs = x.shape[-1]
        ho, wo = ops.conv_out_dims(desc)
        ycs = ops.round_up(desc.Cout, 4)
        fdesc = ops.conv_desc(desc.H, desc.W, desc.Cin, desc.Cout, desc.kH, desc.stride, desc.pad, desc.pad_mode,
                              bool(desc.transposed), act if norm is None else ops.ACT_NONE, slope, desc.output_padding)
        # 3x3 stride-1 layers (the ResnetBlock convs) run as Winograd where that is the smaller GEMM; the weight
        # gradient keeps the direct layout (ddesc)
        ddesc = fdesc
        # (convs with a fused activation -- the VGG19 loss network -- stay on the direct kernel: F(4x4)'s 5x larger rounding
        # noise flips visibly more ReLU / max-pool decisions in that 13-layer net -- input-gradient relative L2 error 3.7e-3
        # instead of 2.3e-3 -- for 2 ms of a 99 ms step)
        if fdesc.act == ops.ACT_NONE and ycs == desc.Cout and int(os.environ.get("T2V_CONV_ALGO", "0")) != 1:
            algo = ops.best_conv_algo(fdesc, xcs, int(os.environ.get("T2V_CONV_ALGO", "0")))
            if algo != ops.ALGO_DIRECT:
                fdesc = ops.with_algo(fdesc, algo)
            elif int(os.environ.get("T2V_CONV_ALGO", "0")) == 0 and desc.stride == 2 and ops.polyphase_pays(fdesc, xcs):
                # the deep stride-2 / transposed layers as polyphase Winograd F(4,2), as in the inference frames
                # (csrc/polyphase.hip); the weight gradient keeps the direct layout (ddesc), the data gradient is the
                # polyphase form of the adjoint geometry (backward.ConvDataGrad)
                fdesc = ops.with_algo(fdesc, ops.ALGO_POLYPHASE)
        pw = cached_pack(w, ("fwd",) + _desc_key(fdesc, xcs), lambda: ops.pack_conv_weight(w.detach().contiguous(), fdesc, xcs))
        c = torch.empty(B, ho, wo, ycs, dtype=torch.float32, device=dev)
        mrs = None
        # weight gradient in the Winograd domain where the forward took F(4x4,3x3) (a quarter of the FLOPs)
        wino_wgrad = fdesc.algo == ops.ALGO_WINOGRAD_F4 and ops.backward_weight_winograd_supported(ddesc, xcs, ycs) \
            and os.environ.get("T2V_WGRAD_WINOGRAD", "1") != "0"
        # ... whose workspace, from the layer's second step on, takes this conv's input transform right now (_keep_v_slot)
        kept = _keep_v_slot(w, x, ddesc, xcs, ycs) if (wino_wgrad and ctx.needs_input_grad[1] and _WG_BATCH[0]) else None
        # (a batch of a direct-algorithm layer -- the discriminators' -- is ONE launch: ops.conv2d_auto_batch)
        if norm is None:
            if kept is not None:
                ops.conv2d_winograd(x[0], pw, b.detach(), fdesc, out=c[0], keep_v=kept)
            else:
                ops.conv2d_auto_batch(x, pw, b.detach(), fdesc, y_cs=ycs, out=c)
            y = c
        else:
            n = ops.conv_stats_buffer(fdesc, dev).numel()
            stats = torch.empty(B * n, dtype=torch.float32, device=dev)
            if kept is not None:
                ops.conv2d_winograd(x[0], pw, b.detach(), fdesc, stats=stats, out=c[0], keep_v=kept)
            else:
                ops.conv2d_auto_batch(x, pw, b.detach(), fdesc, y_cs=ycs, stats=stats, out=c)
            y = torch.empty_like(c)
            g = gamma.detach() if gamma is not None else None
            bt = beta.detach() if beta is not None else None
            # the residual add rides in the norm-apply pass (its gradient is the identity)
            rs = res.detach().contiguous() if (res is not None and res.shape == y.shape) else None
            # (the finalize launch also moves BatchNorm2d's running statistics: running_update_args)
            if norm == "batch" and groups > 1:
                gB, mrs = B // groups, []
                for gi in range(groups):
                    sl = slice(gi * gB, (gi + 1) * gB)
                    mrs.append(ops.batch_norm_finalize(stats[gi * gB * n:(gi + 1) * gB * n], fdesc, gB,
                                                       running=running_update_args(gamma, gB * ho * wo,
                                                                                   bn_times[gi] if bn_times else None)))
                    ops.instance_norm_apply(c[sl], mrs[gi], g, bt, res1=rs[sl] if rs is not None else None, relu=relu, out=y[sl])
            elif norm == "batch":
                mrs = [ops.batch_norm_finalize(stats, fdesc, B, running=running_update_args(gamma, B * ho * wo))]
                ops.instance_norm_apply(c, mrs[0], g, bt, res1=rs, relu=relu, out=y)
            else:
                mrs = []
                for i in range(B):       # BatchNorm2d(train) on a batch of one
                    mrs.append(ops.instance_norm_finalize(stats[i * n:(i + 1) * n], fdesc,
                                                          running=running_update_args(gamma, ho * wo)))
                    ops.instance_norm_apply(c[i], mrs[i], g, bt, res1=rs[i] if rs is not None else None, relu=relu, out=y[i])
            if rs is not None:
                res = None
        if res is not None:
            y = y + res   # residual add (plumbing-level elementwise; its gradient is the identity)
        # every use of the layer in this graph (one per frame of the clip) is counted on the weight: their Winograd-
        # domain gradients are reduced together by the last backward node to run (T2V_WGRAD_BATCH=0: one by one)
        if wino_wgrad and w.requires_grad and _WG_BATCH[0]:
            w._t2v_wg_images = getattr(w, "_t2v_wg_images", 0) + B
            w._t2v_wg_seen = getattr(w, "_t2v_wg_seen", 0) + B
            wino_wgrad = 2
        elif not wino_wgrad and w.requires_grad and _DW_PAIR[0] and \
                (B > 1 or ops.backward_weight_strided_supported(ddesc, xcs, ycs)):
            w._t2v_dw_uses = getattr(w, "_t2v_dw_uses", 0) + 1      # (two or more uses: _paired_direct_wgrad)
        for prm in (w, b, gamma, beta):
            expect_gradient(prm)
        # (an input nobody differentiates -- the discriminators' real pass -- needs no data-gradient conv)
        need_dx = int(need_dx) if (need_dx and ctx.needs_input_grad[0]) else 0      # 2: an input layer (input_gradients_off)
        ctx.meta = (desc, ddesc, norm, relu, act, need_dx, mrs, gamma is not None, wino_wgrad, slope)
        ctx.groups, ctx.pt_skip = groups, int(pt_skip)
        ctx.kept = kept
        ctx.save_for_backward(x, w, c, gamma, beta, y if (norm is None and act != ops.ACT_NONE) else None, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        desc, fdesc, norm, relu, act, need_dx, mrs, affine, wino_wgrad, slope = ctx.meta
        x, w, c, gamma, beta, y_act, b = ctx.saved_tensors
        dy = dy.contiguous()
        dgamma = dbeta = None
        if dy.is_cuda:
            side_gemm_hint()
        # a backward pass that only passes THROUGH this layer (param_gradients_off): data gradient only
        want = [bool(v) for v in ctx.needs_input_grad]
        if need_dx == 2 and _NO_INPUT_DX[0]:
            need_dx = 0
        groups, g0, img0, x_full = ctx.groups, 0, 0, x
        gB = x.shape[0] // groups
        if _NO_PARAM_GRAD and id(getattr(w, "_t2v_owner", w)) in _NO_PARAM_GRAD:
            want[1] = want[2] = want[3] = want[4] = False
            if ctx.pt_skip:
                # ... and only through the passes the loss reaches: the leading `pt_skip` images (the real pass) carry no
                # gradient in this backward -- their rows of dy are never read, their rows of dx are exact zeros
                img0 = ctx.pt_skip
                assert not wino_wgrad and 0 < img0 < x.shape[0] and (groups == 1 or img0 % gB == 0)
                g0 = img0 // gB if groups > 1 else 0
                x, c, dy = x[img0:], c[img0:], dy[img0:]
                y_act = y_act[img0:] if y_act is not None else None
        B = x.shape[0]
        # parameters with a bucket slot get their gradients delivered in place (and None goes back to autograd)
        sl_w = grad_slot(w) if want[1] else None
        sl_b = grad_slot(b) if want[2] else None
        sl_g = grad_slot(gamma) if (affine and want[3]) else None
        sl_bt = grad_slot(beta) if (affine and want[4]) else None
        both = sl_g is not None and sl_bt is not None

        def into_slots():           # both affine gradients have bucket slots: the norm backward's final pass fills them
            if not (both and affine and (want[3] or want[4])):
                return None
            first = not sl_g.filled
            sl_g.filled = sl_bt.filled = True
            return (sl_bt.view, sl_g.view, first)

        def affine_sums(sums):       # [C,2] = (sum g, sum g*xhat) -> d beta, d gamma
            nonlocal dbeta, dgamma
            if both:
                return               # (delivered by the kernel: into_slots)
            else:
                db_, dg_ = sums.t().contiguous().unbind(0)
                dbeta = db_ if dbeta is None else dbeta + db_
                dgamma = dg_ if dgamma is None else dgamma + dg_

        # Where this node's gradient in front of the norm would be read by nothing but the transform A dy A^T (the weight
        # gradient's workspace holds V already, the data gradient takes A dy A^T from there), it is not written at all: the
        # norm backward delivers its two sums, the transform forms the gradient per loaded element (T2V_DY_NORM_FUSED=0: off)
        dgrad_t = wino_wgrad == 2 and os.environ.get("T2V_DGRAD_TRANSPOSED", "1") != "0" and \
            ops.backward_data_winograd_supported(fdesc, x.shape[-1], c.shape[-1])
        lazy_dc = None
        if norm is not None and B == 1 and wino_wgrad == 2 and want[1] and ctx.kept is not None and (dgrad_t or not need_dx) \
                and os.environ.get("T2V_DY_NORM_FUSED", "1") != "0":
            _, sums = ops.instance_norm_backward(c[0], dy[0], mrs[0], gamma, beta, relu, affine_into=into_slots(), sums_only=True)
            if affine and (want[3] or want[4]):
                affine_sums(sums)
            lazy_dc = (c[0], dy[0], mrs[0], gamma, beta, relu, sums)
            dc = None
        elif norm is None:
            # (act_backward's mode 2 is a plain sigmoid; the fused flow / weight head is its mode 4)
            dc = ops.act_backward(dy, y_act, 4 if act == ops.ACT_FLOW_W else act, slope) if act != ops.ACT_NONE else dy
        elif norm == "batch" and groups > 1:
            dc = torch.empty_like(c)
            for gi in range(g0, groups):       # statistics per pass: the norm's adjoint per pass
                sl = slice((gi - g0) * gB, (gi - g0 + 1) * gB)
                _, s_i = ops.instance_norm_backward(c[sl], dy[sl], mrs[gi], gamma, beta, relu, out=dc[sl], affine_into=into_slots())
                if affine and (want[3] or want[4]):
                    affine_sums(s_i)
        elif norm == "batch":
            dc, sums = ops.instance_norm_backward(c, dy, mrs[0], gamma, beta, relu, affine_into=into_slots())
            if affine and (want[3] or want[4]):
                affine_sums(sums)
        else:
            dc = torch.empty_like(c)
            for i in range(B):
                _, s_i = ops.instance_norm_backward(c[i], dy[i], mrs[img0 + i], gamma, beta, relu, out=dc[i], affine_into=into_slots())
                if affine and (want[3] or want[4]):
                    affine_sums(s_i)
        if both:
            sl_g.owner.node_done(sl_g)
            sl_bt.owner.node_done(sl_bt)
        # a bias in front of a norm layer has an exactly zero gradient (the norm removes the channel mean)
        db = None
        if want[2]:
            if sl_b is not None:
                if norm is None:
                    with (wgrad_fork(dc) if wgrad_stream_on(dc) else contextlib.nullcontext()):   # (off the dy -> dx chain too)
                        if sl_b.filled:
                            _acc(sl_b.view, ops.channel_sum(dc, desc.Cout), False)
                        else:
                            ops.channel_sum(dc, desc.Cout, out=sl_b.view)
                    sl_b.filled = True
                    sl_b.owner.node_done(sl_b)      # (outside the scope: a bucket launch must see the MAIN stream as current)
                else:
                    deliver(sl_b, zero=True)
            else:
                db = ops.channel_sum(dc, desc.Cout) if norm is None else _zeros(desc.Cout, x.device)
        if not want[1]:      # frozen weights (the VGG19 feature extractor) / a pass-through backward: data gradient only
            dw = None
        elif wino_wgrad == 2:
            wg_info = []
            if ctx.kept is not None:
                dw = _kept_winograd_wgrad(w, dc, fdesc, ctx.kept, sl_w, wg_info, lazy_dc)
            else:
                dw = _batched_winograd_wgrad(w, x, dc, fdesc, sl_w, wg_info)
        elif wino_wgrad:
            if sl_w is not None:
                with (wgrad_fork(x, dc) if wgrad_stream_on(x) else contextlib.nullcontext()):
                    ops.conv2d_backward_weight_winograd(x, dc, fdesc, accumulate_into=sl_w.view if sl_w.filled else None,
                                                        out=sl_w.view)
                sl_w.filled = True
                sl_w.owner.node_done(sl_w)
                dw = None
            else:
                dw = ops.conv2d_backward_weight_winograd(x, dc, fdesc)
        else:
            paired, dw = _paired_direct_wgrad(w, x, dc, fdesc, sl_w)
            if paired:
                pass
            elif sl_w is not None:
                with (wgrad_fork(x, dc) if wgrad_stream_on(x) else contextlib.nullcontext()):
                    dwp = direct_weight_gradient(x, dc, fdesc)
                    ops.unpack_conv_weight_into(dwp, fdesc, x.shape[-1], sl_w.view, sl_w.filled)
                sl_w.filled = True
                sl_w.owner.node_done(sl_w)
                dw = None
            else:
                dw = ops.unpack_conv_weight(direct_weight_gradient(x, dc, fdesc), fdesc, x.shape[-1])
        dx = None
        xcs_ = x.shape[-1]
        if need_dx and wino_wgrad == 2 and want[1] and len(wg_info) == 3 and dgrad_t:
            # The weight gradient has just put A dy A^T of these images into its workspace: the data gradient by the
            # TRANSPOSED Winograd algorithm reads it from there -- U^T dM on the layer's own 256 tiles instead of the
            # full-correlation form's 289 -> 320, no second transform of dy, no flipped filter transform
            ws_, total_, done_ = wg_info
            # ... and where the fixed-grid GEMM has its [K][N] form for the shape, U^T is the forward pass's own packing read in
            # place (T2V_DGRAD_FORWARD_WEIGHTS=0: the transposed copy, same bits)
            fw = os.environ.get("T2V_DGRAD_FORWARD_WEIGHTS", "1") != "0" \
                and ops.backward_data_winograd_takes_forward_weights(fdesc, xcs_, c.shape[-1])
            if fw:
                # (wino_wgrad: the forward pass ran this layer as F(4x4) and holds that packing under this key)
                f4 = ops.with_algo(fdesc, ops.ALGO_WINOGRAD_F4)
                ut = cached_pack(w, ("fwd",) + _desc_key(f4, xcs_), lambda: ops.pack_conv_weight(w.detach().contiguous(), f4, xcs_))
            else:
                ut = cached_pack(w, ("dgradT",) + _desc_key(fdesc, xcs_), lambda: ops.pack_conv_weight_transposed(w.detach(), fdesc, xcs_))
            dx = torch.empty_like(x)
            for i in range(B):
                ops.conv2d_backward_data_winograd(fdesc, total_, done_ + i, ws_, xcs_, ut, out=dx[i], forward_weights=fw)
        elif need_dx:
            dg = ConvDataGrad(fdesc)
            dg.packed = cached_pack(w, ("dgrad",) + _desc_key(fdesc, x.shape[-1]), lambda: dg.refresh(w.detach()).packed)
            if ops.round_up(fdesc.Cin, 4) == x.shape[-1]:
                dx = torch.empty_like(x_full)
                if img0:
                    ops.zero_(dx[:img0])
                dg.batch(dc, dx[img0:])
            else:   # x carries more channel storage than the layer reads: zero gradient there
                dx = torch.zeros_like(x_full)
                dx[img0:, ..., :ops.round_up(fdesc.Cin, 4)] = torch.stack([dg(dc[i]) for i in range(B)])
        elif img0:
            raise RuntimeError("conv_block: pt_skip is for layers with a data gradient")
        return (dx, dw, db, dgamma, dbeta, (dy if ctx.needs_input_grad[5] else None)) + (None,) * 9


class _CatParams(torch.autograd.Function):
    """torch.cat([a, b], 0) of two parameters (the flow head and the weight head read the same features: one
    3-output conv).  Backward splits the gradient and delivers the halves into the parameters' bucket slots, like
    every other node that owns a parameter gradient."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.shape[0]
        ctx.save_for_backward(a, b)
        expect_gradient(a)
        expect_gradient(b)
        return torch.cat([a, b], 0)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga, gb = g[:ctx.n].contiguous(), g[ctx.n:].contiguous()
        out = []
        for p, gp, need in ((a, ga, ctx.needs_input_grad[0]), (b, gb, ctx.needs_input_grad[1])):
            sl = grad_slot(p) if need else None
            if sl is not None:
                deliver(sl, gp)
                out.append(None)
            else:
                out.append(gp if need else None)
        return tuple(out)


def conv_block(x, w, b, desc, gamma=None, beta=None, res=None, norm="instance", relu=1, act=ops.ACT_NONE,
               need_dx=True, slope=0.2, groups=1, bn_times=None, pt_skip=0):
    """slope: negative-side factor of act == ACT_LRELU (0.2 in the discriminators; 0 makes it a ReLU).
    groups / bn_times / pt_skip: several independent passes in one batch (_ConvBlock.forward)."""
    return _ConvBlock.apply(x, w, b, gamma, beta, res, desc, norm, relu, act, need_dx, slope, groups, bn_times, pt_skip)


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return torch.stack([ops.avgpool3x3s2(x[i]) for i in range(x.shape[0])])

    @staticmethod
    def backward(ctx, dy):
        H, W = ctx.hw
        dy = dy.contiguous()
        return torch.stack([ops.avgpool3x3s2_backward(dy[i], H, W) for i in range(dy.shape[0])])


class _MseConst(torch.autograd.Function):
    """mean((x - c)^2) over the LOGICAL channel 0 of a [.., 4]-padded logit tensor."""

    @staticmethod
    def forward(ctx, x, c):
        logits = x[..., 0].contiguous()
        ctx.c, ctx.shape = c, x.shape
        ctx.save_for_backward(logits)
        return ops.sum_sq_diff_const(logits, c)[0] / logits.numel()

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        d = ops.sum_sq_diff_const_backward(logits, ctx.c, 1.0 / logits.numel()) * g
        dx = torch.zeros(ctx.shape, dtype=torch.float32, device=logits.device)
        dx[..., 0] = d
        return dx, None


class _L1(torch.autograd.Function):
    """mean(|a - b|); gradient flows to `a` only (b is the detached real-branch feature)."""

    @staticmethod
    def forward(ctx, a, b, nlogical):
        ctx.save_for_backward(a, b)
        ctx.n = nlogical
        return ops.sum_abs_diff(a.contiguous(), b.contiguous())[0] / nlogical

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.sum_abs_diff_backward(a.contiguous(), b.contiguous(), 1.0 / ctx.n) * g, None, None


class _WarpComposite(torch.autograd.Function):
    """out = raw*w + resample(prev[..., c0:c0+3], flow)*(1-w) on [B,H,W,4] tensors; fw = (flow_x, flow_y, w, 0).
    Backward: t2v_flow_warp_composite_backward (SpatialGridSamplerBilinear_updateGradInput, THCUNN.h:1055, fused
    with the blend's adjoint); the gradient with respect to `prev` only when autograd asks for it (the generated
    previous frames are detached: max_frames_backpropagate 1)."""

    @staticmethod
    def forward(ctx, raw, fw, prev, c0):
        ctx.c0 = c0
        ctx.save_for_backward(raw, fw, prev)
        out = torch.empty_like(raw)
        for i in range(raw.shape[0]):
            ops.flow_warp_composite(raw[i], fw[i], prev[i], c0, out=out[i])
        return out

    @staticmethod
    def backward(ctx, dy):
        raw, fw, prev = ctx.saved_tensors
        dy = dy.contiguous()
        res = [ops.flow_warp_composite_backward(dy[i], None, raw[i], fw[i], prev[i], ctx.c0, ctx.needs_input_grad[2])
               for i in range(raw.shape[0])]
        d_raw = torch.stack([r[0] for r in res]) if len(res) > 1 else res[0][0].unsqueeze(0)
        d_fw = torch.stack([r[1] for r in res]) if len(res) > 1 else res[0][1].unsqueeze(0)
        d_prev = None
        if ctx.needs_input_grad[2]:
            d_prev = torch.stack([r[2] for r in res])
        return d_raw, d_fw, d_prev, None


class _Resample(torch.autograd.Function):
    """resample(img[..., c0:c0+3], flow) -> [B,H,W,4]: the warp losses' plain grid_sample (same kernels, no blend)."""

    @staticmethod
    def forward(ctx, fw, img, c0):
        ctx.c0 = c0
        ctx.save_for_backward(fw, img)
        out = torch.empty(fw.shape[:3] + (4,), dtype=torch.float32, device=fw.device)
        for i in range(fw.shape[0]):
            ops.flow_warp(fw[i], img[i], c0, out=out[i])
        return out

    @staticmethod
    def backward(ctx, dy):
        fw, img = ctx.saved_tensors
        dy = dy.contiguous()
        res = [ops.flow_warp_composite_backward(None, dy[i], None, fw[i], img[i], ctx.c0, ctx.needs_input_grad[1])
               for i in range(fw.shape[0])]
        d_fw = torch.stack([r[1] for r in res])
        d_img = torch.stack([r[2] for r in res]) if ctx.needs_input_grad[1] else None
        return d_fw, d_img, None


class _MaskedL1(torch.autograd.Function):
    """vid2vid's MaskedL1Loss [RECALL upstream models/networks.py]: L1Loss(a*mask, b*mask), mean over ALL B*C*H*W
    elements; a, b: [B,H,W,cs] (b None: zero target), mask: [B,H,W] of {0,1} or None, over the channels
    [c0, c0+C) (the flow / weight planes of the fused head are read in place).  Gradient to `a` only."""

    @staticmethod
    def forward(ctx, a, b, mask, c0, C):
        a = a.contiguous()
        ctx.cw = (c0, C)
        ctx.n = (a.numel() // a.shape[-1]) * C
        ctx.save_for_backward(a, b, mask)
        return ops.sum_abs_diff_masked(a, b, mask, c0, C)[0] / ctx.n

    @staticmethod
    def backward(ctx, g):
        a, b, mask = ctx.saved_tensors
        return ops.sum_abs_diff_masked_backward(a, b, mask, ctx.cw[0], ctx.cw[1], 1.0 / ctx.n) * g, None, None, None, None


def masked_l1(a, b, mask, C, c0=0):
    return _MaskedL1.apply(a, b.detach().contiguous() if b is not None else None,
                           mask.contiguous() if mask is not None else None, c0, C)


# ------------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------------
class TrainableGenerator(torch.nn.Module):
    """CompositeGenerator with or without its flow branch, upstream parameter names."""

    def __init__(self, spec, state_dict, device="cuda"):
        super().__init__()
        assert not spec.is_local, "train step: the global generator (n_scales_spatial 1)"
        self.spec = spec
        self.keys = layer_keys(spec)
        self.params = torch.nn.ParameterDict()
        for k, v in state_dict.items():
            self.params[k.replace(".", "/")] = torch.nn.Parameter(v.to(device, torch.float32).contiguous())

    def p(self, key):
        return self.params[key.replace(".", "/")]

    def named_upstream_parameters(self):
        return {k.replace("/", "."): v for k, v in self.params.items()}

    def forward(self, pose, prev, use_raw_only=False, full=False, need_flow=True):
        """pose [1,H,W,12], prev [1,H,W,8] NHWC -> fake [1,H,W,4] (RGB in channels 0..2).  full=True returns
        (fake, raw, flow_w): raw = the tanh image before the blend, flow_w [1,H,W,4] = (flow_x, flow_y in pixels,
        weight, 0); both None without a flow branch.  use_raw_only: the first frame of a sequence under
        --no_first_img (fake is raw; flow and weight are still computed, the flow losses see them)."""
        s = self.spec
        n, G = s.n_downsample, s.ngf
        H, W = pose.shape[1], pose.shape[2]
        norm = "instance"   # BatchNorm2d(train) with N=1 == instance norm + affine
        it = iter(self.keys)

        def cna(x, desc, relu=1, res=None, need_dx=True):
            ck, nk, kind = next(it)
            g = self.p(nk + ".weight") if s.norm == "batch" else None
            b = self.p(nk + ".bias") if s.norm == "batch" else None
            return conv_block(x, self.p(ck + ".weight"), self.p(ck + ".bias"), desc, g, b, res, norm, relu,
                              need_dx=need_dx)

        def encoder(x, in_nc, nb):
            h = cna(x, ops.conv_desc(H, W, in_nc, G, 7, 1, 3, ops.PAD_REFLECT), need_dx=False)
            for i in range(n):
                h = cna(h, ops.conv_desc(H >> i, W >> i, G << i, G << (i + 1), 3, 2, 1, ops.PAD_ZERO))
            return resblocks(h, nb)

        def resblocks(h, nb):
            C, hb, wb = G << n, H >> n, W >> n
            d = ops.conv_desc(hb, wb, C, C, 3, 1, 1, ops.PAD_REFLECT)
            for _ in range(nb):
                t = cna(h, d, relu=1)
                h = cna(t, d, relu=0, res=h)
            return h

        def decoder(h):
            for i in range(n):
                l = n - i
                h = cna(h, ops.conv_desc(H >> l, W >> l, G << l, G << (l - 1), 3, 2, 1, ops.PAD_ZERO, True))
            return h

        nb_enc, nb_res = s.n_blocks - s.n_blocks // 2, s.n_blocks // 2
        # (the independent halves -- the two encoders, the image and flow branches -- on a second stream, as the inference
        # path runs them, were tried in round 3: +0.7 % with the fixed-grid kernels on; removed)
        d = encoder(pose, s.input_nc, nb_enc) + encoder(prev, s.prev_nc, nb_enc)
        img_feat = decoder(resblocks(d, nb_res))
        ck, _, _ = next(it)
        raw = conv_block(img_feat, self.p(ck + ".weight"), self.p(ck + ".bias"),
                         ops.conv_desc(H, W, G, 3, 7, 1, 3, ops.PAD_REFLECT), norm=None, relu=0, act=ops.ACT_TANH)
        if s.no_flow:
            return (raw, None, None) if full else raw
        if use_raw_only and not need_flow:
            # raw-only first frame and no loss reads its flow / weight maps: the flow branch would be a dead part of
            # the graph (its batched weight-gradient slots would never be reduced) -- do not run it
            return (raw, raw, None) if full else raw
        flow_feat = decoder(resblocks(d, nb_res))
        (kf, kw), _, _ = next(it)
        # model_final_flow (2 outputs, x20) and model_final_w (1 output, sigmoid) read the same features: one
        # 3-output conv, as in the inference path; autograd's cat splits the gradient back onto the two parameters
        w3 = _CatParams.apply(self.p(kf + ".weight"), self.p(kw + ".weight"))
        b3 = _CatParams.apply(self.p(kf + ".bias"), self.p(kw + ".bias"))
        fw = conv_block(flow_feat, w3, b3, ops.conv_desc(H, W, G, 3, 7, 1, 3, ops.PAD_REFLECT), norm=None, relu=0,
                        act=ops.ACT_FLOW_W, slope=20.0 * (2 ** s.scale))
        fake = raw if use_raw_only else _WarpComposite.apply(raw, fw, prev, s.prev_nc - 3)
        return (fake, raw, fw) if full else fake


class TrainableDiscriminator(torch.nn.Module):
    """MultiscaleDiscriminator (num_D PatchGANs, getIntermFeat), upstream parameter names
    `scale{i}_layer{j}.{0,1}.*`; BatchNorm statistics over the batch."""

    def __init__(self, input_nc, state_dict, ndf=64, n_layers=3, num_D=2, norm="batch", device="cuda"):
        super().__init__()
        self.input_nc, self.n_layers, self.num_D, self.norm = input_nc, n_layers, num_D, norm
        self.params = torch.nn.ParameterDict()
        for k, v in state_dict.items():
            if "running" in k or "num_batches" in k:
                continue
            self.params[k.replace(".", "/")] = torch.nn.Parameter(v.to(device, torch.float32).contiguous())
        self.ndfs = [min(ndf * 2 ** (num_D - 1 - i), 64) for i in range(num_D)]
        self._frozen = False

    def p(self, key):
        t = self.params[key.replace(".", "/")]
        if not self._frozen:
            return t
        d = t.detach()
        d._t2v_owner = t
        return d

    def named_upstream_parameters(self):
        return {k.replace("/", "."): v for k, v in self.params.items()}

    def _single(self, x, i, groups=1, bn_times=None, pt_skip=0):
        ndf = self.ndfs[i]
        chans = [(self.input_nc, ndf, 2, False)]
        nf = ndf
        for _ in range(1, self.n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            chans.append((nf_prev, nf, 2, True))
        nf_prev, nf = nf, min(nf * 2, 512)
        chans += [(nf_prev, nf, 1, True), (nf, 1, 1, False)]
        feats, cur = [], x
        for j, (cin, cout, stride, has_norm) in enumerate(chans):
            pre = "scale%d_layer%d" % (i, j)
            desc = ops.conv_desc(cur.shape[1], cur.shape[2], cin, cout, 4, stride, 2, ops.PAD_ZERO)
            last = j == len(chans) - 1
            if has_norm:
                g = self.p(pre + ".1.weight") if self.norm == "batch" else None
                b = self.p(pre + ".1.bias") if self.norm == "batch" else None
                cur = conv_block(cur, self.p(pre + ".0.weight"), self.p(pre + ".0.bias"), desc, g, b, None, self.norm, 2,
                                 groups=groups, bn_times=bn_times, pt_skip=pt_skip)
            else:
                cur = conv_block(cur, self.p(pre + ".0.weight"), self.p(pre + ".0.bias"), desc, norm=None, relu=0,
                                 act=ops.ACT_NONE if last else ops.ACT_LRELU, need_dx=2 if j == 0 else True,
                                 groups=groups, pt_skip=pt_skip)
            feats.append(cur)
        return feats

    def forward(self, x, frozen=False, groups=1, bn_times=None, pt_skip=0):
        """x [B,H,W,cs] -> result[i] = stage outputs of the i-th finest scale.  frozen=True: the parameters enter
        detached (the generator-side passes: only the data gradient is wanted, so the backward nodes skip the
        weight-gradient kernels; the packed-weight cache is keyed on the parameter either way).
        groups > 1: x holds that many independent passes of B / groups images (real | fake | raw): every layer is ONE
        launch over all of them, the BatchNorm statistics stay per pass and move their running averages bn_times[g]
        times; pt_skip: leading images a pass-through backward (the generator's loss) skips (_ConvBlock)."""
        self._frozen = frozen
        try:
            result = []
            for i in range(self.num_D):
                result.append(self._single(x, self.num_D - 1 - i, groups, bn_times, pt_skip))
                if i != self.num_D - 1:
                    x = _AvgPool.apply(x)
            return result
        finally:
            self._frozen = False


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_backward(x, dy.contiguous())


# torchvision vgg19 `features` ($SP/torchvision/models/vgg.py:82, cfg 'E'): conv indices and their widths up to
# relu5_1; 'M' = MaxPool2d(2,2).  The five taps are relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (features[1], [6],
# [11], [20], [29]) [RECALL upstream models/networks.py Vgg19 slices].
VGG19_CFG = [(0, 3, 64, True), (2, 64, 64, False), "M", (5, 64, 128, True), (7, 128, 128, False), "M",
             (10, 128, 256, True), (12, 256, 256, False), (14, 256, 256, False), (16, 256, 256, False), "M",
             (19, 256, 512, True), (21, 512, 512, False), (23, 512, 512, False), (25, 512, 512, False), "M",
             (28, 512, 512, True)]
VGG_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def vgg19_random_state_dict(seed=0):
    """He-initialised stand-in for torchvision's vgg19 weights (which are not in the reference tree): the same keys
    and shapes, for tests and timing."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = {}
    for item in VGG19_CFG:
        if item == "M":
            continue
        idx, cin, cout, _ = item
        sd["features.%d.weight" % idx] = torch.from_numpy(
            (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32))
        sd["features.%d.bias" % idx] = torch.from_numpy((rng.standard_normal(cout) * 0.05).astype(np.float32))
    return sd


class HipVGG19Features(torch.nn.Module):
    """Frozen VGG19 feature extractor of the perceptual loss: 3x3 zero-padded convs with the
    ReLU fused in the conv epilogue, 2x2 max-pools; forward returns the five relu*_1 maps.  Weights: a torchvision
    `vgg19` state dict (keys features.N.weight / .bias)."""

    def __init__(self, state_dict, device="cuda"):
        super().__init__()
        self.w = {}
        for item in VGG19_CFG:
            if item == "M":
                continue
            idx = item[0]
            self.w[idx] = (state_dict["features.%d.weight" % idx].to(device, torch.float32).contiguous(),
                           state_dict["features.%d.bias" % idx].to(device, torch.float32).contiguous())

    def forward(self, x):
        """x [B,H,W,cs>=3] in [-1,1] (no ImageNet normalisation, as upstream) -> [relu1_1, ..., relu5_1]."""
        taps, cur = [], x
        for item in VGG19_CFG:
            if item == "M":
                cur = _MaxPool.apply(cur)
                continue
            idx, cin, cout, tap = item
            desc = ops.conv_desc(cur.shape[1], cur.shape[2], cin, cout, 3, 1, 1, ops.PAD_ZERO)
            w, b = self.w[idx]
            cur = conv_block(cur, w, b, desc, norm=None, relu=0, act=ops.ACT_LRELU, slope=0.0)
            if tap:
                taps.append(cur)
        return taps


def vgg_loss(vgg, fake, real, real_feats=None):
    """VGGLoss [RECALL upstream models/networks.py]: sum_i w_i * L1(vgg(fake)_i, vgg(real)_i.detach()),
    w = 1/32, 1/16, 1/8, 1/4, 1.  Upstream first halves inputs wider than 1024 px with AvgPool2d(2, 2); that only
    happens in multi-scale (n_scales_spatial > 1) training, which this train step does not cover."""
    if fake.shape[2] > 1024:
        raise NotImplementedError("vgg_loss: inputs wider than 1024 px (upstream's AvgPool2d(2,2) pre-scaling)")
    fr = real_feats          # the real frames' taps, when the caller already has them (two fake inputs, one real)
    if fr is None:
        with torch.no_grad():
            fr = vgg(real)
    ff = vgg(fake)
    return sum(wi * _L1.apply(a, b, a.numel()) for wi, a, b in zip(VGG_WEIGHTS, ff, fr))


def gan_loss(pred_scales, target_is_real):
    return sum(_MseConst.apply(st[-1], 1.0 if target_is_real else 0.0) for st in pred_scales)


def feature_matching_loss(pred_fake, pred_real, n_layers=3, lambda_feat=10.0):
    num_D = len(pred_fake)
    total = 0.0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            f, r = pred_fake[i][j], pred_real[i][j].detach()
            total = total + _L1.apply(f, r, f.numel()) * ((1.0 / num_D) * (4.0 / (n_layers + 1)) * lambda_feat)
    return total


class LossBook:
    """The scalar loss terms of one train step, evaluated together (ops.loss_terms: one launch for the values AND the
    gradient seeds of all terms, one per-term final pass).

    Every term's weight is a host number -- lambda_feat, the 1/2 of the discriminator loss, face_weight, 1 / n of the
    mean -- so the gradient of the total with respect to a term's operand is known without any device scalar: the term
    kernel writes it (the `seed`) in the same pass that reduces the value, and the backward passes start from those seeds
    (torch.autograd.grad(outputs=<operands>, grad_outputs=<seeds>)) instead of from a scalar graph of one-element
    multiplies and adds.  Values come back in one vector: `read()` sums them by name on the host."""

    CHUNK = 1 << 14

    def __init__(self, device):
        self.dev = torch.device(device)
        self.terms = []
        self.out = None
        self.zero_names = []      # named terms that are switched off in this step: reported as 0.0
        self._host = self._devb = self._part = None
        self._pending = None

    def reset(self):
        self.terms, self.out, self.zero_names = [], None, []

    def mse(self, name, x, target, weight, seed=None):
        """weight * mean((x[..., 0] - target)^2): x a contiguous [.., cs] block of logit rows (LSGAN; THCUNN.h:356); seed (same
        block of a gradient tensor) receives weight * 2 (x - target) / n in channel 0 and zeros in the pad channels"""
        cs = x.shape[-1]
        n = x.numel() // cs
        assert x.is_contiguous() and (seed is None or (seed.is_contiguous() and seed.shape == x.shape))
        self.terms.append((name, 0, x, None, seed, n, cs, float(target), weight / n, weight / n))

    def l1(self, name, a, b, weight, seed=None):
        """weight * mean(|a - b|) (feature matching; THCUNN.h:18), gradient with respect to `a` only"""
        n = a.numel()
        assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
        assert seed is None or (seed.is_contiguous() and seed.shape == a.shape)
        self.terms.append((name, 1, a, b, seed, n, 1, 0.0, weight / n, weight / n))

    def zero(self, seed):
        """rows of a gradient tensor no loss term reaches in that backward pass: exact zeros"""
        assert seed.is_contiguous()
        self.terms.append((None, 2, seed, None, seed, seed.numel(), 1, 0.0, 0.0, 0.0))

    def run(self):
        """enqueue the launch for the terms collected so far"""
        import numpy as np
        T = len(self.terms)
        if T == 0:
            return
        ch = self.CHUNK
        counts = [max(1, -(-t[5] // ch)) for t in self.terms]
        NC = sum(counts)
        # one staging buffer, 8-byte aligned segments: term_ptrs | chunk_off | term_ints | chunk_term | term_chunk0 | term_floats
        segs, o = {}, 0
        for key, nbytes in (("tp", 32 * T), ("co", 8 * NC), ("ti", 8 * T), ("ct", 4 * NC), ("t0", 4 * (T + 1)), ("tf", 12 * T)):
            segs[key] = (o, nbytes)
            o += -(-nbytes // 8) * 8
        if self._host is None or self._host.numel() < o:
            cap = max(o, 1 << 16)
            self._host = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self._devb = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        if self._part is None or self._part.numel() < NC + T:
            self._part = torch.empty(max(NC + T, 1 << 14), dtype=torch.float32, device=self.dev)
        if self._pending is not None:
            self._pending.synchronize()      # the previous launch's table copy has left the pinned staging buffer
        hv = self._host.numpy()

        def view(key, dt):
            a, nb = segs[key]
            return hv[a:a + nb].view(dt)
        tp, co, ti, ct, t0, tf = view("tp", np.int64), view("co", np.int64), view("ti", np.int32), view("ct", np.int32), \
            view("t0", np.int32), view("tf", np.float32)
        c0 = 0
        for t, (name, op, a, b, seed, n, cs, c, ss, vs) in enumerate(self.terms):
            tp[4 * t:4 * t + 4] = (a.data_ptr(), b.data_ptr() if b is not None else 0, seed.data_ptr() if seed is not None else 0, n)
            ti[2 * t:2 * t + 2] = (op, cs)
            tf[3 * t:3 * t + 3] = (c, ss, vs)
            t0[t] = c0
            k = counts[t]
            ct[c0:c0 + k] = t
            co[c0:c0 + k] = np.arange(k, dtype=np.int64) * ch
            c0 += k
        t0[T] = c0
        self._devb[:o].copy_(self._host[:o], non_blocking=True)

        def dev(key):
            a, nb = segs[key]
            return self._devb[a:a + nb]
        self.out = self._part[NC:NC + T]
        ops.loss_terms(dev("tp"), dev("ti"), dev("tf"), dev("ct"), dev("co"), dev("t0"), T, NC, ch, self._part, self.out)
        self._pending = torch.cuda.current_stream().record_event()
        # the launch is enqueued: let go of the operands (views into the step's autograd graph -- held here they would keep
        # the whole graph alive into the next step's forward pass); the names stay for read()
        self.terms = [(t[0],) for t in self.terms]

    def value_tensors(self):
        """{name: [indices into self.out]} of the named terms"""
        idx = {}
        for t, term in enumerate(self.terms):
            if term[0] is not None:
                idx.setdefault(term[0], []).append(t)
        return idx

    def read(self, extra=None):
        """one host read for the step: {name: value} of the named terms (+ the one-element device tensors in `extra`)"""
        names = self.value_tensors()
        keys = list((extra or {}).keys())
        parts = ([self.out] if self.out is not None else []) + [extra[k].detach().reshape(1).float() for k in keys]
        if not parts:
            return {}
        host = torch.cat(parts).tolist()
        T = len(self.terms) if self.out is not None else 0
        res = {k: sum(host[i] for i in ix) for k, ix in names.items()}
        for j, k in enumerate(keys):
            res[k] = host[T + j]
        return res


# ------------------------------------------------------------------------------------------------
# optimiser + data-parallel gradient exchange
# ------------------------------------------------------------------------------------------------
class FusedAdam:
    """torch-0.4.1 Adam semantics ($SP/torch/optim/adam.py:48-98), one fused HIP kernel per tensor.  As there,
    `state['step']` is per parameter and only advances when that parameter has a gradient (adam.py:58-60, 82): a
    discriminator that sits out the first iterations (temporal windows not yet full, no face box) starts its bias
    correction at 1 when it first trains."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.steps = [0] * len(self.params)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    CHUNK = 1 << 16

    def _tables(self):
        """static device tables of the multi-tensor launch (parameter / moment addresses, lengths, chunk map)"""
        if getattr(self, "_tab", None) is None:
            dev = self.params[0].device
            import math
            ct, co = [], []
            for t, p in enumerate(self.params):
                for c in range(math.ceil(p.numel() / self.CHUNK)):
                    ct.append(t)
                    co.append(c * self.CHUNK)
            self._tab = {
                "nelem": torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev),
                "chunk_tensor": torch.tensor(ct, dtype=torch.int32, device=dev),
                "chunk_off": torch.tensor(co, dtype=torch.int64, device=dev),
                "ptrs_host": torch.zeros(len(self.params), 4, dtype=torch.int64).pin_memory(),
                "ss_host": torch.zeros(len(self.params), dtype=torch.float32).pin_memory(),
                "ptrs": torch.zeros(len(self.params), 4, dtype=torch.int64, device=dev),
                "ss": torch.zeros(len(self.params), dtype=torch.float32, device=dev),
            }
        return self._tab

    def step(self):
        """One launch for all tensors (T2V_ADAM_MULTI=0: one launch per tensor).  Every parameter keeps its own step
        count; a parameter without gradient is skipped, moments included."""
        import math
        if os.environ.get("T2V_ADAM_MULTI", "1") == "0" or not self.params:
            for i, (p, m, v) in enumerate(zip(self.params, self.m, self.v)):
                if p.grad is not None:
                    self.steps[i] += 1
                    ops.adam_step(p.data, p.grad.contiguous(), m, v, self.lr, self.betas[0], self.betas[1], self.eps,
                                  self.steps[i])
                    invalidate_packs(p)   # written through the raw pointer: no tensor version bump
            prefetch_packs(self.params)
            return
        tab = self._tables()
        if getattr(self, "_pending", None) is not None:
            self._pending[1].synchronize()    # the previous step's table copies have left the pinned staging buffers
        grads = []        # kept alive until the launch is enqueued
        ph, sh = tab["ptrs_host"], tab["ss_host"]
        b1, b2 = self.betas
        for i, (p, m, v) in enumerate(zip(self.params, self.m, self.v)):
            g = p.grad
            if g is None:
                ph[i, 1] = 0
                continue
            g = g.contiguous()
            grads.append(g)
            self.steps[i] += 1
            k = self.steps[i]
            ph[i, 0], ph[i, 1], ph[i, 2], ph[i, 3] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            sh[i] = self.lr * math.sqrt(1.0 - b2 ** k) / (1.0 - b1 ** k)
            invalidate_packs(p)
        tab["ptrs"].copy_(ph, non_blocking=True)
        tab["ss"].copy_(sh, non_blocking=True)
        ops.adam_step_multi(tab["ptrs"], tab["nelem"], tab["ss"], tab["chunk_tensor"], tab["chunk_off"], self.CHUNK, b1, b2,
                            self.eps)
        # the pinned staging tables are reused by the next step: make sure this step's copies have been issued from them
        self._pending = (grads, torch.cuda.current_stream().record_event())
        prefetch_packs(self.params)


def _acc(dst, src, overwrite):
    """dst (+)= src on flat views: HIP kernel on device tensors; host tensors only occur in the gloo CPU tests of the
    bucket bookkeeping (the trainer itself cannot exist without the GPU)"""
    if dst.is_cuda:
        ops.accumulate_(dst, src.contiguous().view(-1), overwrite=overwrite)
    elif overwrite:
        dst.view(-1).copy_(src.reshape(-1))
    else:
        dst.view(-1).add_(src.reshape(-1))


def _zero(x):
    ops.zero_(x) if x.is_cuda else x.zero_()


def _scale(x, f):
    ops.scale_(x, f) if x.is_cuda else x.mul_(f)


class GradBuckets:
    """Persistent flat gradient buckets of one optimiser's parameters + their data-parallel exchange.

    Layout (static, identical on every rank): the parameters in REVERSE order -- backward reaches the last layers
    first -- packed into buckets of <= bucket_mb; every parameter's gradient lives at a fixed slice of its bucket
    (`GradSlot.view`, handed to Adam as p.grad).  During a train step the backward nodes deliver into the slots
    (`deliver` / the accumulate flags of the weight-gradient kernels); when the last expected node of the last
    parameter of a bucket has delivered, the bucket's collective is launched from inside the backward pass --
    asynchronously on the process group's stream (RCCL over xGMI; gloo in the single-GPU tests), buckets strictly in
    index order so that every rank issues the same sequence.  Replaces DataParallel's gradient reduce onto GPU 0 through
    10 MiB coalesced copies + re-broadcast of the weights ($SP/torch/cuda/comm.py:24,76-86; nn/parallel/_functions.py)
    and round 2's cat -> all_reduce -> divide -> copy-back (three extra passes over 1.1-1.5 GB per step): the bytes
    on the wire are the gradients themselves, in place.

    T2V_GRAD_RS_AG=1: reduce-scatter + all-gather per bucket instead of one all-reduce (what a ring all-reduce does
    internally; lets a sharded optimiser step sit between the two halves later).  T2V_GRAD_BUCKET_MB overrides 64."""

    def __init__(self, params, bucket_mb=None, name="", register=True):
        """register=False: the parameters keep the gradient slots they have (a temporary set of buckets over parameters a
        trainer owns must not take the trainer's slots away: ADVICE r3)"""
        import torch.distributed as dist
        self.params = list(params)
        self.name = name
        self.collecting = False
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # (T2V_TRAIN_FORCE_DIST=1 runs the collectives on a 1-rank group too: the RCCL path on a single GPU)
        self.exchange = self.world > 1 or (os.environ.get("T2V_TRAIN_FORCE_DIST") == "1" and dist.is_available()
                                            and dist.is_initialized())
        self.rs_ag = os.environ.get("T2V_GRAD_RS_AG", "0") == "1"
        if bucket_mb is None:
            bucket_mb = float(os.environ.get("T2V_GRAD_BUCKET_MB", "64"))
        limit = max(1, int(bucket_mb * (1 << 20) // 4))
        pad = 256 * max(1, self.world)     # bucket sizes divisible by the world size (reduce-scatter shards) and 1 KiB
        order = list(range(len(self.params)))[::-1]
        bounds, members, off, cur, cur_n = [], [], 0, [], 0
        for i in order:
            n = self.params[i].numel()
            if cur and cur_n + n > limit:
                size = (cur_n + pad - 1) // pad * pad
                bounds.append((off, off + size))
                members.append(cur)
                off += size
                cur, cur_n = [], 0
            cur.append((i, cur_n))
            cur_n += n
        if cur:
            size = (cur_n + pad - 1) // pad * pad
            bounds.append((off, off + size))
            members.append(cur)
            off += size
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.bounds = bounds
        self.slots = [None] * len(self.params)
        self.bucket_of = [None] * len(self.params)
        for bi, mem in enumerate(members):
            for i, rel in mem:
                p = self.params[i]
                view = self.flat[bounds[bi][0] + rel: bounds[bi][0] + rel + p.numel()].view(p.shape)
                self.slots[i] = GradSlot(self, bi, i, view)
                if register:
                    p._t2v_gslot = self.slots[i]
                self.bucket_of[i] = bi
        self.members = [[i for i, _ in mem] for mem in members]
        self.nbytes = 0
        self._works, self._open, self._launched = [], [], 0

    # -- one train step -----------------------------------------------------------------------------
    def begin_step(self):
        for sl in self.slots:
            sl.expect = sl.got = 0
            sl.filled = False
        self._main = torch.cuda.current_stream() if (self.params and self.params[0].is_cuda) else None
        self._open = [len(m) for m in self.members]      # parameters of each bucket still waiting for nodes
        self._done = [False] * len(self.params)
        self._works, self._launched, self.nbytes = [], 0, 0
        # T2V_GRAD_DIRECT=0: the backward nodes hand their gradients to autograd as before and absorb() copies the sums
        # into the slots afterwards (one extra pass, no launch from inside the backward pass): the A/B baseline
        self.collecting = os.environ.get("T2V_GRAD_DIRECT", "1") != "0"

    def seal(self):
        """The forward pass is over: parameters no node will deliver to (unused in this step) no longer hold their
        bucket back."""
        if not self.collecting:
            return
        for sl in self.slots:
            if sl.expect == 0 and not self._done[sl.idx]:
                self._done[sl.idx] = True
                self._open[sl.bucket] -= 1
        # (nothing is launched here: a launch from the first delivering node keeps the bucket order intact)

    def node_done(self, sl):
        sl.got += 1
        if sl.got == sl.expect and not self._done[sl.idx]:
            self._done[sl.idx] = True
            self._open[sl.bucket] -= 1
            self._launch_ready()

    def _launch_ready(self):
        """launch, in bucket order, every bucket whose parameters have all received their last expected gradient"""
        while self._launched < len(self.bounds) and self._open[self._launched] == 0:
            self._launch(self._launched)
            self._launched += 1

    def _launch(self, b):
        # parameters no backward node reached contribute exact zeros (and get no optimiser step: see finish)
        for i in self.members[b]:
            if not self.slots[i].filled and not self.slots[i].zero_valid:
                _zero(self.slots[i].view)
        if not self.exchange:
            return
        import torch.distributed as dist
        # The collective is ordered behind the stream it is issued on, and it needs the weight-gradient stream's kernels AND the
        # current stream's (and the step's main stream's, when this node runs on another).  With weight gradients in flight it
        # is issued FROM the weight-gradient stream, which first takes in what the other streams hold so far: the backward
        # pass itself does not stop.  (Until round 6 the current stream joined the weight-gradient stream here, 36 times per
        # step: the data-gradient chain waited for every layer's weight gradient, and the two queues of DESIGN 6b ran one
        # after the other again whenever the exchange was on.)  T2V_EXCHANGE_FROM_SIDE=0: the join.
        cur = torch.cuda.current_stream() if self.flat.is_cuda else None
        side = _WG_SIDE["stream"] if (cur is not None and _WG_SIDE["pending"]
                                      and os.environ.get("T2V_EXCHANGE_FROM_SIDE", "1") != "0") else None
        if side is not None and side != cur:
            side.wait_stream(cur)
            if self._main is not None and cur != self._main:
                side.wait_stream(self._main)
            with torch.cuda.stream(side):
                self._issue(b)
            return
        wgrad_join()
        if self._main is not None and torch.cuda.current_stream() != self._main:
            torch.cuda.current_stream().wait_stream(self._main)
        self._issue(b)

    def _issue(self, b):
        import torch.distributed as dist
        lo, hi = self.bounds[b]
        buf = self.flat[lo:hi]
        avg = dist.get_backend() == "nccl"       # RCCL averages in the collective; gloo sums (scaled in finish)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if self.rs_ag:      # (also on a forced 1-rank group: the RCCL entry points run on device tensors)
            shard = (hi - lo) // self.world
            r = dist.get_rank()
            mine = buf[r * shard:(r + 1) * shard]
            if avg:     # RCCL: in-place reduce-scatter (output = this rank's shard of the input), then the all-gather,
                w1 = dist.reduce_scatter_tensor(mine, buf, op=op, async_op=True)      # both stream-ordered
                self._works.append((w1, None, None, None))
                w2 = dist.all_gather_into_tensor(buf, mine, async_op=True)
                self._works.append((w2, buf, avg, None))
            else:       # gloo (tests): works are not ordered among themselves -- the all-gather is issued in finish()
                out = torch.empty_like(mine)
                w1 = dist.reduce_scatter_tensor(out, buf, op=op, async_op=True)
                self._works.append((w1, buf, avg, (mine, out)))
        else:
            self._works.append((dist.all_reduce(buf, op=op, async_op=True), buf, avg, None))
        self.nbytes += 4 * sum(self.params[i].numel() for i in self.members[b])     # payload (the padding travels too)

    def absorb(self, grads):
        """After autograd.grad: gradients that came back as tensors (parameters reached through torch-native plumbing,
        e.g. the cat of the two flow-head convs; a flushed Winograd reduction) are added into their slots; then every
        bucket not yet launched goes out, in order."""
        wgrad_join()
        for i, g in enumerate(grads):
            if g is not None:
                sl = self.slots[i]
                _acc(sl.view, g, not sl.filled)
                sl.filled = True
        for b in range(self._launched, len(self.bounds)):
            self._launch(b)
        self._launched = len(self.bounds)
        self.collecting = False

    def finish(self):
        """Wait for the collectives; p.grad = the slot (None where no rank-local node delivered: Adam then skips the
        parameter, as torch 0.4.1's does).  Returns the bytes exchanged."""
        import torch.distributed as dist
        wgrad_join()
        for work, buf, avg, shard in self._works:
            work.wait()
            if shard is not None:
                shard[0].copy_(shard[1])
                dist.all_gather_into_tensor(buf, shard[0].clone())
            if buf is not None and not avg and self.world > 1:
                _scale(buf, 1.0 / self.world)
        self._works = []
        for p, sl in zip(self.params, self.slots):
            p.grad = sl.view if sl.filled else None
        return self.nbytes

    def presence(self):
        """a checksum of WHICH parameters received gradients on this rank: must agree across ranks, or the replicas'
        optimiser steps differ (Adam skips parameters without gradient)"""
        return float(sum((k + 1) * (k + 7) for k, sl in enumerate(self.slots) if sl.filled) % 1000003)


def check_presence_across_ranks(buckets, device):
    """Every rank must have delivered gradients to the same parameters.  One tiny MAX / MIN all-reduce per step."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    v = [b.presence() for b in buckets]
    dev = device if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(v + [-x for x in v], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu().tolist()
    n = len(v)
    if any(t[k] != -t[n + k] for k in range(n)):
        raise RuntimeError("data-parallel ranks disagree on which parameters received gradients this step (checksums "
                           "max %s / min %s): their optimiser steps would diverge" % (t[:n], [-x for x in t[n:]]))


class GradientExchange:
    """Stand-alone form for gradients that already sit in p.grad (tests, callers outside the trainer): the gradients
    are moved into a temporary GradBuckets (one pass), exchanged bucket by bucket, and p.grad ends as the averaged slot."""

    def __init__(self, params, bucket_mb=64):
        ps = [p for p in params if p.grad is not None]
        self.b = GradBuckets(ps, bucket_mb, register=False) if ps else None
        if self.b is not None:
            self.b.begin_step()
            self.b.absorb([p.grad for p in ps])

    def finish(self):
        return self.b.finish() if self.b is not None else 0


def allreduce_gradients_begin(params, bucket_mb=64):
    return GradientExchange(params, bucket_mb)


def allreduce_gradients(params, bucket_mb=64):
    """Data-parallel gradient exchange of gradients already in p.grad: bucketed, averaged over ranks; returns the bytes
    exchanged.  Replaces DataParallel's reduce-to-GPU-0 + re-broadcast: replicas are persistent, so
    one collective per bucket is all the communication a step needs."""
    return GradientExchange(params, bucket_mb).finish()


# ------------------------------------------------------------------------------------------------
# face discriminator crop (--add_face_disc) and the training loop
# ------------------------------------------------------------------------------------------------
def get_face_region(pose_maps_u8, fine_size):
    """(ys, ye, xs, xe) of the face crop for a chunk of frames, or None when no frame shows the key colour.

    [RECALL upstream Vid2VidModelD.get_face_region, --openpose_only branch]: the pixels of ALL frames of the chunk
    whose colour is the nose-neck limb's [153,0,51] (keypoint2img.py:180; upstream tests the normalised map for
    R in (0.19,0.21), G < -0.99, B in (-0.61,-0.59), which is that uint8 colour) give one bounding box; the crop is
    centred on the box's midpoint, side fine_size//32*8 (128 at fineSize 512), the centre clamped to
    [side/2, dim-1-side/2]; with no such pixel upstream returns an empty region -- the face terms are skipped then.
    pose_maps_u8: [H,W,3] or [F,H,W,3]."""
    import numpy as np
    from .keypoints import NOSE_NECK_RGB
    a = np.asarray(pose_maps_u8)
    if a.ndim == 3:
        a = a[None]
    H, W = a.shape[1:3]
    side = max(8, fine_size // 32 * 8)
    _, ys, xs = np.nonzero((a == np.array(NOSE_NECK_RGB, np.uint8)).all(3))
    if not ys.size:
        return None
    yc, xc = (int(ys.min()) + int(ys.max())) // 2, (int(xs.min()) + int(xs.max())) // 2
    yc = max(side // 2, min(H - 1 - side // 2, yc))
    xc = max(side // 2, min(W - 1 - side // 2, xc))
    return yc - side // 2, yc + side // 2, xc - side // 2, xc + side // 2


def discriminator_state_dict(input_nc, ndf, n_layers, num_D, norm, seed):
    """seeded random init in upstream key names (weights_init: N(0,0.02) convs, N(1,0.02) BN weight)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = {}
    for i in range(num_D):
        nd = min(ndf * 2 ** (num_D - 1 - i), 64)
        chans = [(input_nc, nd, False)]
        nf = nd
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            chans.append((nf_prev, nf, True))
        nf_prev, nf = nf, min(nf * 2, 512)
        chans += [(nf_prev, nf, True), (nf, 1, False)]
        for j, (cin, cout, has_norm) in enumerate(chans):
            pre = "scale%d_layer%d" % (i, j)
            sd[pre + ".0.weight"] = torch.from_numpy(rng.normal(0, 0.02, (cout, cin, 4, 4)).astype("float32"))
            b = 1.0 / (cin * 16) ** 0.5
            sd[pre + ".0.bias"] = torch.from_numpy(rng.uniform(-b, b, (cout,)).astype("float32"))
            if has_norm and norm == "batch":
                sd[pre + ".1.weight"] = torch.from_numpy(rng.normal(1, 0.02, (cout,)).astype("float32"))
                sd[pre + ".1.bias"] = torch.zeros(cout)
    return sd


class Vid2VidTrainer:
    """One rank of the data-parallel train step: 1 sequence per GPU, `max_frames_per_gpu`
    frames per chunk, previous frames detached (max_frames_backpropagate 1)."""

    def __init__(self, opt, device="cuda:0", seed=1):
        self.opt, self.device = opt, device
        # flags of upstream's train.py that would train a DIFFERENT configuration here: refuse instead of ignoring
        for flag, bad, why in (("no_lsgan", bool(getattr(opt, "no_lsgan", False)), "only the LSGAN (MSE) objective is built"),
                               ("n_scales_spatial", getattr(opt, "n_scales_spatial", 1) > 1, "the train step covers the global generator"),
                               ("max_frames_backpropagate", getattr(opt, "max_frames_backpropagate", 1) > 1,
                                "generated previous frames are always detached (the recipe's default, 1)"),
                               ("use_single_G", bool(getattr(opt, "use_single_G", False)), "no first-frame generator"),
                               ("fg", bool(getattr(opt, "fg", False)), "no foreground / background generators")):
            if bad:
                raise NotImplementedError("--%s: %s" % (flag, why))
        for flag, on in (("pool_size", getattr(opt, "pool_size", 1) > 1), ("niter_fix_global", getattr(opt, "niter_fix_global", 0) > 0)):
            if on:
                print("warning: --%s has no effect here (single-scale training, no image pool)" % flag, flush=True)
        input_nc = opt.label_nc if opt.label_nc != 0 else opt.input_nc
        self.spec = GeneratorSpec(input_nc=input_nc * opt.n_frames_G, prev_nc=(opt.n_frames_G - 1) * opt.output_nc,
                                  ngf=opt.ngf, n_downsample=opt.n_downsample_G, n_blocks=opt.n_blocks,
                                  no_flow=bool(opt.no_flow), norm=opt.norm)
        self.G = TrainableGenerator(self.spec, synthetic_state_dict(self.spec, seed, "vid2vid"), device)
        d_in = input_nc + opt.output_nc
        self.D = TrainableDiscriminator(d_in, discriminator_state_dict(d_in, opt.ndf, opt.n_layers_D, opt.num_D, opt.norm,
                                                                       seed + 1), opt.ndf, opt.n_layers_D, opt.num_D,
                                        opt.norm, device)
        self.Df = None
        if opt.add_face_disc:
            nf = max(1, opt.num_D - 2)
            self.Df = TrainableDiscriminator(d_in, discriminator_state_dict(d_in, opt.ndf, opt.n_layers_D, nf, opt.norm,
                                                                            seed + 2), opt.ndf, opt.n_layers_D, nf,
                                             opt.norm, device)
        # temporal discriminators netD_T{s}: a multiscale PatchGAN over n_frames_D frames spaced
        # n_frames_D**s apart + the n_frames_D-1 optical flows between them (13 channels).  FlowNet2 is not in the
        # reference tree, so the flow channels are zero; the frame channels train as upstream.
        self.tD = opt.n_frames_D
        self.DT = []
        for s in range(max(0, opt.n_scales_temporal)):
            dt_in = opt.output_nc * self.tD + 2 * (self.tD - 1)
            self.DT.append(TrainableDiscriminator(dt_in, discriminator_state_dict(dt_in, opt.ndf, opt.n_layers_D, opt.num_D,
                                                                                  opt.norm, seed + 10 + s),
                                                  opt.ndf, opt.n_layers_D, opt.num_D, opt.norm, device))
        self._hist_real, self._hist_fake = [], []    # frames of the running sequence (fakes detached)
        d_params = list(self.D.parameters()) + (list(self.Df.parameters()) if self.Df else [])
        for dt in self.DT:
            d_params += list(dt.parameters())
        # perceptual loss (the reference trains with it: README.md:171-176 passes no --no_vgg); its weights come from
        # torchvision's download in the reference and from --vgg_weights here
        self.vgg = None
        if not opt.no_vgg:
            if getattr(opt, "vgg_weights", ""):
                self.vgg = HipVGG19Features(torch.load(opt.vgg_weights, map_location="cpu"), device)
            elif getattr(opt, "vgg_random_init", False):
                self.vgg = HipVGG19Features(vgg19_random_state_dict(seed + 20), device)
            else:
                print("VGG loss off: no --vgg_weights <torchvision vgg19 .pth> given (the file is not in the reference "
                      "tree; --no_vgg silences this)", flush=True)
        lr_g = lr_d = opt.lr
        betas = (opt.beta1, 0.999)
        if getattr(opt, "TTUR", False):    # [RECALL upstream: two time-scale update rule]
            betas, lr_g, lr_d = (0.0, 0.9), opt.lr / 2.0, opt.lr * 2.0
        self.lr_scale = (lr_g / opt.lr, lr_d / opt.lr)
        self.optG = FusedAdam(self.G.parameters(), lr_g, betas)
        self.optD = FusedAdam(d_params, lr_d, betas)
        # persistent flat gradient buckets (the backward nodes write into them; exchanged in place)
        self.bucketsG = GradBuckets(self.optG.params, name="G")
        self.bucketsD = GradBuckets(self.optD.params, name="D")
        if getattr(opt, "load_pretrain", ""):
            self.load(opt.which_epoch, opt.load_pretrain)
        self.comm_bytes, self.comm_ms = 0, 0.0
        self.time_comm = os.environ.get("T2V_TRAIN_COMM_TIMING", "0") == "1"

    def _d_input(self, A3, img4):
        z = torch.zeros(A3.shape[:-1] + (2,), dtype=torch.float32, device=A3.device)
        return torch.cat([A3, img4[..., :3], z], -1).contiguous()

    def _vgg_term(self, fake, raw, real):
        """lambda_feat * VGGLoss(fake, real) (+ the same on the raw frame with a flow branch) [RECALL upstream: criterionVGG],
        or None without a VGG19"""
        if self.vgg is None:
            return None
        with torch.no_grad():
            real_taps = self.vgg(real)
        loss = vgg_loss(self.vgg, fake, real, real_taps) * self.opt.lambda_feat
        if raw is not None:
            loss = loss + vgg_loss(self.vgg, raw, real, real_taps) * self.opt.lambda_feat
        return loss

    def _flow_terms(self, fake, fw_all, real, real_prev, prevs, flow_ref, conf_ref, first):
        """-> (sum of the flow / warp / weight losses, {name: term})"""
        opt, dev = self.opt, fake.device
        F_, H, W = fake.shape[0], fake.shape[1], fake.shape[2]
        # flow / warp / weight losses [RECALL upstream Vid2VidModelD.forward, compute_flow_losses]:
        #   F_Flow = MaskedL1(flow, flow_ref, conf) * lambda_F      F_Warp = MaskedL1(resample(real_B_prev, flow), real_B, conf) * lambda_T
        #   W      = MaskedL1(weight, 0, conf)  (--no_first_img)    G_Warp = MaskedL1(fake_B, resample(fake_B_prev, flow_ref), conf) * lambda_T
        # flow_ref / conf_ref come from FlowNet2 upstream; here they are inputs (zero flow by default)
        with torch.no_grad():
            if flow_ref is None:
                if not getattr(self, "_warned_zero_flow", False):
                    self._warned_zero_flow = True
                    print("warning: flow / warp / weight losses run against a synthetic ZERO reference flow (FlowNet2 is not "
                          "in the reference tree; pass flow_ref / conf_ref, or --no_flow to train without the flow branch)",
                          flush=True)
                flow_ref = torch.zeros(F_, H, W, 4, dtype=torch.float32, device=dev)
                real_prev_warp = real_prev
            else:
                real_prev_warp = torch.stack([ops.flow_warp(flow_ref[i], real_prev[i], 0) for i in range(F_)])
            if conf_ref is None:
                conf_ref = ((real[..., :3] - real_prev_warp[..., :3]).norm(dim=-1) < 0.02).float()
            fake_prev = torch.cat(prevs, 0)
            fake_prev_warp = torch.stack([ops.flow_warp(flow_ref[i], fake_prev[i], self.spec.prev_nc - 3)
                                          for i in range(F_)])
            if first:
                # no generated previous frame exists for a sequence's first frame: upstream warps the REAL previous
                # frame there (fake_B_prev = real_B_prev[:, 0:1] when there is no previous chunk [RECALL
                # compute_fake_B_prev]); the all-zero FIFO stays the generator's INPUT only
                fake_prev_warp[0] = real_prev_warp[0]
        loss_F_flow = masked_l1(fw_all, flow_ref, conf_ref, 2, 0) * opt.lambda_F
        loss_F_warp = masked_l1(_Resample.apply(fw_all, real_prev.contiguous(), 0), real, conf_ref, 3) * opt.lambda_T
        # the weight-map loss exists only under --no_first_img upstream (zero otherwise) [RECALL]
        loss_W = masked_l1(fw_all, None, conf_ref, 1, 2) if getattr(opt, "no_first_img", False) else 0.0
        loss_G_warp = masked_l1(fake, fake_prev_warp, conf_ref, 3) * opt.lambda_T
        return (loss_F_flow + loss_F_warp + loss_W + loss_G_warp,
                {"F_Flow": loss_F_flow, "F_Warp": loss_F_warp, "W": loss_W, "G_Warp": loss_G_warp})

    def _d_pass(self, net, inputs, book, names, face_weight=1.0):
        """One discriminator on its real input and its one or two generated inputs (fake, raw) as ONE batch of passes
        (TrainableDiscriminator.forward(groups=...)): every layer is one conv launch, one data-gradient launch and one
        weight-gradient launch for all passes.  Registers the LSGAN and feature-matching terms of compute_loss_D /
        compute_loss_G [RECALL upstream Vid2VidModelD] with `book` and returns (tensors, seeds for G's backward, tensors,
        seeds for D's backward).
        The one forward on a generated input serves both losses (as T2V_D_SHARED_FWD did for the one-pass-per-launch path): D's
        loss reaches D's parameters through it, G's loss reaches the frames through it with D's parameter gradients switched
        off (param_gradients_off); the running statistics still move twice for those passes (upstream's two forwards).
        names = (D, G_GAN, G_GAN_Feat) of the reported sums: D = 0.5 (real + fake [+ raw, with the real term counted again
        as upstream does])."""
        opt = self.opt
        n_d, n_gan, n_feat = names
        P = len(inputs)
        F_ = inputs[0].shape[0]
        feats = net(torch.cat(inputs, 0), groups=P, bn_times=[1] + [2] * (P - 1), pt_skip=F_)
        num_D = len(feats)
        fm_w = 0.0 if opt.no_ganFeat else (1.0 / num_D) * (4.0 / (opt.n_layers_D + 1)) * opt.lambda_feat * face_weight
        tG, sG, tD, sD = [], [], [], []
        if not fm_w:
            book.zero_names.append(n_feat)        # (--no_ganFeat: the term is reported as 0, as the pass-by-pass form does)
        for st in feats:
            logits = st[-1]
            gG, gD = torch.empty_like(logits), torch.empty_like(logits)
            book.zero(gG[:F_])
            book.mse(n_d, logits[:F_], 1.0, 0.5 * (P - 1), seed=gD[:F_])
            for g in range(1, P):
                sl = slice(g * F_, (g + 1) * F_)
                book.mse(n_d, logits[sl], 0.0, 0.5, seed=gD[sl])
                book.mse(n_gan, logits[sl], 1.0, face_weight, seed=gG[sl])
            tG.append(logits); sG.append(gG); tD.append(logits); sD.append(gD)
            if fm_w:
                for f in st[:-1]:
                    gf = torch.empty_like(f)
                    book.zero(gf[:F_])
                    for g in range(1, P):
                        sl = slice(g * F_, (g + 1) * F_)
                        book.l1(n_feat, f[sl], f[:F_], fm_w, seed=gf[sl])
                    tG.append(f); sG.append(gf)
        return tG, sG, tD, sD

    def _d_backward_on_its_own_stream(self, outputs, seeds, d_params):
        """The discriminators' own backward pass (their LSGAN terms: seeds on the logits only) without the autograd engine:
        every output is the end of a chain of _ConvBlock nodes, a node object is its own `ctx`, so each chain is
        `dy = _ConvBlock.backward(node, dy)[0]` from the logits down to the layer that owes no data gradient
        (input_gradients_off).  The nodes deliver their parameter gradients into the bucket slots as under the engine -- the
        same kernels on the same operands in the same order: the same bits -- but on THIS call's stream, the trainer's
        `_d_stream`, which waits for everything the current stream holds (forward passes, loss seeds) and which the caller
        joins before it touches D's gradients.  The engine would run these nodes on the stream of their forward, i.e. behind
        the generator's whole backward pass.  The graph stays alive until the step returns (the generator's pass retains it),
        so the saved tensors both passes read are not handed back to the allocator while this stream still reads them.
        Returns the stream, or None: not switched on (T2V_D_BWD_STREAM=1; OFF by default: the step is 1 ms faster with it where
        no process group exists -- scripts/train_bench.py, 86.9 -> 85.9 ms -- and 3 ms SLOWER in bench.py's block, which runs
        the step as a rank of a job: 84.4 -> 87.3 ms) or not applicable (a parameter without a bucket slot --
        T2V_GRAD_DIRECT=0 -- or an output that is no _ConvBlock result): the caller takes the engine."""
        if os.environ.get("T2V_D_BWD_STREAM", "0") != "1" or not outputs or not outputs[0].is_cuda:
            return None
        node_cls = _ConvBlock._backward_cls
        if any(grad_slot(p) is None for p in d_params) or any(not isinstance(y.grad_fn, node_cls) for y in outputs):
            return None
        if getattr(self, "_d_stream", None) is None:
            self._d_stream = torch.cuda.Stream()
        side = self._d_stream
        side.wait_stream(torch.cuda.current_stream())
        # the weight gradients of these nodes stay ON this stream (it is a side stream already): on the shared weight-gradient
        # stream they would stand in front of every bucket the generator's pass sends off (GradBuckets._launch orders a
        # collective behind that stream) -- the filler would hold the exchange up (measured with the join of that time:
        # 90.5 ms with the exchange against 82.1 without).  Two streams carry GEMMs from here on: one block per CU
        if os.environ.get("T2V_TRAIN_SK_HINT", "1") != "0" and wgrad_stream_on(outputs[0]):
            _WG_SIDE["gemm"] = True
        _WG_SIDE["inline"] = True
        try:
            with torch.cuda.stream(side), input_gradients_off():
                for y, g in zip(outputs, seeds):
                    node = y.grad_fn
                    while g is not None and isinstance(node, node_cls):
                        res = _ConvBlock.backward(node, g)
                        if any(r is not None for r in res[1:]):
                            raise RuntimeError("discriminator backward by hand: a node returned a gradient it should have delivered")
                        g, node = res[0], node.next_functions[0][0]
        finally:
            _WG_SIDE["inline"] = False
        side_gemm_hint()
        return side

    def _train_step_batched(self, pose, real, face_boxes, prev, real_prev, flow_ref, conf_ref, first, fakes, raws, fws, prevs,
                            fake, A3):
        """The step from the generated frames on (default path): every discriminator runs its real / fake / raw passes as one
        batch (_d_pass), the LSGAN and feature-matching terms are one launch (LossBook), and both backward passes start from
        the terms' gradient seeds.  The flow / warp / weight and VGG terms stay scalar nodes of the autograd graph."""
        opt, dev = self.opt, pose.device
        F_, H, W = pose.shape[0], pose.shape[1], pose.shape[2]
        flow_on = not self.spec.no_flow
        book = self.book = getattr(self, "book", None) or LossBook(dev)
        book.reset()
        raw = torch.cat(raws, 0) if flow_on else None
        fw_all = torch.cat(fws, 0) if (flow_on and real_prev is not None) else None
        tG, sG, tD, sD = self._d_pass(self.D, [self._d_input(A3, real), self._d_input(A3, fake)]
                                      + ([self._d_input(A3, raw)] if flow_on else []), book, ("D", "G_GAN", "G_GAN_Feat"))
        rest, extra = None, {}          # the terms that stay on the scalar graph: their sum, their values
        vgg_term = self._vgg_term(fake, raw, real)
        if vgg_term is not None:
            rest, extra["G_VGG"] = vgg_term, vgg_term
        if flow_on and real_prev is not None:
            flow_sum, flow_named = self._flow_terms(fake, fw_all, real, real_prev, prevs, flow_ref, conf_ref, first)
            rest = flow_sum if rest is None else rest + flow_sum
            extra.update(flow_named)
        if self.Df is not None and face_boxes is not None:
            def crop(t):
                return torch.stack([t[i, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(face_boxes)]).contiguous()
            cA = crop(A3)
            # face_weight = 2 on the generator's face terms [RECALL upstream Vid2VidModelD.forward]
            a, b_, c, d = self._d_pass(self.Df, [self._d_input(cA, crop(real)), self._d_input(cA, crop(fake))], book,
                                         ("D_f", "G_f_GAN", "G_f_GAN_Feat"), 2.0)
            tG += a; sG += b_; tD += c; sD += d
        # temporal discriminators: every window (t-2d, t-d, t), d = n_frames_D**s, that ends on a frame of this chunk;
        # older frames come from the sequence history (generated ones detached)
        if self.DT:
            n_old = len(self._hist_real)
            reals = self._hist_real + [real[i] for i in range(F_)]
            fks = self._hist_fake + [fake[i] for i in range(F_)]
            for sc, dt in enumerate(self.DT):
                d_ = self.tD ** sc
                ends = [t for t in range(n_old, n_old + F_) if t - (self.tD - 1) * d_ >= 0]
                if not ends:
                    continue

                def stack(frames):
                    rows = []
                    for t in ends:
                        w = [frames[t - (self.tD - 1 - k) * d_][..., :3] for k in range(self.tD)]
                        z = torch.zeros(H, W, ops.round_up(dt.input_nc, 4) - 3 * self.tD, dtype=torch.float32, device=dev)
                        rows.append(torch.cat(w + [z], -1))
                    return torch.stack(rows).contiguous()
                a, b_, c, d = self._d_pass(dt, [stack(reals), stack(fks)], book,
                                           ("D_T%d" % sc, "G_T_GAN%d" % sc, "G_T_GAN_Feat%d" % sc))
                tG += a; sG += b_; tD += c; sD += d
            keep = (self.tD - 1) * self.tD ** (len(self.DT) - 1)
            self._hist_real = [r.detach() for r in reals][-keep:]
            self._hist_fake = [f.detach() for f in fks][-keep:]
        book.run()          # values and gradient seeds of all LSGAN / feature-matching terms: one launch
        g_params, d_params = self.optG.params, self.optD.params
        self.bucketsG.seal()
        self.bucketsD.seal()
        if rest is not None and torch.is_tensor(rest) and rest.requires_grad:
            tG, sG = tG + [rest], sG + [None]
        # D's loss reaches D's parameters through plain chains of _ConvBlock nodes (logits -> ... -> first layer, no data
        # gradient into the frames) and shares nothing with the generator's backward pass but the saved tensors both read:
        # with T2V_D_BWD_STREAM=1 the chains are walked by hand on a stream of their own BEFORE the generator's pass is
        # enqueued, so that the GPU runs them beside its first ~20 ms, which otherwise have no second queue (measured both
        # ways, off by default: DESIGN 6b "the discriminators' own backward pass beside the generator's")
        d_side = self._d_backward_on_its_own_stream(tD, sD, d_params)
        with param_gradients_off(d_params):
            gG = torch.autograd.grad(tG, g_params, grad_outputs=sG, retain_graph=True, allow_unused=True)
        gG = flush_pending_weight_gradients(g_params, gG)
        self.bucketsG.absorb(gG)
        if d_side is not None:
            torch.cuda.current_stream().wait_stream(d_side)
            gD = [None] * len(d_params)
        else:
            with input_gradients_off():      # D's loss: no data gradient into the (attached) generated frames
                gD = torch.autograd.grad(tD, d_params, grad_outputs=sD, allow_unused=True)
        gD = flush_pending_weight_gradients(d_params, gD)
        self.bucketsD.absorb(gD)
        if self.time_comm:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.comm_bytes = self.bucketsG.finish() + self.bucketsD.finish()
        if self.time_comm and self.comm_bytes:
            torch.cuda.synchronize()
            self.comm_ms = 1e3 * (time.perf_counter() - t0)
        check_presence_across_ranks([self.bucketsG, self.bucketsD], dev)
        self.optG.step()
        self.optD.step()
        losses = book.read({k: v for k, v in extra.items() if torch.is_tensor(v)})      # (the step's one host read)
        losses.update({k: float(v) for k, v in extra.items() if not torch.is_tensor(v)})
        for k in book.zero_names:
            losses.setdefault(k, 0.0)
        ops.check_async_errors()      # (the read-back above synchronised: errors the step's kernels could only flag)
        # (reported in the order the one-term-per-launch path reports them)
        order = ["G_GAN", "G_GAN_Feat", "D", "G_VGG", "F_Flow", "F_Warp", "W", "G_Warp", "G_f_GAN", "G_f_GAN_Feat", "D_f"] + \
            [n % sc for sc in range(len(self.DT)) for n in ("G_T_GAN%d", "G_T_GAN_Feat%d", "D_T%d")]
        return {k: losses[k] for k in order if k in losses}, prev

    def train_step(self, pose, real, face_boxes=None, prev=None, real_prev=None, flow_ref=None, conf_ref=None):
        """pose [F,H,W,12] (sliding windows), real [F,H,W,4] NHWC on the device; face_boxes: list of
        (ys,ye,xs,xe) per frame; prev: the carried FIFO of generated frames (None: a new sequence starts).
        With the flow branch: real_prev [F,H,W,4] = the real frame before each frame enables the flow / warp
        losses against flow_ref [F,H,W,4] (flow_x, flow_y in pixels; default zero flow) and its confidence mask
        conf_ref [F,H,W] (default: ||real - resample(real_prev, flow_ref)|| < 0.02, the rule upstream's FlowNet2
        wrapper applies [RECALL]).  Returns (dict of scalar losses, FIFO for the next chunk)."""
        with batched_weight_gradients(self.optG.params + self.optD.params):
            return self._train_step(pose, real, face_boxes, prev, real_prev, flow_ref, conf_ref)

    def _train_step(self, pose, real, face_boxes, prev, real_prev=None, flow_ref=None, conf_ref=None):
        opt, dev = self.opt, pose.device
        F_, H, W = pose.shape[0], pose.shape[1], pose.shape[2]
        self.bucketsG.begin_step()
        self.bucketsD.begin_step()
        flow_on = not self.spec.no_flow
        first = prev is None
        if first:   # a new sequence starts: zero previous frames, raw-only first frame (--no_first_img)
            prev = torch.zeros(1, H, W, ops.round_up(self.spec.prev_nc, 4), dtype=torch.float32, device=dev)
            self._hist_real, self._hist_fake = [], []
        fakes, raws, fws, prevs = [], [], [], []
        for f in range(F_):
            fk, rw, fw = self.G(pose[f:f + 1], prev, use_raw_only=first and f == 0, full=True,
                                need_flow=real_prev is not None)
            fakes.append(fk)
            raws.append(rw)
            fws.append(fw)
            prevs.append(prev)
            nprev = torch.zeros_like(prev)
            nprev[..., :3] = prev[..., 3:6]
            nprev[..., 3:6] = fk.detach()[..., :3]
            prev = nprev
        fake = torch.cat(fakes, 0)
        A3 = pose[..., 6:9]
        # compute_loss_D(netD, real_A, real_B, fake) [RECALL upstream Vid2VidModelD]: real pass, fake pass on the
        # detached frame (D's loss), fake pass for G's GAN + feature-matching loss.  The generator-side passes run
        # with D's parameters detached: only their data gradient is wanted.
        # One forward on the fake frames serves both losses (T2V_D_SHARED_FWD=0: two forwards, as upstream runs them): D's
        # loss reaches D's parameters through it, G's loss reaches the frames through it with D's parameter gradients
        # switched off for that backward pass (param_gradients_off) -- the same values either way, a D forward less per
        # fake input.  The running statistics still move twice (upstream's two forwards).
        shared = os.environ.get("T2V_D_SHARED_FWD", "1") != "0"
        # T2V_D_BATCHED=0: one launch per pass and per loss term, the scalar graph on autograd (the form this path replaced;
        # kept as its test twin)
        batched = shared and os.environ.get("T2V_D_BATCHED", "1") != "0"
        self._shared_d = shared
        if batched:
            return self._train_step_batched(pose, real, face_boxes, prev, real_prev, flow_ref, conf_ref, first, fakes, raws, fws,
                                            prevs, fake, A3)

        def d_fake(net, x_attached):
            """-> (prediction for D's loss, prediction for G's loss)"""
            if shared:
                with bn_updates(2):
                    pf = net(x_attached)
                return pf, pf
            return net(x_attached.detach()), net(x_attached, frozen=True)

        pr = self.D(self._d_input(A3, real))
        pfd, pfg = d_fake(self.D, self._d_input(A3, fake))
        loss_D_real, loss_D_fake = gan_loss(pr, True), gan_loss(pfd, False)
        loss_G_gan = gan_loss(pfg, True)
        loss_G_fm = feature_matching_loss(pfg, pr, opt.n_layers_D, opt.lambda_feat) if not opt.no_ganFeat else 0.0
        raw = fw_all = None
        if flow_on:
            # with a flow branch upstream scores fake_B_raw with the same discriminator (and VGG) as well; its second
            # real pass returns the values of the first (same weights, same batch): that term is simply added again
            raw = torch.cat(raws, 0)
            fw_all = torch.cat(fws, 0) if real_prev is not None else None
            pfd_r, pfg_r = d_fake(self.D, self._d_input(A3, raw))
            loss_D_real = loss_D_real + gan_loss(pr, True)
            loss_D_fake = loss_D_fake + gan_loss(pfd_r, False)
            loss_G_gan = loss_G_gan + gan_loss(pfg_r, True)
            if not opt.no_ganFeat:
                loss_G_fm = loss_G_fm + feature_matching_loss(pfg_r, pr, opt.n_layers_D, opt.lambda_feat)
        loss_D = 0.5 * (loss_D_fake + loss_D_real)
        loss_G = loss_G_gan + loss_G_fm
        loss_G_vgg = self._vgg_term(fake, raw if flow_on else None, real)
        if loss_G_vgg is not None:
            loss_G = loss_G + loss_G_vgg
        def _f(t):   # kept on the device: one host read at the end of the step instead of a sync per loss term
            return t.detach() if torch.is_tensor(t) else float(t)

        losses = {"G_GAN": _f(loss_G_gan), "G_GAN_Feat": _f(loss_G_fm), "D": _f(loss_D)}
        if loss_G_vgg is not None:
            losses["G_VGG"] = _f(loss_G_vgg)
        if flow_on and real_prev is not None:
            flow_sum, flow_named = self._flow_terms(fake, fw_all, real, real_prev, prevs, flow_ref, conf_ref, first)
            loss_G = loss_G + flow_sum
            losses.update({k: _f(v) for k, v in flow_named.items()})
        if self.Df is not None and face_boxes is not None:
            def crop(t):
                return torch.stack([t[i, b[0]:b[1], b[2]:b[3]] for i, b in enumerate(face_boxes)]).contiguous()
            fr = self.Df(self._d_input(crop(A3), crop(real)))
            ffd, ffg = d_fake(self.Df, self._d_input(crop(A3), crop(fake)))
            loss_Df = 0.5 * (gan_loss(ffd, False) + gan_loss(fr, True))
            # face_weight = 2 on the generator's face terms [RECALL upstream Vid2VidModelD.forward]
            lg = gan_loss(ffg, True) * 2.0
            lf = feature_matching_loss(ffg, fr, opt.n_layers_D, opt.lambda_feat) * 2.0 if not opt.no_ganFeat else 0.0
            loss_G = loss_G + lg + lf
            loss_D = loss_D + loss_Df
            losses.update({"G_f_GAN": _f(lg), "G_f_GAN_Feat": _f(lf), "D_f": _f(loss_Df)})
        # temporal discriminators: every window (t-2d, t-d, t), d = n_frames_D**s, that ends on a frame of this chunk;
        # older frames come from the sequence history (generated ones detached)
        if self.DT:
            n_old = len(self._hist_real)
            reals = self._hist_real + [real[i] for i in range(F_)]
            fks = self._hist_fake + [fake[i] for i in range(F_)]
            for sc, dt in enumerate(self.DT):
                d = self.tD ** sc
                ends = [t for t in range(n_old, n_old + F_) if t - (self.tD - 1) * d >= 0]
                if not ends:
                    continue
                def stack(frames):
                    rows = []
                    for t in ends:
                        w = [frames[t - (self.tD - 1 - k) * d][..., :3] for k in range(self.tD)]
                        z = torch.zeros(H, W, ops.round_up(dt.input_nc, 4) - 3 * self.tD, dtype=torch.float32, device=dev)
                        rows.append(torch.cat(w + [z], -1))
                    return torch.stack(rows).contiguous()
                tr, tf = stack(reals), stack(fks)
                pr_t = dt(tr)
                pfd_t, pfg_t = d_fake(dt, tf)
                l_dt = 0.5 * (gan_loss(pfd_t, False) + gan_loss(pr_t, True))
                lg = gan_loss(pfg_t, True)
                lf = feature_matching_loss(pfg_t, pr_t, opt.n_layers_D, opt.lambda_feat) if not opt.no_ganFeat else 0.0
                loss_G = loss_G + lg + lf
                loss_D = loss_D + l_dt
                losses.update({"G_T_GAN%d" % sc: _f(lg), "G_T_GAN_Feat%d" % sc: _f(lf), "D_T%d" % sc: _f(l_dt)})
            keep = (self.tD - 1) * self.tD ** (len(self.DT) - 1)
            self._hist_real = [r.detach() for r in reals][-keep:]
            self._hist_fake = [f.detach() for f in fks][-keep:]
        g_params = self.optG.params
        d_params = self.optD.params
        # G's backward: the nodes deliver into bucketsG and launch its buckets' collectives as they complete; what is
        # left goes out in absorb() -- all of it in flight under the discriminators' backward pass
        self.bucketsG.seal()
        self.bucketsD.seal()
        with param_gradients_off(d_params if getattr(self, "_shared_d", False) else []):
            gG = torch.autograd.grad(loss_G, g_params, retain_graph=True, allow_unused=True)
        gG = flush_pending_weight_gradients(g_params, gG)
        self.bucketsG.absorb(gG)
        with input_gradients_off():      # D's loss: no data gradient into the (attached) fake frames
            gD = torch.autograd.grad(loss_D, d_params, allow_unused=True)
        gD = flush_pending_weight_gradients(d_params, gD)      # (a discriminator layer one of whose passes fed no loss term)
        self.bucketsD.absorb(gD)
        if self.time_comm:       # T2V_TRAIN_COMM_TIMING=1: what of the exchange is still running once the backward kernels
            torch.cuda.synchronize()      # have drained = its exposed (not hidden) part
        t0 = time.perf_counter()
        self.comm_bytes = self.bucketsG.finish() + self.bucketsD.finish()
        if self.time_comm and self.comm_bytes:
            torch.cuda.synchronize()
            self.comm_ms = 1e3 * (time.perf_counter() - t0)
        check_presence_across_ranks([self.bucketsG, self.bucketsD], dev)
        self.optG.step()
        self.optD.step()
        keys = [k for k, v in losses.items() if torch.is_tensor(v)]
        if keys:
            for k, v in zip(keys, torch.stack([losses[k].reshape(()) for k in keys]).tolist()):
                losses[k] = v
            ops.check_async_errors()      # (the read-back above synchronised: errors the step's kernels could only flag)
        return losses, prev

    def update_learning_rate(self, epoch):
        """linear decay to zero over the niter_decay epochs after `niter` [RECALL upstream update_learning_rate]"""
        lr = self.opt.lr * max(0.0, 1.0 - (epoch - self.opt.niter) / float(max(1, self.opt.niter_decay)))
        print("update learning rate: %f -> %f" % (self.optG.lr / self.lr_scale[0], lr), flush=True)
        self.optG.lr, self.optD.lr = lr * self.lr_scale[0], lr * self.lr_scale[1]

    def load(self, epoch_label, directory=None):
        """--continue_train / --load_pretrain DIR: the nets saved by save() (upstream file names); missing files keep
        their initialisation"""
        import os
        d = directory or os.path.join(self.opt.checkpoints_dir, self.opt.name)
        nets = [("G0", self.G), ("D", self.D)] + ([("D_f", self.Df)] if self.Df is not None else []) + \
               [("D_T%d" % sc, dt) for sc, dt in enumerate(self.DT)]
        for tag, net in nets:
            path = os.path.join(d, "%s_net_%s.pth" % (epoch_label, tag))
            if not os.path.exists(path):
                print("continue_train: %s not found, keeping the initial weights" % path, flush=True)
                continue
            sd = torch.load(path, map_location="cpu")
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}    # saved from nn.DataParallel
            with torch.no_grad():
                for k, p in net.named_upstream_parameters().items():
                    if k not in sd:
                        print("continue_train: %s has no %s, keeping its initial value" % (path, k), flush=True)
                        continue
                    p.copy_(sd[k].to(p.device, torch.float32))
                    p._t2v_packs = p._t2v_repack = p._t2v_pack_event = None
                    if k.endswith(".weight") and p.dim() == 1 and k[:-len("weight")] + "running_mean" in sd:
                        base = k[:-len("weight")]
                        nbt = sd.get(base + "num_batches_tracked")
                        p._t2v_running = [sd[base + "running_mean"].to(p.device, torch.float32).clone(),
                                          sd[base + "running_var"].to(p.device, torch.float32).clone(),
                                          int(nbt) if nbt is not None else 0]

    def save(self, epoch_label, progress=None):
        """progress = (epoch, samples done in it): written to iter.txt beside the nets every time `latest` is saved, so
        that --continue_train never pairs newer weights with an older position."""
        import os
        d = os.path.join(self.opt.checkpoints_dir, self.opt.name)
        os.makedirs(d, exist_ok=True)
        if progress is not None and epoch_label == "latest":
            with open(os.path.join(d, "iter.txt"), "w") as fh:
                fh.write("%d\n%d\n" % (int(progress[0]), int(progress[1])))
        def state_dict(net):
            """upstream's file layout: the parameters plus, for every BatchNorm2d (a 1-D `.weight`), the buffers torch
            0.4.1's module carries ($SP/torch/nn/modules/batchnorm.py: running_mean, running_var, num_batches_tracked)
            so that the file also loads strictly into the reference's own modules.  The norm layers run on batch
            statistics in training AND at test time, so the running statistics are never read: they are
            tracked all the same (running_update_args: momentum 0.1, unbiased variance, one update per forward) and
            written as they stand."""
            sd = {}
            for k, v in net.named_upstream_parameters().items():
                sd[k] = v.detach().cpu()
                if k.endswith(".weight") and v.dim() == 1 and self.opt.norm == "batch":
                    base = k[:-len("weight")]
                    rs = running_stats(v)
                    sd[base + "running_mean"] = rs[0].detach().cpu()
                    sd[base + "running_var"] = rs[1].detach().cpu()
                    sd[base + "num_batches_tracked"] = torch.tensor(rs[2], dtype=torch.long)
            return sd

        torch.save(state_dict(self.G), os.path.join(d, "%s_net_G0.pth" % epoch_label))
        torch.save(state_dict(self.D), os.path.join(d, "%s_net_D.pth" % epoch_label))
        if self.Df is not None:
            torch.save(state_dict(self.Df), os.path.join(d, "%s_net_D_f.pth" % epoch_label))
        for sc, dt in enumerate(self.DT):
            torch.save(state_dict(dt), os.path.join(d, "%s_net_D_T%d.pth" % (epoch_label, sc)))


def run_train(opt, steps=None):
    """train.py main: one process per GPU under torchrun (the reference used nn.DataParallel threads).

    Epoch structure of upstream train.py [RECALL]: `niter` epochs at the initial learning rate + `niter_decay` epochs
    of linear decay to zero; an epoch is one clip per training sequence (batchSize = number of ranks clips per
    iteration); `latest` checkpoints every `save_latest_freq` samples and at every `save_epoch_freq`-th epoch end (also
    under the epoch number), with `iter.txt` = (epoch, samples done in it) beside them; `--continue_train` reloads
    the `which_epoch` nets, reads iter.txt and resumes there (fresh Adam moments: upstream saves none).
    `steps` caps the total number of iterations (tests, benches)."""
    import os
    import time
    import numpy as np
    from . import distributed as Dm
    if "WORLD_SIZE" in os.environ:
        rank, local_rank, world = Dm.init_from_env()
    else:       # a plain single-device run computes on --gpu_ids[0], as the reference's single-GPU recipes do
        rank, local_rank, world = 0, (opt.gpu_ids[0] if getattr(opt, "gpu_ids", None) else 0), 1
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    trainer = Vid2VidTrainer(opt, dev)
    Dm.rendezvous("trainer built")      # (before the first gradient exchange: a rank that died building its replica is named)
    if rank == 0 and getattr(opt, "batchSize", world) not in (1, world):
        print("warning: --batchSize %d, but the batch is one clip per rank = %d (list %d devices in --gpu_ids)"
              % (opt.batchSize, world, opt.batchSize), flush=True)
    F_ = opt.max_frames_per_gpu
    synthetic = getattr(opt, "synthetic_data", False)
    ds = None
    if not synthetic:
        from .pose_dataset import TrainPoseDataset
        ds = TrainPoseDataset(opt, seed=1000 + rank)
    per_epoch = 1 if synthetic else max(1, len(ds) // world)          # iterations per epoch (world clips each)
    iter_path = os.path.join(opt.checkpoints_dir, opt.name, "iter.txt")
    start_epoch, epoch_iter = 1, 0
    if opt.continue_train:
        trainer.load(opt.which_epoch)
        try:
            start_epoch, epoch_iter = [int(v) for v in np.loadtxt(iter_path, delimiter=",", dtype=int)]
        except (OSError, ValueError):
            start_epoch, epoch_iter = 1, 0
        if rank == 0:
            print("Resuming from epoch %d at iteration %d" % (start_epoch, epoch_iter), flush=True)
        if start_epoch > opt.niter:
            trainer.update_learning_rate(start_epoch - 1)
        if ds is not None and start_epoch > opt.niter_step:
            ds.update_training_batch((start_epoch - 1) // max(1, opt.niter_step))

    def checkpoint(label, epoch, done):
        if rank == 0:
            trainer.save(label, (epoch, done))

    rng = np.random.default_rng(100 + rank)          # synthetic data: every rank has its own sequence
    tG = opt.n_frames_G
    last_pos = (start_epoch, epoch_iter)
    stats, it, total_samples = [], 0, (start_epoch - 1) * per_epoch * world + epoch_iter
    last_epoch = opt.niter + opt.niter_decay
    for epoch in range(start_epoch, last_epoch + 1):
        for k in range(epoch_iter // world, per_epoch):
            if steps is not None and it >= steps:
                break
            ts = time.perf_counter()
            if synthetic:
                H = W = opt.fineSize
                pose_np = np.where(rng.random((F_, H, W, 1)) < 0.02, rng.uniform(-1, 1, (F_, H, W, 9)), -1.0).astype(np.float32)
                pose = torch.zeros(F_, H, W, 12, device=dev)
                pose[..., :9] = torch.from_numpy(pose_np).to(dev)
                real = torch.zeros(F_, H, W, 4, device=dev)
                real[..., :3] = torch.tanh(torch.from_numpy(rng.standard_normal((F_, H, W, 3)).astype(np.float32))).to(dev)
                side = max(8, opt.fineSize // 32 * 8)
                boxes = [(H // 8, H // 8 + side, (W - side) // 2, (W - side) // 2 + side)] * F_ if opt.add_face_disc else None
                real_prev = None
                if not trainer.spec.no_flow:    # the real frame before each frame (flow / warp losses)
                    real_prev = torch.cat([torch.tanh(torch.from_numpy(rng.standard_normal((1, H, W, 4)).astype(np.float32))).to(dev),
                                           real[:-1]], 0)
                    real_prev[..., 3] = 0
                losses, _ = trainer.train_step(pose, real, boxes, real_prev=real_prev)
                what = ""
            else:
                # one clip per rank per iteration, walked in chunks of max_frames_per_gpu frames with the generated
                # frames carried over (detached) from chunk to chunk, one optimiser step per chunk -- upstream's
                # truncated recurrence
                clip = ds.sample(((epoch - 1) * per_epoch + k) * world + rank)
                A = torch.from_numpy(clip["A"]).to(dev)                      # [T,H,W,3] uint8
                B = torch.from_numpy(clip["B"]).to(dev)
                T_, H, W = A.shape[0], A.shape[1], A.shape[2]
                if world > 1:   # every rank must run the same number of chunks (one gradient all-reduce per chunk)
                    import torch.distributed as dist
                    tmin = torch.tensor([T_], device=dev)
                    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
                    T_ = int(tmin.item())
                prev = None
                for c0 in range(tG - 1, T_, F_):
                    fr = list(range(c0, min(c0 + F_, T_)))
                    pose = torch.zeros(len(fr), H, W, 12, device=dev)
                    for j, t in enumerate(fr):
                        for f in range(tG):                                    # window: oldest frame first
                            ops.pose_u8_to_f32(A[t - tG + 1 + f], pose[j], 3 * f)
                    real = torch.zeros(len(fr), H, W, 4, device=dev)
                    real[..., :3] = (B[fr].float() / 255.0 - 0.5) / 0.5
                    real_prev = None
                    if not trainer.spec.no_flow:
                        real_prev = torch.zeros(len(fr), H, W, 4, device=dev)
                        real_prev[..., :3] = (B[[t - 1 for t in fr]].float() / 255.0 - 0.5) / 0.5
                    boxes = None
                    if opt.add_face_disc:      # one region for the chunk (upstream boxes the whole batch of frames)
                        box = get_face_region(clip["A"][fr], opt.fineSize)
                        if world > 1:
                            # the face terms decide which parameters (D_f) have gradients: every rank must take the same
                            # branch, or the ranks' gradient buckets would differ in size and order -- the face
                            # discriminator trains on a chunk only when EVERY rank's chunk shows a face
                            import torch.distributed as dist
                            has = torch.tensor([1 if box is not None else 0], device=dev if dist.get_backend() == "nccl" else "cpu")
                            dist.all_reduce(has, op=dist.ReduceOp.MIN)
                            if int(has.item()) == 0:
                                box = None
                        boxes = [box] * len(fr) if box is not None else None
                    losses, prev = trainer.train_step(pose, real, boxes, prev, real_prev=real_prev)
                what = ", seq %s, %d frames %dx%d step %d" % (clip["seq"], T_ - tG + 1, H, W, clip["t_step"])
            torch.cuda.synchronize()
            stats.append(time.perf_counter() - ts)
            it += 1
            total_samples += world
            if rank == 0 and ((it - 1) % max(1, opt.print_freq // 100) == 0 or it == steps):
                print("(iter %d, epoch %d%s, %.0f ms, all-reduce %.1f MB%s) %s"
                      % (it - 1, epoch, what, 1e3 * stats[-1], trainer.comm_bytes / 2**20,
                         ", %.1f ms exposed" % trainer.comm_ms if trainer.time_comm else "",
                         " ".join("%s: %.3f" % kv for kv in losses.items())), flush=True)
            last_pos = (epoch, (k + 1) * world)
            if total_samples % max(1, opt.save_latest_freq) < world:
                checkpoint("latest", epoch, (k + 1) * world)
        else:
            epoch_iter = 0
            last_pos = (epoch + 1, 0)
            if epoch % max(1, opt.save_epoch_freq) == 0:
                checkpoint("latest", epoch + 1, 0)
                checkpoint(str(epoch), epoch + 1, 0)
            if epoch > opt.niter:
                trainer.update_learning_rate(epoch)
            if ds is not None and epoch % max(1, opt.niter_step) == 0:
                ds.update_training_batch(epoch // max(1, opt.niter_step))
            continue
        break   # the step cap was reached
    if world > 1:
        # the replicas started from the same seed and applied the same averaged gradients: their weights must still be
        # equal -- a cheap end-of-run check that the exchange reached every parameter on every rank
        import torch.distributed as dist
        chk = torch.stack([p.detach().double().sum() for p in trainer.optG.params + trainer.optD.params]).sum().reshape(1)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        if any(float(c) != float(allc[0]) for c in allc):
            raise RuntimeError("data-parallel replicas diverged: parameter checksums %s" % [float(c) for c in allc])
        if rank == 0:
            print("replicas in sync after %d steps (parameter checksum %.9g on all %d ranks)" % (it, float(allc[0]), world),
                  flush=True)
    if rank == 0:      # the final (or step-capped) save carries its position too
        trainer.save("latest", last_pos)
    if world > 1:
        Dm.leave_group()
    return {"ms_per_step": 1e3 * float(np.median(stats)) if stats else 0.0, "steps": it, "world": world}
