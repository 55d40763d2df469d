"""Train step of the vid2vid pose model on the HIP path.

torch.autograd is used as the TAPE only: every differentiable op below is a `torch.autograd.Function`
whose forward and backward are calls into libt2v_hip.so (implicit-GEMM conv, fused norm statistics,
weight-gradient kernel, norm backward, ...).  Tensors are NHWC fp32 on the device; parameters are
torch-layout fp32 `nn.Parameter`s (upstream state-dict names) that are re-packed for the kernels each
step.  Channel concat / crop / detach / scalar loss arithmetic stay torch-native plumbing.

Scope (what the reference's training command README.md:171-176 exercises with --openpose_only):
generator (no flow branch), multiscale image discriminator (--num_D 2), face discriminator
(--add_face_disc), LSGAN + feature-matching losses, Adam(lr 2e-4, beta1 0.5), data-parallel gradient
all-reduce.  Not built: VGG and FlowNet2-based losses / temporal discriminator (external weights that
are not in the reference tree).
"""
import torch

from . import ops
from .backward import ConvDataGrad
from .generator import GeneratorSpec, layer_keys, synthetic_state_dict  # noqa: F401


# ------------------------------------------------------------------------------------------------
# differentiable building blocks
# ------------------------------------------------------------------------------------------------
class _ConvBlock(torch.autograd.Function):
    """y = act(norm(conv(x) + b)) + res  on a batch [B,H,W,cs].

    norm: None | 'instance' (statistics per image) | 'batch' (statistics over the batch, BatchNorm2d in
    train mode); relu: 0 none / 1 ReLU / 2 LeakyReLU(0.2) applied after the norm; without a norm the
    activation `act` (ACT_NONE / ACT_TANH / ACT_LRELU) is fused into the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, res, desc, norm, relu, act, need_dx):
        B = x.shape[0]
        dev = x.device
        xcs = x.shape[-1]
        ho, wo = ops.conv_out_dims(desc)
        ycs = ops.round_up(desc.Cout, 4)
        fdesc = ops.conv_desc(desc.H, desc.W, desc.Cin, desc.Cout, desc.kH, desc.stride, desc.pad, desc.pad_mode,
                              bool(desc.transposed), act if norm is None else ops.ACT_NONE, 0.2, desc.output_padding)
        pw = ops.pack_conv_weight(w.detach().contiguous(), fdesc, xcs)
        c = torch.empty(B, ho, wo, ycs, dtype=torch.float32, device=dev)
        mrs = None
        if norm is None:
            for i in range(B):
                ops.conv2d(x[i], pw, b.detach(), fdesc, y_cs=ycs, out=c[i])
            y = c
        else:
            n = ops.conv_stats_buffer(fdesc, dev).numel()
            stats = torch.empty(B * n, dtype=torch.float32, device=dev)
            for i in range(B):
                ops.conv2d(x[i], pw, b.detach(), fdesc, y_cs=ycs, stats=stats[i * n:(i + 1) * n], out=c[i])
            y = torch.empty_like(c)
            g = gamma.detach() if gamma is not None else None
            bt = beta.detach() if beta is not None else None
            if norm == "batch":
                mrs = [ops.batch_norm_finalize(stats, fdesc, B)]
                ops.instance_norm_apply(c, mrs[0], g, bt, relu=relu, out=y)
            else:
                mrs = []
                for i in range(B):
                    mrs.append(ops.instance_norm_finalize(stats[i * n:(i + 1) * n], fdesc))
                    ops.instance_norm_apply(c[i], mrs[i], g, bt, relu=relu, out=y[i])
        if res is not None:
            y = y + res   # residual add (plumbing-level elementwise; its gradient is the identity)
        ctx.meta = (desc, fdesc, norm, relu, act, need_dx, mrs, gamma is not None)
        ctx.save_for_backward(x, w, c, gamma, beta, y if (norm is None and act != ops.ACT_NONE) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        desc, fdesc, norm, relu, act, need_dx, mrs, affine = ctx.meta
        x, w, c, gamma, beta, y_act = ctx.saved_tensors
        dy = dy.contiguous()
        B = x.shape[0]
        dgamma = dbeta = None
        if norm is None:
            dc = ops.act_backward(dy, y_act, act, 0.2) if act != ops.ACT_NONE else dy
        elif norm == "batch":
            dc, sums = ops.instance_norm_backward(c, dy, mrs[0], gamma, beta, relu)
            if affine:
                dbeta, dgamma = sums[:, 0].contiguous(), sums[:, 1].contiguous()
        else:
            dc = torch.empty_like(c)
            tot = None
            for i in range(B):
                d_i, s_i = ops.instance_norm_backward(c[i], dy[i], mrs[i], gamma, beta, relu)
                dc[i] = d_i
                tot = s_i if tot is None else tot + s_i
            if affine:
                dbeta, dgamma = tot[:, 0].contiguous(), tot[:, 1].contiguous()
        db = ops.channel_sum(dc, desc.Cout)
        dwp = ops.conv2d_backward_weight(x, dc, fdesc)
        dw = ops.unpack_conv_weight(dwp, fdesc, x.shape[-1])
        dx = None
        if need_dx:
            dg = ConvDataGrad(fdesc).refresh(w.detach())
            dx = torch.stack([dg(dc[i]) for i in range(B)])
            if dx.shape[-1] != x.shape[-1]:
                pad = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
                pad[..., :dx.shape[-1]] = dx
                dx = pad
        return dx, dw, db, dgamma, dbeta, (dy if ctx.needs_input_grad[5] else None), None, None, None, None, None


def conv_block(x, w, b, desc, gamma=None, beta=None, res=None, norm="instance", relu=1, act=ops.ACT_NONE,
               need_dx=True):
    return _ConvBlock.apply(x, w, b, gamma, beta, res, desc, norm, relu, act, need_dx)


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


# ------------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------------
class TrainableGenerator(torch.nn.Module):
    """CompositeGenerator without flow branch (what --openpose_only trains), upstream parameter names."""

    def __init__(self, spec, state_dict, device="cuda"):
        super().__init__()
        assert spec.no_flow and not spec.is_local, "train step: global generator without flow branch"
        self.spec = spec
        self.keys = layer_keys(spec)
        self.params = torch.nn.ParameterDict()
        for k, v in state_dict.items():
            self.params[k.replace(".", "/")] = torch.nn.Parameter(v.to(device, torch.float32).contiguous())

    def p(self, key):
        return self.params[key.replace(".", "/")]

    def named_upstream_parameters(self):
        return {k.replace("/", "."): v for k, v in self.params.items()}

    def forward(self, pose, prev):
        """pose [1,H,W,12], prev [1,H,W,8] NHWC -> fake [1,H,W,4] (tanh RGB in channels 0..2)."""
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

        nb_enc, nb_res = s.n_blocks - s.n_blocks // 2, s.n_blocks // 2
        d = encoder(pose, s.input_nc, nb_enc) + encoder(prev, s.prev_nc, nb_enc)
        h = resblocks(d, nb_res)
        for i in range(n):
            l = n - i
            h = cna(h, ops.conv_desc(H >> l, W >> l, G << l, G << (l - 1), 3, 2, 1, ops.PAD_ZERO, True))
        ck, _, _ = next(it)
        return conv_block(h, self.p(ck + ".weight"), self.p(ck + ".bias"),
                          ops.conv_desc(H, W, G, 3, 7, 1, 3, ops.PAD_REFLECT), norm=None, relu=0, act=ops.ACT_TANH)


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

    def p(self, key):
        return self.params[key.replace(".", "/")]

    def named_upstream_parameters(self):
        return {k.replace("/", "."): v for k, v in self.params.items()}

    def _single(self, x, i):
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
                cur = conv_block(cur, self.p(pre + ".0.weight"), self.p(pre + ".0.bias"), desc, g, b, None, self.norm, 2)
            else:
                cur = conv_block(cur, self.p(pre + ".0.weight"), self.p(pre + ".0.bias"), desc, norm=None, relu=0,
                                 act=ops.ACT_NONE if last else ops.ACT_LRELU)
            feats.append(cur)
        return feats

    def forward(self, x):
        """x [B,H,W,cs] -> result[i] = stage outputs of the i-th finest scale."""
        result = []
        for i in range(self.num_D):
            result.append(self._single(x, self.num_D - 1 - i))
            if i != self.num_D - 1:
                x = _AvgPool.apply(x)
        return result


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


# ------------------------------------------------------------------------------------------------
# optimiser + data-parallel gradient exchange
# ------------------------------------------------------------------------------------------------
class FusedAdam:
    """torch-0.4.1 Adam semantics ($SP/torch/optim/adam.py:48-98), one fused HIP kernel per tensor."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.step_no = lr, betas, eps, 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        self.step_no += 1
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is not None:
                ops.adam_step(p.data, p.grad.contiguous(), m, v, self.lr, self.betas[0], self.betas[1], self.eps,
                              self.step_no)


def allreduce_gradients(params, bucket_mb=64):
    """Data-parallel gradient exchange: bucketed all-reduce (RCCL over xGMI; gloo on CPU), averaged over
    ranks.  Replaces DataParallel's reduce-to-GPU-0 + re-broadcast: replicas are
    persistent, so one all-reduce per bucket is all the communication a step needs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    limit = bucket_mb * (1 << 20) // 4
    nbytes, bucket, size = 0, [], 0

    def flush(bucket):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat)
        flat /= world
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        return flat.numel() * 4

    for g in grads:
        if size + g.numel() > limit and bucket:
            nbytes += flush(bucket)
            bucket, size = [], 0
        bucket.append(g)
        size += g.numel()
    if bucket:
        nbytes += flush(bucket)
    return nbytes
