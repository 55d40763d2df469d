"""Discriminators of the vid2vid train step on the HIP path -- forward pass (SURVEY.md section 8a
rows a15-a17, App. A.3): NLayerDiscriminator (4x4 PatchGAN, LeakyReLU 0.2, getIntermFeat),
MultiscaleDiscriminator (num_D scales, AvgPool 3/2/1 without pad count), LSGAN and feature-matching
losses.  State-dict key names are upstream's (`scale{i}_layer{j}.{0,1}.*`).

BatchNorm2d (the upstream default --norm batch) runs in train mode, i.e. with statistics over the
whole batch: each image's conv writes its per-tile partials and one finalize merges all of them.
The backward pass / optimiser loop is not part of this round.
"""
import torch

from . import ops


class HipNLayerDiscriminator:
    def __init__(self, input_nc, ndf=64, n_layers=3, norm="batch", device="cuda"):
        self.device = torch.device(device)
        self.n_layers = n_layers
        self.norm = norm
        chans = [(input_nc, ndf, 2, False)]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            chans.append((nf_prev, nf, 2, True))
        nf_prev, nf = nf, min(nf * 2, 512)
        chans.append((nf_prev, nf, 1, True))
        chans.append((nf, 1, 1, False))
        self.layers = chans       # (Cin, Cout, stride, has_norm)
        self.params = None

    def load_state_dict(self, sd, prefix="model"):
        """keys: f"{prefix}{j}.0.weight/.bias" (conv) and f"{prefix}{j}.1.weight/.bias" (norm affine)."""
        self.raw = []
        for j, (cin, cout, stride, has_norm) in enumerate(self.layers):
            w = sd["%s%d.0.weight" % (prefix, j)].to(self.device, torch.float32).contiguous()
            b = sd["%s%d.0.bias" % (prefix, j)].to(self.device, torch.float32).contiguous()
            g = bt = None
            if has_norm and self.norm == "batch":
                g = sd["%s%d.1.weight" % (prefix, j)].to(self.device, torch.float32).contiguous()
                bt = sd["%s%d.1.bias" % (prefix, j)].to(self.device, torch.float32).contiguous()
            self.raw.append((w, b, g, bt))
        self._packed = {}
        return self

    def _conv(self, j, H, W, act, act_scale=1.0):
        cin, cout, stride, _ = self.layers[j]
        key = (j, H, W)
        if key not in self._packed:
            desc = ops.conv_desc(H, W, cin, cout, 4, stride, 2, ops.PAD_ZERO, False, act, act_scale)
            self._packed[key] = (desc, ops.pack_conv_weight(self.raw[j][0], desc, ops.round_up(cin, 4)))
        return self._packed[key]

    def forward(self, x):
        """x: [B, H, W, cs] NHWC batch.  Returns the list of the n_layers+2 stage outputs, each
        [B, h, w, C] (C rounded up to 4 for the 1-channel logits: channel 0 is the logit)."""
        B = x.shape[0]
        feats = []
        cur = x
        for j, (cin, cout, stride, has_norm) in enumerate(self.layers):
            H, W = cur.shape[1], cur.shape[2]
            last = j == len(self.layers) - 1
            act = ops.ACT_NONE if (has_norm or last) else ops.ACT_LRELU
            desc, pw = self._conv(j, H, W, act, 0.2)
            ho, wo = ops.conv_out_dims(desc)
            ycs = ops.round_up(cout, 4)
            y = torch.empty(B, ho, wo, ycs, dtype=torch.float32, device=self.device)
            w, b, g, bt = self.raw[j]
            if has_norm:
                n = ops.conv_stats_buffer(desc, self.device).numel()
                stats = torch.empty(B * n, dtype=torch.float32, device=self.device)
                for i in range(B):
                    ops.conv2d(cur[i], pw, b, desc, y_cs=ycs, stats=stats[i * n:(i + 1) * n], out=y[i])
                if self.norm == "batch":
                    mr = ops.batch_norm_finalize(stats, desc, B)
                    ops.instance_norm_apply(y, mr, g, bt, relu=2, out=y)
                else:
                    for i in range(B):
                        mr = ops.instance_norm_finalize(stats[i * n:(i + 1) * n], desc)
                        ops.instance_norm_apply(y[i], mr, relu=2, out=y[i])
            else:
                for i in range(B):
                    ops.conv2d(cur[i], pw, b, desc, y_cs=ycs, out=y[i])
            feats.append(y)
            cur = y
        return feats


class HipMultiscaleDiscriminator:
    """num_D PatchGANs on an AvgPool pyramid; result[i] = stage outputs of the i-th finest scale."""

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=2, norm="batch", device="cuda"):
        self.num_D = num_D
        self.n_layers = n_layers
        self.nets = [HipNLayerDiscriminator(input_nc, min(ndf * 2 ** (num_D - 1 - i), 64), n_layers, norm, device)
                     for i in range(num_D)]

    def load_state_dict(self, sd):
        for i, net in enumerate(self.nets):
            net.load_state_dict(sd, prefix="scale%d_layer" % i)
        return self

    def forward(self, x):
        result = []
        for i in range(self.num_D):
            result.append(self.nets[self.num_D - 1 - i].forward(x))
            if i != self.num_D - 1:
                x = torch.stack([ops.avgpool3x3s2(x[b]) for b in range(x.shape[0])])
        return result


def gan_loss(pred_scales, target_is_real):
    """LSGAN (`--no_lsgan` unset): sum over scales of MSELoss(last stage, 1|0), 'elementwise_mean'
    ($SP/torch/nn/modules/loss.py:12,361).  pred_scales: output of HipMultiscaleDiscriminator.forward."""
    total = None
    for stages in pred_scales:
        logits = stages[-1][..., 0].contiguous()
        term = ops.sum_sq_diff_const(logits, 1.0 if target_is_real else 0.0) / logits.numel()
        total = term if total is None else total + term
    return total


def feature_matching_loss(pred_fake, pred_real, n_layers=3, lambda_feat=10.0):
    """sum_scales sum_stages (1/num_D) * (4/(n_layers+1)) * L1(fake_j, real_j) * lambda_feat
    (SURVEY App. A.3).  All stages but the last (the logits) take part."""
    num_D = len(pred_fake)
    total = None
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            f, r = pred_fake[i][j], pred_real[i][j]
            term = ops.sum_abs_diff(f, r) / f.numel() * ((1.0 / num_D) * (4.0 / (n_layers + 1)) * lambda_feat)
            total = term if total is None else total + term
    return total
