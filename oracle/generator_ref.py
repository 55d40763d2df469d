"""CPU restatement (stock torch.nn, fp32) of the vid2vid generators/discriminators.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED: architecture follows
SURVEY.md Appendix A.1-A.3 (recollection of upstream NVIDIA/vid2vid models/networks.py, which
the reference only links to: /root/reference/README.md:18,246).  Operator semantics are pinned
to the torch-0.4.1 sources vendored in the reference ($SP = venv_vid2vid/lib/python3.7/
site-packages):

* reflection pad without edge repeat ............ $SP/torch/nn/modules/padding.py:217-255
* Conv2d weight [Cout,Cin,kH,kW] ................ $SP/torch/nn/modules/conv.py:28-47,289-300
* ConvTranspose2d weight [Cin,Cout,kH,kW], output_padding ... conv.py:29-30,676-691
* InstanceNorm2d: eps 1e-5, biased var, affine=False, no running stats
                                                  $SP/torch/nn/modules/instancenorm.py:6-7,44-49
  (= batch_norm(training=True) on [1,B*C,H,W] ... $SP/torch/nn/functional.py:1258-1301)
* BatchNorm2d(affine) in train mode, N=1 == IN + affine
* grid_sample: bilinear, corner-aligned ("x=-1 left-top pixel, x=1 right-bottom pixel"),
  padding 'border' ................................ $SP/torch/nn/functional.py:2046-2093
  -> here: align_corners=True made explicit (modern default differs)
* AvgPool2d(3, 2, 1, count_include_pad=False) ..... $SP/torch/nn/modules/pooling.py:536-543

State-dict key names follow nn.Sequential indices of upstream so that real
latest_net_G0.pth checkpoints load unchanged (SURVEY App. A.1 last paragraph).
"""
import copy
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F


def get_norm_layer(norm_type="batch"):
    # upstream default --norm batch -> BatchNorm2d(affine=True); 'instance' -> affine=False
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False)
    raise NotImplementedError(norm_type)


class ResnetBlock(nn.Module):
    """x + [ReflPad1, Conv3, N, ReLU, ReflPad1, Conv3, N](x)   (SURVEY App. A `RB(c)`)."""

    def __init__(self, dim, norm_layer, activation=None):
        super().__init__()
        activation = activation or nn.ReLU(True)
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0), norm_layer(dim), activation,
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0), norm_layer(dim))

    def forward(self, x):
        return x + self.conv_block(x)


def _c7(i, o):
    return [nn.ReflectionPad2d(3), nn.Conv2d(i, o, 7, padding=0)]


def resample(image, flow):
    """SURVEY App. A.1 `resample`: base grid linspace(-1,1) + flow/((W-1)/2,(H-1)/2), then grid_sample(bilinear,
    border), corner aligned -- written out the way torch 0.4.1's SpatialGridSamplerBilinear kernel evaluates it
    (declared $SP/torch/lib/THCUNN.h:1048-1062, reached from $SP/torch/nn/functional.py:2046-2093; the .cu body is
    not vendored: [RECALL]): un-normalise ix = ((x+1)/2)*(W-1); the four corners are floor(ix), floor(ix)+1 (same
    in y); bilinear weights from the UNCLIPPED position; for padding 'border' the corner INDICES are clipped into
    the image.  Autograd through this expression gives exactly that kernel's updateGradInput: gradGrid from the
    (clipped) corner values times the weight derivatives -- zero with respect to a coordinate that lies outside
    the image or exactly on the last row / column (both corners clip to the same pixel).  Modern
    F.grid_sample(align_corners=True, padding_mode='border') returns the same forward values and the same
    gradients strictly inside the image (tests/test_cpu_oracle_and_host.py pins both).  The two differ only on a
    set of measure zero: a coordinate exactly on the border (modern torch reports a zero gradient on the first AND
    the last row / column, this rule only on the last) or exactly on an integer position, where the bilinear
    interpolant has a kink and either one-sided derivative is "the" gradient."""
    b, c, h, w = image.shape
    # (on the image's device and in its dtype: the tests also evaluate the oracle on the GPU and in fp64)
    hor = torch.linspace(-1.0, 1.0, w, dtype=image.dtype, device=image.device).view(1, 1, 1, w).expand(b, -1, h, -1)
    ver = torch.linspace(-1.0, 1.0, h, dtype=image.dtype, device=image.device).view(1, 1, h, 1).expand(b, -1, -1, w)
    grid = torch.cat([hor, ver], 1)
    flow = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    final = grid + flow
    ix = ((final[:, 0] + 1.0) / 2.0) * (w - 1)          # [b,h,w]
    iy = ((final[:, 1] + 1.0) / 2.0) * (h - 1)
    # Outside the image both clipped corners of an axis are the same pixel, so the value is that pixel's and the
    # gradient with respect to that coordinate vanishes: clamping the coordinate's value (gradient cut where it
    # was clamped) and taking corners floor, min(floor+1, last) selects the same taps with the same weights.
    ixd, iyd = ix.detach(), iy.detach()
    ixc, iyc = ixd.clamp(0, w - 1), iyd.clamp(0, h - 1)
    x0, y0 = ixc.floor(), iyc.floor()
    x1, y1 = (x0 + 1).clamp(max=w - 1), (y0 + 1).clamp(max=h - 1)
    inx = ((ixd >= 0) & (ixd <= w - 1)).to(ix.dtype)
    iny = ((iyd >= 0) & (iyd <= h - 1)).to(iy.dtype)
    fx = (ixc - x0) + (ix - ixd) * inx
    fy = (iyc - y0) + (iy - iyd) * iny
    flat = image.reshape(b, c, h * w)

    def tap(yy, xx):
        idx = (yy * w + xx).long().reshape(b, 1, h * w).expand(-1, c, -1)
        return flat.gather(2, idx).reshape(b, c, h, w)

    wx1, wy1 = fx.unsqueeze(1), fy.unsqueeze(1)
    wx0, wy0 = 1.0 - wx1, 1.0 - wy1
    return tap(y0, x0) * (wx0 * wy0) + tap(y0, x1) * (wx1 * wy0) + tap(y1, x0) * (wx0 * wy1) + tap(y1, x1) * (wx1 * wy1)


def resample_modern(image, flow):
    """the same operation through modern torch's F.grid_sample (cross-check of `resample`)"""
    b, c, h, w = image.shape
    hor = torch.linspace(-1.0, 1.0, w, dtype=image.dtype, device=image.device).view(1, 1, 1, w).expand(b, -1, h, -1)
    ver = torch.linspace(-1.0, 1.0, h, dtype=image.dtype, device=image.device).view(1, 1, h, 1).expand(b, -1, -1, w)
    grid = torch.cat([hor, ver], 1)
    flow = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    final = (grid + flow).permute(0, 2, 3, 1)
    return F.grid_sample(image, final, mode="bilinear", padding_mode="border", align_corners=True)


class CompositeGenerator(nn.Module):
    """netG0."""

    def __init__(self, input_nc, output_nc, prev_output_nc, ngf=128, n_downsampling=3, n_blocks=9,
                 no_flow=False, norm="batch"):
        super().__init__()
        norm_layer = get_norm_layer(norm)
        act = nn.ReLU(True)
        self.no_flow = no_flow
        self.n_downsampling = n_downsampling
        down = _c7(input_nc, ngf) + [norm_layer(ngf), act]
        for i in range(n_downsampling):
            m = 2 ** i
            down += [nn.Conv2d(ngf * m, ngf * m * 2, 3, stride=2, padding=1), norm_layer(ngf * m * 2), act]
        m = 2 ** n_downsampling
        for _ in range(n_blocks - n_blocks // 2):
            down += [ResnetBlock(ngf * m, norm_layer, act)]
        down_img = _c7(prev_output_nc, ngf) + [norm_layer(ngf), act]
        down_img += copy.deepcopy(down[4:])
        res = [ResnetBlock(ngf * m, norm_layer, act) for _ in range(n_blocks // 2)]
        up = []
        for i in range(n_downsampling):
            mm = 2 ** (n_downsampling - i)
            up += [nn.ConvTranspose2d(ngf * mm, ngf * mm // 2, 3, stride=2, padding=1, output_padding=1),
                   norm_layer(ngf * mm // 2), act]
        self.model_down_seg = nn.Sequential(*down)
        self.model_down_img = nn.Sequential(*down_img)
        self.model_res_img = nn.Sequential(*res)
        self.model_up_img = nn.Sequential(*up)
        self.model_final_img = nn.Sequential(*(_c7(ngf, output_nc) + [nn.Tanh()]))
        if not no_flow:
            self.model_res_flow = nn.Sequential(*copy.deepcopy(res))
            self.model_up_flow = nn.Sequential(*copy.deepcopy(up))
            self.model_final_flow = nn.Sequential(*_c7(ngf, 2))
            self.model_final_w = nn.Sequential(*(_c7(ngf, 1) + [nn.Sigmoid()]))

    def forward(self, x, img_prev, use_raw_only=False):
        d = self.model_down_seg(x) + self.model_down_img(img_prev)
        img_feat = self.model_up_img(self.model_res_img(d))
        img_raw = self.model_final_img(img_feat)
        flow = weight = flow_feat = None
        if not self.no_flow:
            flow_feat = self.model_up_flow(self.model_res_flow(d))
            flow = self.model_final_flow(flow_feat) * 20
            weight = self.model_final_w(flow_feat)
        if use_raw_only or self.no_flow:
            img_final = img_raw
        else:
            img_warp = resample(img_prev[:, -3:], flow)
            img_final = img_raw * weight + img_warp * (1 - weight)
        return img_final, flow, weight, img_raw, img_feat, flow_feat


class CompositeLocalGenerator(nn.Module):
    """netG{s>=1}: fine-scale enhancer fed by the coarse scale's features."""

    def __init__(self, input_nc, output_nc, prev_output_nc, ngf_global=128, n_blocks_local=3, scale=1,
                 no_flow=False, norm="batch"):
        super().__init__()
        norm_layer = get_norm_layer(norm)
        act = nn.ReLU(True)
        self.no_flow = no_flow
        self.scale = scale
        ngf = ngf_global // (2 ** scale)
        self.model_down_seg = nn.Sequential(*(_c7(input_nc, ngf) + [norm_layer(ngf), act,
                                              nn.Conv2d(ngf, ngf * 2, 3, stride=2, padding=1), norm_layer(ngf * 2), act]))
        self.model_down_img = nn.Sequential(*(_c7(prev_output_nc, ngf) + [norm_layer(ngf), act,
                                              nn.Conv2d(ngf, ngf * 2, 3, stride=2, padding=1), norm_layer(ngf * 2), act]))
        up = [ResnetBlock(ngf * 2, norm_layer, act) for _ in range(n_blocks_local)]
        up += [nn.ConvTranspose2d(ngf * 2, ngf, 3, stride=2, padding=1, output_padding=1), norm_layer(ngf), act]
        self.model_up_img = nn.Sequential(*up)
        self.model_final_img = nn.Sequential(*(_c7(ngf, output_nc) + [nn.Tanh()]))
        if not no_flow:
            self.model_up_flow = nn.Sequential(*copy.deepcopy(up))
            self.model_final_flow = nn.Sequential(*_c7(ngf, 2))
            self.model_final_w = nn.Sequential(*(_c7(ngf, 1) + [nn.Sigmoid()]))

    def forward(self, x, img_prev, img_feat_coarse, flow_feat_coarse, use_raw_only=False):
        flow_multiplier = 20 * (2 ** self.scale)
        d = self.model_down_seg(x) + self.model_down_img(img_prev)
        img_feat = self.model_up_img(d + img_feat_coarse)
        img_raw = self.model_final_img(img_feat)
        flow = weight = flow_feat = None
        if not self.no_flow:
            flow_feat = self.model_up_flow(d + flow_feat_coarse)
            flow = self.model_final_flow(flow_feat) * flow_multiplier
            weight = self.model_final_w(flow_feat)
        if use_raw_only or self.no_flow:
            img_final = img_raw
        else:
            img_warp = resample(img_prev[:, -3:], flow)
            img_final = img_raw * weight + img_warp * (1 - weight)
        return img_final, flow, weight, img_raw, img_feat, flow_feat


class NLayerDiscriminator(nn.Module):
    """PatchGAN: kw=4, padw=2, LeakyReLU(0.2), returns all stage outputs."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm="batch"):
        super().__init__()
        norm_layer = get_norm_layer(norm)
        self.n_layers = n_layers
        kw, padw = 4, 2
        seq = [[nn.Conv2d(input_nc, ndf, kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq += [[nn.Conv2d(nf_prev, nf, kw, stride=2, padding=padw), norm_layer(nf), nn.LeakyReLU(0.2, True)]]
        nf_prev, nf = nf, min(nf * 2, 512)
        seq += [[nn.Conv2d(nf_prev, nf, kw, stride=1, padding=padw), norm_layer(nf), nn.LeakyReLU(0.2, True)]]
        seq += [[nn.Conv2d(nf, 1, kw, stride=1, padding=padw)]]
        for n, s in enumerate(seq):
            setattr(self, "model" + str(n), nn.Sequential(*s))

    def forward(self, x):
        res = [x]
        for n in range(self.n_layers + 2):
            res.append(getattr(self, "model" + str(n))(res[-1]))
        return res[1:]


class MultiscaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=2, norm="batch"):
        super().__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, min(ndf * 2 ** (num_D - 1 - i), 64), n_layers, norm)
            for j in range(n_layers + 2):
                setattr(self, "scale" + str(i) + "_layer" + str(j), getattr(netD, "model" + str(j)))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x):
        result = []
        for i in range(self.num_D):
            model = [getattr(self, "scale" + str(self.num_D - 1 - i) + "_layer" + str(j))
                     for j in range(self.n_layers + 2)]
            r = [x]
            for m in model:
                r.append(m(r[-1]))
            result.append(r[1:])
            if i != self.num_D - 1:
                x = self.downsample(x)
        return result


class VGG19Features(nn.Module):
    """torchvision `vgg19().features` up to relu5_1 ($SP/torchvision/models/vgg.py:82 cfg 'E': 64 64 M 128 128 M
    256x4 M 512x4 M 512...; 3x3 convs, padding 1, ReLU; MaxPool2d(2,2)), sliced at relu1_1, relu2_1, relu3_1,
    relu4_1, relu5_1 = features[1], [6], [11], [20], [29] [RECALL upstream models/networks.py Vgg19].  Parameter
    names follow torchvision (features.N.weight / .bias) so its state dict loads directly."""

    CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
    TAPS = (1, 6, 11, 20, 29)

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in self.CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
                cin = v
        self.features = nn.Sequential(*layers)

    def forward(self, x):
        out = []
        for i, m in enumerate(self.features):
            x = m(x)
            if i in self.TAPS:
                out.append(x)
        return out


def vgg_loss_ref(vgg, fake, real):
    """VGGLoss [RECALL upstream models/networks.py]: sum_i w_i * L1(vgg(x)_i, vgg(y)_i.detach()), weights
    1/32, 1/16, 1/8, 1/4, 1; no input normalisation; 2x average-pool while wider than 1024."""
    weights = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)
    while fake.shape[3] > 1024:
        fake = F.avg_pool2d(fake, 3, 2, 1, count_include_pad=False)
        real = F.avg_pool2d(real, 3, 2, 1, count_include_pad=False)
    fx, fy = vgg(fake), vgg(real)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(weights, fx, fy))


def weights_init(m, gen):
    """vid2vid `weights_init` [RECALL]: Conv* weight ~ N(0,0.02); BatchNorm2d weight ~ N(1,0.02), bias 0."""
    name = m.__class__.__name__
    if name.find("Conv") != -1 and hasattr(m, "weight") and m.weight is not None and m.weight.dim() == 4:
        m.weight.data.normal_(0.0, 0.02, generator=gen)
    elif name.find("BatchNorm2d") != -1:
        m.weight.data.normal_(1.0, 0.02, generator=gen)
        m.bias.data.fill_(0)


class Vid2VidInferenceRef:
    """`Vid2VidModelG.inference` / `generate_frame_infer` on the CPU, fp32.

    * --no_first_img: first frame of a sequence uses zeros for the tG-1 previous outputs and
      `use_raw_only=True` (reference flag: /root/reference/text2video_audio.sh:42).
    * 2-deep FIFO of generated frames; reset by `reset()` when the dataset reports change_seq.
    * n_scales_spatial>1: AvgPool(3,2,1,count_include_pad=False) pyramid, coarse->fine.
    """

    def __init__(self, nets, n_frames_G=3, output_nc=3, no_first_img=True):
        self.nets = nets  # [netG0, netG1, ...]
        self.n_scales = len(nets)
        self.tG = n_frames_G
        self.output_nc = output_nc
        self.no_first_img = no_first_img
        self.fake_B_prev = None
        self.pool = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
        # train-mode BN on purpose (upstream never calls .eval() on G at test time)
        for n in self.nets:
            n.train()

    def reset(self):
        self.fake_B_prev = None

    def _pyr(self, t):
        out = [t]
        for _ in range(1, self.n_scales):
            s = out[-1].shape
            out.append(self.pool(out[-1].reshape(-1, s[-3], s[-2], s[-1])).reshape(*s[:-2], s[-2] // 2, s[-1] // 2))
        return out

    @torch.no_grad()
    def inference(self, A):
        """A: [1, tG, 3, H, W] in [-1,1].  Returns fake_B [1,3,H,W] at the finest scale."""
        _, tG, nc, H, W = A.shape
        first = self.fake_B_prev is None
        if first:
            assert self.no_first_img, "first-frame generator not part of the reference's flag set"
            z = torch.zeros(tG - 1, self.output_nc, H, W, dtype=A.dtype, device=A.device)
            self.fake_B_prev = self._pyr(z)
        real_A = self._pyr(A)
        use_raw_only = self.no_first_img and first
        img_feat = flow_feat = None
        fake_B = None
        for s in range(self.n_scales):
            si = self.n_scales - 1 - s
            a = real_A[si]
            h, w = a.shape[-2:]
            x = a[0, :tG].reshape(1, -1, h, w)
            p = self.fake_B_prev[si].reshape(1, -1, h, w)
            if s == 0:
                out = self.nets[0](x, p, use_raw_only)
            else:
                out = self.nets[s](x, p, img_feat, flow_feat, use_raw_only)
            fake_B, img_feat, flow_feat = out[0], out[4], out[5]
            self.fake_B_prev[si] = torch.cat([self.fake_B_prev[si][1:], fake_B], 0)
        return fake_B
