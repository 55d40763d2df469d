"""Per-parameter gradient error of the (face) discriminator against the oracle in fp64, next to the fp32 oracle's own: where
does the HIP path lose accuracy on small maps?  Usage: df_grad_probe.py [size ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle.generator_ref import MultiscaleDiscriminator, weights_init
from text2video_amd import train as T

torch.manual_seed(0)
for size in [int(a) for a in sys.argv[1:]] or [128, 64, 256]:
    for B in (2, 1):
        ref = MultiscaleDiscriminator(6, 64, 3, 1, "batch").train()
        gen = torch.Generator().manual_seed(7)
        ref.apply(lambda m: weights_init(m, gen))
        dsd = {k: v.clone() for k, v in ref.state_dict().items() if "running" not in k and "num_batches" not in k}
        x = torch.tanh(torch.randn(B, 6, size, size, generator=torch.Generator().manual_seed(3)))
        mse = torch.nn.MSELoss()

        def run(dtype):
            net = MultiscaleDiscriminator(6, 64, 3, 1, "batch").train()
            net.load_state_dict(ref.state_dict())
            net = net.to(dtype)
            p = net(x.to(dtype))
            loss = sum(mse(q[-1], torch.ones_like(q[-1])) for q in p) + sum(q[j].abs().mean() for q in p for j in range(4))
            g = torch.autograd.grad(loss, list(net.parameters()))
            return float(loss), {k: v for (k, _), v in zip(net.named_parameters(), g)}
        l64, g64 = run(torch.float64)
        l32, g32 = run(torch.float32)
        Dh = T.TrainableDiscriminator(6, dsd, 64, 3, 1, "batch", "cuda:0")
        xh = torch.zeros(B, size, size, 8, device="cuda:0")
        xh[..., :6] = x.permute(0, 2, 3, 1).cuda()
        ph = Dh(xh)
        lossh = T.gan_loss(ph, True) + sum(T._L1.apply(q[j], torch.zeros_like(q[j]), q[j].numel()) for q in ph for j in range(4))
        gh = torch.autograd.grad(lossh, list(Dh.parameters()), allow_unused=True)
        gh = {k: v for (k, _), v in zip(Dh.named_upstream_parameters().items(), gh)}
        print("size %d batch %d: loss fp64 %.6f  HIP %+.1e  fp32 %+.1e" % (size, B, l64, float(lossh) - l64, l32 - l64))
        for k, r in g64.items():
            s = r.abs().max().item()
            if s < 1e-9:
                continue
            print("   %-26s HIP %.1e   fp32 oracle %.1e" % (k, (gh[k].cpu().double() - r).abs().max().item() / s,
                                                          (g32[k].double() - r).abs().max().item() / s))
