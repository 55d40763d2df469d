"""Time one train iteration pieces at config-5 size on one GPU: generator forward+backward for
`frames` frames (1 sequence per GPU, max_frames_per_gpu 2), discriminator losses, Adam."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text2video_amd import train as T
from text2video_amd.generator import GeneratorSpec, synthetic_state_dict
from oracle.generator_ref import MultiscaleDiscriminator, weights_init

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--ngf", type=int, default=128)
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--vgg", action="store_true", help="add the VGG19 perceptual loss (seeded random weights)")
args = ap.parse_args()
dev = "cuda:0"
H = W = args.size
spec = GeneratorSpec(ngf=args.ngf, n_downsample=3, n_blocks=9, no_flow=True, norm="batch")
G = T.TrainableGenerator(spec, synthetic_state_dict(spec, 1, "vid2vid"), dev)
Dr = MultiscaleDiscriminator(6, 64, 3, 2, "batch")
gen = torch.Generator().manual_seed(1); Dr.apply(lambda m: weights_init(m, gen))
D = T.TrainableDiscriminator(6, Dr.state_dict(), 64, 3, 2, "batch", dev)
optG, optD = T.FusedAdam(G.parameters()), T.FusedAdam(D.parameters())
vgg = T.HipVGG19Features(T.vgg19_random_state_dict(5), dev) if args.vgg else None
F = args.frames
pose = torch.randn(F, H, W, 12, device=dev).clamp(-1, 1); pose[..., 9:] = 0
real = torch.tanh(torch.randn(F, H, W, 4, device=dev)); real[..., 3] = 0
A = pose[..., 6:9]
z2 = torch.zeros(F, H, W, 2, device=dev)
def d_in(img): return torch.cat([A, img[..., :3], z2], -1).contiguous()
def step():
    with T.batched_weight_gradients(optG.params):
        return _step()
def _step():
    prev = torch.tanh(torch.randn(1, H, W, 8, device=dev)); prev[..., 6:] = 0
    fakes = []
    for f in range(F):
        fk = G(pose[f:f + 1], prev); fakes.append(fk)
        nprev = torch.zeros_like(prev); nprev[..., :3] = prev[..., 3:6]; nprev[..., 3:6] = fk.detach()[..., :3]; prev = nprev
    fake = torch.cat(fakes, 0)
    pr = D(d_in(real)); pfd = D(d_in(fake.detach()))
    loss_D = 0.5 * (T.gan_loss(pfd, False) + T.gan_loss(pr, True))
    pfg = D(d_in(fake))
    loss_G = T.gan_loss(pfg, True) + T.feature_matching_loss(pfg, pr)
    if vgg is not None:
        loss_G = loss_G + T.vgg_loss(vgg, fake, real) * 10.0
    optG.zero_grad(); optD.zero_grad()
    gG = torch.autograd.grad(loss_G, list(G.parameters()), retain_graph=True)
    gD = torch.autograd.grad(loss_D, list(D.parameters()))
    for p, g in zip(G.parameters(), gG): p.grad = g
    for p, g in zip(D.parameters(), gD): p.grad = g
    optG.step(); optD.step()
    return loss_G.item(), loss_D.item()
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters): lg, ld = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
print("train iteration %dx%d ngf%d, %d frames: %.1f ms/step (loss_G %.3f loss_D %.3f), peak mem %.1f GB"
      % (H, W, args.ngf, F, dt * 1e3, lg, ld, torch.cuda.max_memory_allocated() / 2**30))
