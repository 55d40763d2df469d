"""Golden taps of the VGG19 feature extractor, generated with the REFERENCE's own vendored torchvision
(/root/reference/venv_vid2vid/lib/python3.7/site-packages/torchvision/models/vgg.py, `vgg19()` = VGG(make_layers(cfg['E']))),
imported from where it lies (nothing is copied).  Weights: text2video_amd.train.vgg19_random_state_dict(11) (seeded,
torchvision's key names); input: seeded tanh(randn) 1x3x16x16; outputs: features[1], [6], [11], [20], [29]
(relu1_1 .. relu5_1, the taps of upstream's Vgg19 wrapper).  Run in the build container only:

    python tests/golden/make_vgg_golden.py        # writes tests/golden/vgg19_taps.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from text2video_amd.train import vgg19_random_state_dict  # noqa: E402

REF = "/root/reference/venv_vid2vid/lib/python3.7/site-packages/torchvision/models/vgg.py"
spec = importlib.util.spec_from_file_location("reference_torchvision_vgg", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
net = mod.vgg19(pretrained=False, init_weights=False).eval()
sd = vgg19_random_state_dict(11)
net.features.load_state_dict({k[len("features."):]: v for k, v in sd.items()}, strict=False)   # convs past relu5_1 keep their init (unused)
x = torch.tanh(torch.randn(1, 3, 16, 16, generator=torch.Generator().manual_seed(12)))
taps, cur = [], x
with torch.no_grad():
    for i, layer in enumerate(net.features):
        cur = layer(cur)
        if i in (1, 6, 11, 20, 29):
            taps.append(cur.clone().numpy())
        if i == 29:
            break
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vgg19_taps.npz")
np.savez_compressed(out, x=x.numpy(), seed=np.int64(11), **{"tap%d" % i: t for i, t in enumerate(taps)})
print("wrote", out, [t.shape for t in taps], "%.1f KB" % (os.path.getsize(out) / 1024))
