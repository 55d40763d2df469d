"""Goldens for the pose dataset's image transforms and for torch-0.4.1's default conv initialisation, generated with the
REFERENCE's own vendored sources imported from where they lie (nothing is copied):

  * $SP/torchvision/transforms/functional.py  -> to_tensor (:38-60), normalize (:185-208), resize NEAREST (:211-244)
    on uint8 pose-map-like images (incl. the values 0, 1, 127, 128, 254, 255 and a real fadg0 skeleton canvas size)
  * $SP/torch/nn/modules/conv.py:40-47        -> _ConvNd.reset_parameters: U(-stdv, stdv) with stdv = 1/sqrt(in_channels*prod(k))
    for Conv2d AND ConvTranspose2d (whose weight is [in, out, k, k]: in_channels is dim 0), bias alike

$SP = /root/reference/venv_vid2vid/lib/python3.7/site-packages.  The vendored torchvision imports the long-removed
`PIL.PILLOW_VERSION`; this script aliases it to the installed Pillow's version string before the import (the only
shim).  conv.py is executed inside a throw-away package whose relative imports (.module, .utils, ..functional) resolve
to the installed torch's.  Run in the build container only:   python tests/golden/make_transforms_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import PIL
import torch
from PIL import Image

SP = "/root/reference/venv_vid2vid/lib/python3.7/site-packages"
HERE = os.path.dirname(os.path.abspath(__file__))

if not hasattr(PIL, "PILLOW_VERSION"):
    PIL.PILLOW_VERSION = PIL.__version__
    Image.PILLOW_VERSION = PIL.__version__
import collections.abc  # noqa: E402,F401  (functional.py reads collections.abc.Iterable)

spec = importlib.util.spec_from_file_location("reference_tv_functional", os.path.join(SP, "torchvision/transforms/functional.py"))
TF = importlib.util.module_from_spec(spec)
spec.loader.exec_module(TF)

rng = np.random.default_rng(31)
# ---- to_tensor + normalize(.5,.5): pose-map-like uint8 HWC images
img = rng.integers(0, 256, size=(24, 20, 3), dtype=np.uint8)
img[0, :6, 0] = [0, 1, 127, 128, 254, 255]
t = TF.normalize(TF.to_tensor(Image.fromarray(img)), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
all_values = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
t_all = TF.normalize(TF.to_tensor(Image.fromarray(all_values)), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
# ---- resize NEAREST: a sparse skeleton-like map on the L2 driver's 512x384 canvas -> scaleHeight 512 -> (680 x 512)
sk = np.zeros((384, 512, 3), np.uint8)
pts = rng.integers(0, [384, 512], size=(4000, 2))
sk[pts[:, 0], pts[:, 1]] = rng.integers(1, 256, size=(4000, 3), dtype=np.uint8)
big = np.asarray(TF.resize(Image.fromarray(sk), (512, 680), Image.NEAREST))       # size = (h, w)
small_src = rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
small = np.asarray(TF.resize(Image.fromarray(small_src), (64, 85), Image.NEAREST))
down = np.asarray(TF.resize(Image.fromarray(small_src), (30, 40), Image.NEAREST))

# ---- conv default init: run the vendored _ConvNd.reset_parameters, recording the bounds it hands to uniform_
pkg = types.ModuleType("refnn")
pkg.__path__ = []
pkg.functional = torch.nn.functional
sys.modules["refnn"] = pkg
sub = types.ModuleType("refnn.modules")
sub.__path__ = [os.path.join(SP, "torch/nn/modules")]
sys.modules["refnn.modules"] = sub
sys.modules["refnn.modules.module"] = torch.nn.modules.module
sys.modules["refnn.modules.utils"] = torch.nn.modules.utils
cspec = importlib.util.spec_from_file_location("refnn.modules.conv", os.path.join(SP, "torch/nn/modules/conv.py"))
conv = importlib.util.module_from_spec(cspec)
sys.modules["refnn.modules.conv"] = conv
cspec.loader.exec_module(conv)

calls = []
orig_uniform = torch.Tensor.uniform_


def spy(self, a=0.0, b=1.0, **kw):
    calls.append((tuple(self.shape), float(a), float(b)))
    return orig_uniform(self, a, b, **kw)


torch.Tensor.uniform_ = spy
layers = [("Conv2d", 9, 128, 7), ("Conv2d", 6, 128, 7), ("Conv2d", 128, 256, 3), ("Conv2d", 1024, 1024, 3), ("Conv2d", 128, 3, 7),
          ("ConvTranspose2d", 1024, 512, 3), ("ConvTranspose2d", 256, 128, 3), ("Conv2d", 6, 64, 4)]
rows = []
try:
    for kind, cin, cout, k in layers:
        calls.clear()
        torch.manual_seed(0)
        m = getattr(conv, kind)(cin, cout, k)
        (wshape, wa, wb), (bshape, ba, bb) = calls
        assert wa == -wb and ba == -bb and wb == bb
        assert float(m.weight.abs().max()) <= wb
        rows.append((kind == "ConvTranspose2d", cin, cout, k) + wshape + (wb,))
finally:
    torch.Tensor.uniform_ = orig_uniform

out = os.path.join(HERE, "transforms.npz")
np.savez_compressed(out, img=img, img_norm=t.numpy(), all_values_norm=t_all.numpy(), sk=sk, sk_680x512=big, small_src=small_src,
                    small_85x64=small, down_40x30=down,
                    conv_init=np.array(rows, dtype=np.float64))   # transposed, cin, cout, k, weight shape (4), stdv
print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
for r in rows:
    print(r)
