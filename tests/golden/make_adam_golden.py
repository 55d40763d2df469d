"""Golden Adam steps produced by the REFERENCE's own vendored torch 0.4.1 optimiser source
(/root/reference/venv_vid2vid/lib/python3.7/site-packages/torch/optim/{optimizer,adam}.py), imported from where it lies
into a scratch package (the files are pure Python on top of tensor ops; `collections.Iterable` is aliased for
Python >= 3.10).  vid2vid's settings: lr 2e-4, betas (0.5, 0.999), eps 1e-8.  Run in the build container only:

    python tests/golden/make_adam_golden.py        # writes tests/golden/adam041.npz
"""
import collections
import collections.abc
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

collections.Iterable = collections.abc.Iterable
SP = "/root/reference/venv_vid2vid/lib/python3.7/site-packages/torch/optim"
pkg = types.ModuleType("reference_optim")
pkg.__path__ = [SP]
sys.modules["reference_optim"] = pkg


def load(name):
    spec = importlib.util.spec_from_file_location("reference_optim." + name, os.path.join(SP, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["reference_optim." + name] = mod
    spec.loader.exec_module(mod)
    return mod


load("optimizer")
Adam = load("adam").Adam
rng = np.random.default_rng(21)
n, steps = 96, 5
p0 = rng.standard_normal(n).astype(np.float32) * 0.05
grads = (rng.standard_normal((steps, n)) * np.array([1e-3, 1.0, 30.0, 1e-6, 0.1])[:, None]).astype(np.float32)
grads[3, :8] = 0.0                                   # zero gradients: the update is -lr * m_hat / (sqrt(v_hat) + eps) of the history
p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
opt = Adam([p], lr=2e-4, betas=(0.5, 0.999), eps=1e-8)
after = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for s in range(steps):
        p.grad = torch.from_numpy(grads[s].copy())
        opt.step()
        after.append(p.detach().numpy().copy())
st = opt.state[p]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adam041.npz")
np.savez_compressed(out, p0=p0, grads=grads, p_after=np.stack(after), exp_avg=st["exp_avg"].numpy(),
                    exp_avg_sq=st["exp_avg_sq"].numpy(), lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8)
print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
