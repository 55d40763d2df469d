"""How close does a split-bf16 contraction come to fp32?  (numerics only, CPU; no kernel uses it.)
a = a1 + a2 + a3 with bf16 terms (round to nearest even); a.b ~ a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1, each product
exact in fp32, accumulated in fp32 -- what six bf16 MFMAs with an fp32 accumulator would compute.  Compared with an
fp64 dot product on the K = 1024 contraction of the Winograd GEMM stage, next to the plain fp32 dot."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x1 = bf16(x)
    x2 = bf16(x - x1)
    x3 = bf16(x - x1 - x2)
    return x1, x2, x3


rng = np.random.default_rng(0)
M, N, K = 256, 512, 1024
for name, A, B in (("unit normal", rng.standard_normal((M, K)), rng.standard_normal((N, K)) * 0.02),
                   ("post-ReLU x wide-range weights", np.maximum(rng.standard_normal((M, K)), 0) * np.exp(rng.standard_normal((M, 1))),
                    rng.standard_normal((N, K)) * 0.02 * np.exp(rng.standard_normal((1, K))))):
    A = A.astype(np.float32); B = B.astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    scale = np.sqrt((ref ** 2).mean())
    f32 = (A @ B.T).astype(np.float64)
    a, b = split3(A), split3(B)
    acc = np.zeros((M, N), np.float32)
    for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)):
        acc = acc + a[i] @ b[j].T
    acc3 = np.zeros((M, N), np.float32)
    for i, j in ((0, 0), (0, 1), (1, 0)):
        acc3 = acc3 + a[i] @ b[j].T
    for tag, got in (("fp32 dot", f32), ("bf16x3, 6 products", acc.astype(np.float64)), ("bf16x3, 3 products", acc3.astype(np.float64)),
                     ("bf16 x bf16 (1 product)", (a[0] @ b[0].T).astype(np.float64))):
        e = got - ref
        print("%-32s %-26s rms %.2e  max %.2e  (of the output's rms)" % (name, tag, np.sqrt((e ** 2).mean()) / scale, np.abs(e).max() / scale))
