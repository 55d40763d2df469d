"""N independent recurrences advanced in lock-step on one GPU (t2v_generator_forward_batch): every sequence's frames
must be the frames the single-sequence path generates for it -- the batch only changes how the ResnetBlock chains'
kernels are launched (image index in the transforms' grids, N x T tile rows per Winograd GEMM), never a value that
depends on another sequence; norm statistics stay per image."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_generator import _build, _pose_seq  # noqa: E402

BATCH_CASES = [
    # bottleneck 64x32 (128 tiles of 4x4): Winograd F(4x4,3x3) chains -> the batched kernels
    ("f4_flow", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 128, 256),
    ("f4_noflow_instancenorm", dict(ngf=16, n_downsample=2, n_blocks=3, no_flow=True, norm="instance"), 1, 256, 128),
    # ragged 4x4 tile grid (32x85 bottleneck, tiles padded)
    ("f4_ragged", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 128, 340),
    # F(2x2) / direct ResnetBlock convs: the per-image fallback of the chain
    ("f2_flow", dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 1, 64, 128),
    ("direct_flow", dict(ngf=32, n_downsample=3, n_blocks=3, no_flow=False, norm="batch"), 1, 64, 64),
    # two spatial scales: global generator on the pyramid level + local enhancer, both batched
    ("two_scale_flow", dict(ngf=64, n_downsample=2, n_blocks=2, no_flow=False, norm="batch"), 2, 64, 64),
]


def _window(poses, t):
    from text2video_amd import ops
    A = poses[t - 2:t + 1]
    return ops.nchw_to_nhwc(A.reshape(9, A.shape[-2], A.shape[-1]).contiguous().cuda())


@pytest.mark.parametrize("name,spec_kw,scales,H,W", BATCH_CASES, ids=[c[0] for c in BATCH_CASES])
def test_batched_sequences_equal_the_single_sequence_frames(name, spec_kw, scales, H, W):
    """Three sequences, free-running for 4 frames, advanced (a) one at a time, (b) all three per call, (c) two per
    call with the third joining one frame late (its raw-only first frame meets the others' blended frames in one
    batch).  Frames must agree; equality is reported (the GEMM's tile shape may differ between batch sizes)."""
    from text2video_amd.generator import Recurrence
    _, hip = _build(spec_kw, scales)
    n_seq, n_fr = 3, 4
    seqs = [_pose_seq(n_fr + 2, H, W, seed=10 + i) for i in range(n_seq)]
    single = []
    for i in range(n_seq):
        st = Recurrence()
        single.append([hip.inference_nhwc_batch([_window(seqs[i], t)], [st])[0].clone() for t in range(2, n_fr + 2)])
    # (b) lock-step, batch 3
    states = [Recurrence() for _ in range(n_seq)]
    worst, equal = 0.0, True
    for k, t in enumerate(range(2, n_fr + 2)):
        outs = hip.inference_nhwc_batch([_window(seqs[i], t) for i in range(n_seq)], states)
        for i in range(n_seq):
            d = (outs[i] - single[i][k]).abs().max().item()
            worst, equal = max(worst, d), equal and d == 0.0
    # (c) sequence 2 starts one step later than 0 and 1
    states = [Recurrence() for _ in range(n_seq)]
    for k in range(n_fr):
        members = [0, 1] + ([2] if k >= 1 else [])
        outs = hip.inference_nhwc_batch([_window(seqs[i], 2 + (k if i < 2 else k - 1)) for i in members],
                                        [states[i] for i in members])
        for j, i in enumerate(members):
            d = (outs[j] - single[i][k if i < 2 else k - 1]).abs().max().item()
            worst, equal = max(worst, d), equal and d == 0.0
    print("%s: batched vs single-sequence frames: max|delta| = %.3g (%s)" % (name, worst, "bit-equal" if equal else "not bit-equal"))
    assert worst <= 2e-4


def test_batch2_frame_matches_the_oracle():
    """teacher-forced: two sequences' frames in one batch, each against the CPU oracle run on that sequence (<= 1e-3)"""
    from text2video_amd.generator import Recurrence
    spec_kw = dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=False, norm="batch")
    H, W = 128, 256
    ref, hip = _build(spec_kw, 1)
    seqs = [_pose_seq(5, H, W, seed=20 + i) for i in range(2)]
    wants, fifos = [[], []], [[], []]
    for i in range(2):
        ref.reset()
        for t in range(2, 5):
            fifos[i].append(None if ref.fake_B_prev is None else [p.clone() for p in ref.fake_B_prev])
            wants[i].append(ref.inference(seqs[i][t - 2:t + 1].unsqueeze(0)))
    states = [Recurrence(), Recurrence()]
    worst = 0.0
    for k, t in enumerate(range(2, 5)):
        for i in range(2):
            if fifos[i][k] is not None:
                hip.load_prev(fifos[i][k], states[i])
        outs = hip.inference_nhwc_batch([_window(seqs[i], t) for i in range(2)], states)
        for i in range(2):
            got = outs[i][..., :3].permute(2, 0, 1).cpu()
            worst = max(worst, (got - wants[i][k][0]).abs().max().item())
    print("batch 2 vs oracle: max|delta| = %.3g" % worst)
    assert worst <= 1e-3


def test_batch_argument_errors():
    from text2video_amd.generator import Recurrence
    _, hip = _build(dict(ngf=16, n_downsample=2, n_blocks=2, no_flow=True, norm="instance"), 1)
    p = _pose_seq(3, 64, 64)
    q = _pose_seq(3, 64, 128)
    with pytest.raises(ValueError):
        hip.inference_nhwc_batch([_window(p, 2), _window(q, 2)], [Recurrence(), Recurrence()])     # two geometries
    with pytest.raises(ValueError):
        hip.nets[0].forward_batch([_window(p, 2)] * 9, [None] * 9, [True] * 9)                      # > T2V_MAX_BATCH
