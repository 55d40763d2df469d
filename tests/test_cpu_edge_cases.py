"""CPU edge cases of the host side: empty / short sequences, how_many, unusual JSON content."""
import json
import os
import shutil

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_CMD = ("--name fadg0 --dataroot %s --dataset_mode pose --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 "
           "--openpose_only --how_many 1200 --no_first_img --random_drop_prob 0")


def _opt(dataroot, extra=()):
    from text2video_amd.options import TestOptions
    return TestOptions().parse((REF_CMD % dataroot).split() + list(extra))


def _mk(tmp, seqs):
    """seqs: {name: n_frames}; frames are copies of a fixture JSON"""
    src = os.path.join(GOLD, "keypoints_fadg0", "sa1_000_keypoints.json")
    for name, n in seqs.items():
        d = os.path.join(tmp, "test_openpose", name)
        os.makedirs(d)
        for i in range(n):
            shutil.copyfile(src, os.path.join(d, "%05d.json" % i))
    return tmp


def test_missing_dataroot_and_empty_sequences(tmp_path):
    from text2video_amd.pose_dataset import PoseDataset
    with pytest.raises(FileNotFoundError):
        PoseDataset(_opt(str(tmp_path / "nope")))
    root = _mk(str(tmp_path / "a"), {"empty": 0, "short": 2, "ok": 3})
    ds = PoseDataset(_opt(root, ["--fast_pose"]))
    # `empty` has no files (ignored), `short` has fewer pose maps than n_frames_G (no output), `ok` yields 1 frame
    assert len(ds) == 1 and ds[0]["seq"] == "ok" and ds[0]["change_seq"]
    assert ds.seq_lengths() == {"ok": 3, "short": 2}
    assert len(list(ds.iter_prefetch(workers=1))) == 1
    assert list(ds.iter_prefetch(workers=2, limit=0)) == []


def test_mismatched_image_count_is_an_error(tmp_path):
    from PIL import Image
    from text2video_amd.pose_dataset import PoseDataset
    root = _mk(str(tmp_path / "b"), {"s": 4})
    os.makedirs(os.path.join(root, "test_img", "s"))
    for i in range(3):   # one image short
        Image.new("RGB", (512, 384)).save(os.path.join(root, "test_img", "s", "%04d.jpg" % i))
    with pytest.raises(ValueError, match="4 pose files vs 3 images"):
        PoseDataset(_opt(root))


def test_rasteriser_degenerate_json(tmp_path):
    """no people / zero-confidence people / coincident points produce a black (or partial) map, never an error"""
    from text2video_amd.keypoints import read_keypoints
    p = str(tmp_path / "k.json")
    json.dump({"people": []}, open(p, "w"))
    assert read_keypoints(p, (64, 48)).sum() == 0
    person = {"pose_keypoints_2d": [0.0] * 75, "face_keypoints_2d": [0.0] * 210, "hand_left_keypoints_2d": [],
              "hand_right_keypoints_2d": []}
    json.dump({"people": [person]}, open(p, "w"))
    m = read_keypoints(p, (64, 48), hand_discs=False)
    assert m.shape == (48, 64, 3) and m.sum() == 0
    # two identical valid points: zero-length segment -> nothing drawn; points off-canvas are clamped
    person["pose_keypoints_2d"][0:6] = [10.0, 10.0, 0.9, 10.0, 10.0, 0.9]
    person["pose_keypoints_2d"][6:9] = [500.0, 400.0, 0.9]     # far outside a 64x48 canvas
    json.dump({"people": [person]}, open(p, "w"))
    m = read_keypoints(p, (64, 48), hand_discs=False)
    assert m.shape == (48, 64, 3)
    # two people accumulate with uint8 wrap-around
    json.dump({"people": [person, person]}, open(p, "w"))
    m2 = read_keypoints(p, (64, 48), hand_discs=False)
    assert np.array_equal(m2, (m.astype(np.uint16) * 2 % 256).astype(np.uint8))


def test_chunk_restriction_resets_recurrence(tmp_path):
    from text2video_amd.distributed import assign_chunks
    from text2video_amd.pose_dataset import PoseDataset
    root = _mk(str(tmp_path / "c"), {"only": 9})
    ds = PoseDataset(_opt(root, ["--fast_pose"]))
    plan = assign_chunks(ds.seq_lengths(), 2)
    assert sum(len(p) for p in plan) == 2
    outs = []
    for r in range(2):
        d = PoseDataset(_opt(root, ["--fast_pose"]))
        d.restrict(plan[r])
        items = [(it["A_path"], it["change_seq"]) for it in d]
        assert items[0][1] and not any(c for _, c in items[1:])      # exactly one recurrence reset per chunk
        outs += [os.path.basename(p) for p, _ in items]
    assert sorted(outs) == ["%05d.json" % i for i in range(2, 9)]    # every output frame exactly once


def test_face_region_rule():
    """--add_face_disc crop [RECALL upstream get_face_region]: bounding box of the nose-neck colour over ALL frames of
    the chunk, centred on its midpoint, centre clamped into the image, None without the colour."""
    import numpy as np
    from text2video_amd.keypoints import NOSE_NECK_RGB
    from text2video_amd.train import get_face_region
    a = np.zeros((2, 512, 320, 3), np.uint8)
    assert get_face_region(a, 512) is None
    a[0, 100:140, 150:154] = NOSE_NECK_RGB
    a[1, 120:180, 160:166] = NOSE_NECK_RGB
    ys, ye, xs, xe = get_face_region(a, 512)
    assert (ye - ys, xe - xs) == (128, 128)
    assert (ys + ye) // 2 == (100 + 179) // 2 and (xs + xe) // 2 == (150 + 165) // 2
    b = np.zeros((64, 64, 3), np.uint8)
    b[0:3, 60:64] = NOSE_NECK_RGB                      # at the corner: the centre is clamped so the crop stays inside
    ys, ye, xs, xe = get_face_region(b, 128)            # side 32
    assert (ys, ye, xs, xe) == (0, 32, 31, 63)


def test_raster_pool_respawns_a_dead_worker_and_returns_writable_maps():
    """The pool is cached per process (test_fifo.py serves many requests): a worker that died must not fail every later
    job of its I/O thread -- the job is retried once on a fresh worker.  Maps come back writable."""
    import os
    import numpy as np
    from text2video_amd.raster_pool import RasterPool
    from text2video_amd.keypoints import read_keypoints
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keypoints_fadg0")
    src = os.path.join(gold, sorted(f for f in os.listdir(gold) if f.startswith("sa1_"))[0])
    job = (src, (128, 96), (128, 96), False, False, False, True, False)
    pool = RasterPool(1)
    try:
        want = read_keypoints(src, (128, 96), hand_discs=False)
        a = pool.submit(job).result(timeout=120)
        assert np.array_equal(a, want) and a.flags.writeable
        pool._procs[0].kill()                      # the worker dies between two requests
        pool._procs[0].wait()
        b = pool.submit(job).result(timeout=120)
        assert np.array_equal(b, want)
        c = pool.submit(job).result(timeout=120)   # and the replacement keeps serving
        assert np.array_equal(c, want)
    finally:
        pool.close()


def test_lane_plan_and_lockstep_iteration_cover_the_same_items():
    """PoseDataset.iter_lanes (N recurrences per step for t2v_generator_forward_batch): the same windows, names and
    change_seq flags as the one-at-a-time iteration, every item exactly once, at most one item per lane and step,
    --how_many applied in dataset order before the recurrences are dealt to the lanes."""
    import json
    import os
    import numpy as np
    from text2video_amd.options import TestOptions
    from text2video_amd.pose_dataset import PoseDataset
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keypoints_fadg0")
    files = sorted(f for f in os.listdir(gold) if f.startswith("sa1_"))
    frames = [json.load(open(os.path.join(gold, f))) for f in files]
    opt = TestOptions().parse(["--name", "x", "--dataroot", "x", "--dataset_mode", "pose", "--loadSize", "64", "--no_first_img",
                               "--random_drop_prob", "0"])
    ds = PoseDataset.from_memory(opt, {"tmp": [frames[i % len(frames)] for i in range(7)],
                                       "tmp_smooth": [frames[(2 * i) % len(frames)] for i in range(5)],
                                       "third": [frames[(3 * i) % len(frames)] for i in range(4)]})
    ref = {d["A_path"]: (d["A"], d["change_seq"]) for d in ds.iter_prefetch(1)}
    assert ds.lane_plan(2) == [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]]          # longest first, then least loaded; dataset order inside
    assert ds.lane_plan(1) == [list(range(10))] and ds.lane_plan(8) == [[0, 1, 2, 3, 4], [5, 6, 7], [8, 9]]
    assert ds.lane_plan(2, limit=6) == [[0, 1, 2, 3, 4], [5]]
    for workers in (1, 2):
        for lanes in (1, 2, 3):
            got = {}
            for step in ds.iter_lanes(lanes, workers):
                assert 1 <= len(step) <= lanes and len({k for k, _ in step}) == len(step)
                for k, d in step:
                    assert d["A_path"] not in got
                    got[d["A_path"]] = (d["A"], d["change_seq"])
            assert got.keys() == ref.keys()
            for name in ref:
                assert np.array_equal(ref[name][0], got[name][0]) and ref[name][1] == got[name][1], name
    cut = [d["A_path"] for step in ds.iter_lanes(2, 1, limit=6) for _, d in step]
    assert sorted(cut) == sorted(list(ref)[:6])


def test_packed_weight_cache_remembers_what_the_last_step_used(monkeypatch):
    """train.py's cache of packed weight copies (host logic, no GPU): a copy is made once per optimiser write; after
    `invalidate_packs` the parameter remembers the makers of the copies USED since the last write (and only those), which
    `prefetch_packs` re-runs on the side stream on a GPU -- on the CPU it just drops the list; T2V_TRAIN_PACK_CACHE=0
    bypasses the cache."""
    import torch
    from text2video_amd import train as T
    w = torch.nn.Parameter(torch.arange(6.0))
    calls = []

    def maker(tag):
        def make():
            calls.append(tag)
            return w.detach() * 2
        return make
    a1 = T.cached_pack(w, "a", maker("a"))
    a2 = T.cached_pack(w, "a", maker("a-again"))
    b1 = T.cached_pack(w, "b", maker("b"))
    assert a1 is a2 and calls == ["a", "b"] and torch.equal(b1, w.detach() * 2)
    T.invalidate_packs(w)
    assert w._t2v_packs is None and sorted(w._t2v_repack) == ["a", "b"]
    T.prefetch_packs([w])                     # CPU tensor: nothing is made ahead, the list is dropped
    assert w._t2v_repack is None and getattr(w, "_t2v_pack_event", None) is None
    T.cached_pack(w, "a", maker("a2"))        # next "step" uses only "a"
    T.invalidate_packs(w)
    assert list(w._t2v_repack) == ["a"] and calls == ["a", "b", "a2"]
    # a view that names its owner shares the owner's cache
    v = w.detach()
    v._t2v_owner = w
    T.prefetch_packs([w])
    x1 = T.cached_pack(v, "c", maker("c"))
    assert T.cached_pack(w, "c", maker("c-again")) is x1
    monkeypatch.setenv("T2V_TRAIN_PACK_CACHE", "0")
    n = len(calls)
    T.cached_pack(w, "c", maker("c3"))
    assert len(calls) == n + 1


def test_collected_weight_gradient_bookkeeping(monkeypatch):
    """train._paired_direct_wgrad (host logic, no GPU; the kernels replaced by CPU stand-ins that sum x^T dy): a layer counted
    N times in forward reduces ONCE, when its N-th backward node arrives -- two single images through the two-pointer entry,
    anything else concatenated; every node counts as delivered exactly once; a layer one of whose passes never comes back is
    reduced by flush_pending_weight_gradients over what did arrive; a scope left with a stash raises."""
    import pytest
    import torch
    from text2video_amd import ops
    from text2video_amd import train as T
    calls = []

    def bw(x, dy, desc, accumulate_into=None):
        calls.append(("batch", int(x.shape[0])))
        return torch.einsum("bi,bj->ij", x.reshape(x.shape[0], -1), dy.reshape(dy.shape[0], -1))

    def bw_pair(x0, dy0, x1, dy1, desc, accumulate_into=None):
        calls.append(("pair", 2))
        return torch.outer(x0.reshape(-1), dy0.reshape(-1)) + torch.outer(x1.reshape(-1), dy1.reshape(-1))
    monkeypatch.setattr(ops, "conv2d_backward_weight", bw)
    monkeypatch.setattr(ops, "conv2d_backward_weight_pair", bw_pair)
    monkeypatch.setattr(ops, "backward_weight_strided_supported", lambda d, a, b: True)
    monkeypatch.setattr(ops, "unpack_conv_weight", lambda dwp, d, xcs: dwp)

    def into(dwp, d, xcs, view, acc):
        view.copy_(view + dwp if acc else dwp)
    monkeypatch.setattr(ops, "unpack_conv_weight_into", into)
    g = torch.Generator().manual_seed(0)

    def pair(B):
        return torch.randn(B, 3, generator=g), torch.randn(B, 2, generator=g)

    def want(parts):
        return sum(torch.einsum("bi,bj->ij", x, dy) for x, dy in parts)
    w = torch.nn.Parameter(torch.zeros(3, 2))
    # (1) without a bucket slot: the gradient comes back from the last node, None from the earlier ones
    with T.batched_weight_gradients([w]):
        w._t2v_dw_uses = 3
        parts = [pair(2), pair(2), pair(2)]
        outs = [T._paired_direct_wgrad(w, x, dy, "desc", None) for x, dy in parts]
        assert [o[0] for o in outs] == [True, True, True] and outs[0][1] is None and outs[1][1] is None
        assert torch.allclose(outs[2][1], want(parts)) and calls == [("batch", 6)]
        assert w._t2v_dw_stash is None and w._t2v_dw_uses == 0
        # two single images: the two-pointer entry, no copy
        del calls[:]
        w._t2v_dw_uses = 2
        parts = [pair(1), pair(1)]
        outs = [T._paired_direct_wgrad(w, x, dy, "desc", None) for x, dy in parts]
        assert torch.allclose(outs[1][1], want(parts)) and calls == [("pair", 2)]
        # one use: not taken, the caller reduces the node alone
        w._t2v_dw_uses = 1
        assert T._paired_direct_wgrad(w, *pair(2), "desc", None) == (False, None)
    # (2) with a bucket slot: every node counts once, the slot is written once
    gb = T.GradBuckets([w], bucket_mb=1, register=True)
    with T.batched_weight_gradients([w]):
        gb.begin_step()
        for _ in range(3):
            T.expect_gradient(w)
        gb.seal()
        w._t2v_dw_uses = 3
        sl = T.grad_slot(w)
        parts = [pair(2), pair(1), pair(2)]          # (ragged batches are concatenated all the same)
        del calls[:]
        for i, (x, dy) in enumerate(parts):
            assert T._paired_direct_wgrad(w, x, dy, "desc", sl) == (True, None)
            assert sl.filled is (i == 2)
        assert calls == [("batch", 5)] and torch.allclose(sl.view, want(parts))
        gb.absorb([None])
        gb.finish()
        assert torch.allclose(w.grad, want(parts))
    # (3) a pass that never comes back: the flush reduces what arrived; leaving the scope without it raises
    with T.batched_weight_gradients([w]):
        w._t2v_dw_uses = 3
        parts = [pair(2), pair(2)]
        for x, dy in parts:
            assert T._paired_direct_wgrad(w, x, dy, "desc", None) == (True, None)
        out = T.flush_pending_weight_gradients([w], [None])
        assert torch.allclose(out[0], want(parts)) and w._t2v_dw_stash is None
    with pytest.raises(RuntimeError, match="flush_pending_weight_gradients"):
        with T.batched_weight_gradients([w]):
            w._t2v_dw_uses = 2
            T._paired_direct_wgrad(w, *pair(1), "desc", None)


def test_direct_weight_gradient_splits_a_batch_that_passes_the_kernels_offset_limit(monkeypatch):
    """a discriminator layer's passes are reduced as ONE batch; the weight-gradient kernel addresses its operands
    with 32-bit byte offsets, so a batch whose x or dY would pass 2 GiB (many frames per GPU at 2048x1024) must be reduced in
    chunks that accumulate -- host logic, the kernel replaced by a stand-in."""
    import torch
    from text2video_amd import ops
    from text2video_amd import train as T
    calls = []

    def fake(x, dy, desc, accumulate_into=None):
        calls.append((x.shape[0], accumulate_into is not None))
        part = torch.einsum("bhwc,bhwn->cn", x, dy)
        return part if accumulate_into is None else accumulate_into.add_(part)
    monkeypatch.setattr(ops, "conv2d_backward_weight", fake)
    x, dy = torch.randn(6, 5, 7, 4), torch.randn(6, 5, 7, 8)
    whole = T.direct_weight_gradient(x, dy, None)
    assert calls == [(6, False)]
    del calls[:]
    monkeypatch.setattr(T, "_WGRAD_MAX_BYTES", 4 * (5 + 8) * (7 + 8) * 4 * 2 + 1)       # room for two (padded) images per launch
    chunked = T.direct_weight_gradient(x, dy, None)
    assert calls == [(2, False), (2, True), (2, True)] and torch.allclose(chunked, whole, atol=1e-5)
    del calls[:]
    monkeypatch.setattr(T, "_WGRAD_MAX_BYTES", 16)                                      # not even one: one image per launch
    assert torch.allclose(T.direct_weight_gradient(x, dy, None), whole, atol=1e-5) and [c[0] for c in calls] == [1] * 6


def test_trace_overlap_reads_a_two_queue_kernel_trace(tmp_path, capsys):
    """scripts/trace_overlap.py (the two-queue picture of a train step) on a synthetic rocprofv3 kernel trace: two
    steps, each closed by the generator's and the discriminators' Adam launches; in the last one a side-queue weight gradient
    overlaps two main-queue kernels for 60 of its 100 us and 30 us are idle."""
    import csv
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "trace_overlap", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "trace_overlap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows, t = [], 0

    def k(name, q, start, dur):
        rows.append({"Kernel_Name": name, "Queue_Id": q, "Start_Timestamp": start, "End_Timestamp": start + dur})
    for step in range(2):
        base = step * 10_000_000
        k("void t2v::wino_gemm_sk_kernel<x>(p)", "1", base, 50_000)
        k("t2v::loss_terms_kernel(a)", "1", base + 50_000, 10_000)
        k("void t2v::wino_wgrad_sk_kernel<16, 4>(p)", "3", base + 60_000, 100_000)        # side queue
        k("void t2v::wino_gemm_sk_kernel<x>(p)", "1", base + 100_000, 40_000)             # overlaps 40 us
        k("t2v::winograd4_dy_kernel(a)", "1", base + 140_000, 20_000)                     # overlaps 20 us
        k("t2v::adam_multi_kernel(a)", "1", base + 190_000, 2_000_000)                    # (30 us idle before it)
        k("t2v::adam_multi_kernel(a)", "1", base + 2_190_000, 100_000)
    path = tmp_path / "t_kernel_trace.csv"
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    mod.main(str(path), 2)
    out = capsys.readouterr().out
    assert "7 launches" in out and "queues ['1', '3']" in out
    assert "idle 0.03 ms" in out and "two or more 0.06 ms" in out
    assert "first wino_wgrad_sk 0.06" in out and "first adam_multi 0.19" in out
    assert "side queue active from 0.06 to 0.16 ms" in out
