"""World-size-2 `gloo` tests (CPU) of the multi-GPU path: sequence-chunk assignment and the frame
all-gather.  The per-rank "generator" here is a stand-in recurrence with the same dependency
structure as the real one (frame t needs t-1, t-2; reset per chunk); the collective and the
bookkeeping are the code under test."""
import os
import socket
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _poses(seq, n):
    seed = zlib.crc32(seq.encode())      # str hash() is salted per process
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, 4, 5)).astype(np.float32))


def _fake_generate(poses):
    """stand-in with the real recurrence shape: out_t = g(window of 3 poses, out_{t-1}, out_{t-2})"""
    prev = [torch.zeros_like(poses[0]), torch.zeros_like(poses[0])]
    outs = []
    for t in range(2, poses.shape[0]):
        o = torch.tanh(poses[t - 2:t + 1].sum(0) * 0.3 + 0.5 * prev[1] - 0.25 * prev[0])
        prev = [prev[1], o]
        outs.append(o)
    return torch.stack(outs)


def _worker(rank, world, port, seq_lengths, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from text2video_amd import distributed as D
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    plan = D.assign_chunks(seq_lengths, world)
    rng = {s: _poses(s, n) for s, n in seq_lengths.items()}
    mine = []
    for seq, s, e, first_out in plan[rank]:
        mine.append(_fake_generate(rng[seq][s:e]))
    local = torch.cat(mine) if mine else torch.zeros(0, 4, 5)
    counts = [sum((e - s) - 2 for _, s, e, _ in p) for p in plan]
    blocks = D.gather_ragged_frames(local, counts)
    # equal-shape fast path
    k = min(counts)
    full = D.gather_frames(local[:k])
    assert full.shape[0] == world * k
    for q in range(world):
        assert torch.equal(full[q * k:(q + 1) * k], blocks[q][:k])
    if rank == 0:
        torch.save({"plan": plan, "blocks": blocks}, os.path.join(tmpdir, "out.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seq_lengths", [{"tmp": 12, "tmp_smooth": 12}, {"only": 21}, {"a": 9, "b": 14, "c": 5}])
def test_two_rank_gloo_chunk_sharding(tmp_path, seq_lengths):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), seq_lengths, str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "out.pt"), weights_only=False)
    plan, blocks = res["plan"], res["blocks"]
    # every output frame of every sequence is produced exactly once
    covered = {}
    for r, p in enumerate(plan):
        off = 0
        for seq, s, e, first_out in p:
            n = (e - s) - 2
            for j in range(n):
                assert (seq, first_out + j) not in covered
                covered[(seq, first_out + j)] = blocks[r][off + j]
            off += n
    assert set(covered) == {(s, t) for s, n in seq_lengths.items() for t in range(2, n)}
    assert all(len(p) > 0 for p in plan)     # both ranks have work
    # each chunk equals a single-process run of the same chunk (sequence-level parity target)
    for r, p in enumerate(plan):
        off = 0
        for seq, s, e, first_out in p:
            poses = _poses(seq, seq_lengths[seq])
            want = _fake_generate(poses[s:e])
            assert torch.equal(blocks[r][off:off + want.shape[0]], want)
            off += want.shape[0]


def test_chunk_bounds_and_assignment_edge_cases():
    from text2video_amd.distributed import assign_chunks, chunk_bounds
    assert chunk_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert chunk_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    plan = assign_chunks({"tmp": 87, "tmp_smooth": 87}, 2)          # config 1: two sequences -> two GPUs
    assert [len(p) for p in plan] == [1, 1] and plan[0][0][1:3] == (0, 87)
    plan = assign_chunks({"seq": 514}, 8)                            # config 3: 512 frames -> 8 x 64
    assert sorted((e - s) - 2 for p in plan for _, s, e, _ in p) == [64] * 8
    firsts = sorted(fo for p in plan for _, _, _, fo in p)
    assert firsts == [2 + 64 * i for i in range(8)]
    assert assign_chunks({"short": 2}, 2) == [[], []]                # shorter than the window: no output
    plan = assign_chunks({"s": 4}, 4)                                # 2 outputs cannot feed 4 ranks
    assert sum(len(p) for p in plan) <= 2


def _ar_worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from text2video_amd import distributed as D
    from text2video_amd.train import allreduce_gradients
    D.init_from_env("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 70000, 3, 1 << 18, 17)]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params.append(torch.nn.Parameter(torch.zeros(4)))          # a parameter without gradient is skipped
    nbytes = allreduce_gradients(params, bucket_mb=1)           # several buckets
    assert nbytes == 4 * sum(p.numel() for p in params[:-1])
    for i, p in enumerate(params[:-1]):
        want = (1 + 2) / 2.0 * (i + 1)                          # mean over the two ranks
        assert torch.allclose(p.grad, torch.full_like(p, want))
    # two exchanges in flight at once (generator gradients under the discriminators' backward pass)
    from text2video_amd.train import allreduce_gradients_begin
    a = [torch.nn.Parameter(torch.zeros(n)) for n in (300000, 9)]
    b = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 40000)]
    for i, p in enumerate(a):
        p.grad = torch.full_like(p, float(rank) + i)
    xa = allreduce_gradients_begin(a, bucket_mb=1)
    for i, p in enumerate(b):                                   # "backward of D" while xa is in flight
        p.grad = torch.full_like(p, 10.0 * rank - i)
    xb = allreduce_gradients_begin(b, bucket_mb=1)
    assert xa.finish() == 4 * 300009 and xb.finish() == 4 * 40007
    for i, p in enumerate(a):
        assert torch.allclose(p.grad, torch.full_like(p, 0.5 + i))
    for i, p in enumerate(b):
        assert torch.allclose(p.grad, torch.full_like(p, 5.0 - i))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    """Bucketed, averaged gradient all-reduce of the train step (replaces DataParallel's GPU-0 star)."""
    mp.spawn(_ar_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _fake_generate_from(poses, state, n_frames):
    """the stand-in recurrence with an explicit start state (None: zeros) -> (frames, FIFO state after them)"""
    prev = [torch.zeros_like(poses[0]), torch.zeros_like(poses[0])] if state is None else [state[0], state[1]]
    outs = []
    stop = poses.shape[0] if n_frames is None else 2 + n_frames
    for t in range(2, stop):
        o = torch.tanh(poses[t - 2:t + 1].sum(0) * 0.3 + 0.5 * prev[1] - 0.25 * prev[0])
        prev = [prev[1], o]
        outs.append(o)
    return torch.stack(outs), torch.stack(prev)


def _stitch_worker(rank, world, port, seq_lengths, mode, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from text2video_amd import distributed as D
    D.init_from_env("gloo")
    shard, stitch, rounds, how_many = mode
    plan = D.plan_units(seq_lengths, world, 3, shard, how_many)
    poses = {s: _poses(s, n) for s, n in seq_lengths.items()}

    def generate(unit, state, n_frames):
        seq, s, e, first_out = unit
        return _fake_generate_from(poses[seq][s:e], state, n_frames)

    frames = D.run_units(plan[rank], plan, rank, generate, stitch, rounds)
    everything = [None] * world
    dist.all_gather_object(everything, [(u, f) for u, f in zip(plan[rank], frames)])
    if rank == 0:
        torch.save(everything, os.path.join(tmpdir, "out.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _collect(tmp_path):
    got = {}
    for per_rank in torch.load(os.path.join(str(tmp_path), "out.pt"), weights_only=False):
        for (seq, s, e, first_out), f in per_rank:
            for j in range(f.shape[0]):
                assert (seq, first_out + j) not in got
                got[(seq, first_out + j)] = f[j]
    return got


def _two_scale_generate(poses, packed_state, n_frames):
    """a recurrence with the state layout of the two-scale generator (n_scales_spatial 2): per pyramid level an [h, w, 8]
    FIFO holding the two previous outputs in channels 0-2 / 3-5; the fine output depends on the coarse one.  State in and
    out in the packed form the tail exchange moves (distributed.pack_state)."""
    from text2video_amd import distributed as D
    H, W = 4, 6
    if packed_state is None:
        fine, coarse = torch.zeros(H, W, 8), torch.zeros(H // 2, W // 2, 8)
    else:
        fine, coarse = D.unpack_state(packed_state)
    outs = []
    stop = poses.shape[0] if n_frames is None else 2 + n_frames
    for t in range(2, stop):
        drive = poses[t - 2:t + 1].sum()
        oc = torch.tanh(0.3 * drive + 0.5 * coarse[..., 3:6] - 0.25 * coarse[..., 0:3] + torch.arange(3) * 0.01)
        up = oc.repeat_interleave(2, 0).repeat_interleave(2, 1)
        of = torch.tanh(0.2 * drive + 0.4 * fine[..., 3:6] - 0.3 * fine[..., 0:3] + 0.5 * up)
        coarse = torch.cat([coarse[..., 3:6], oc, torch.zeros(H // 2, W // 2, 2)], -1)
        fine = torch.cat([fine[..., 3:6], of, torch.zeros(H, W, 2)], -1)
        outs.append(of)
    return torch.stack(outs), D.pack_state([fine, coarse])


def _two_scale_stitch_worker(rank, world, port, n, stitch, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from text2video_amd import distributed as D
    D.init_from_env("gloo")
    plan = D.plan_units({"only": n}, world, 3, True)
    poses = _poses("only", n)
    frames = D.run_units(plan[rank], plan, rank, lambda u, st, k: _two_scale_generate(poses[u[1]:u[2]], st, k), stitch, 1)
    everything = [None] * world
    dist.all_gather_object(everything, [(u, f) for u, f in zip(plan[rank], frames)])
    if rank == 0:
        torch.save(everything, os.path.join(tmpdir, "out.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_stitch_pass_carries_both_pyramid_levels_of_a_two_scale_generator(tmp_path):
    """configs[2] x configs[3]: a generator with n_scales_spatial 2 keeps one FIFO per pyramid level; the chunk tails the
    ranks all-gather carry both (packed into one tensor), and the stitched chunks reproduce the unsharded sequence bit for
    bit.  pack_state / unpack_state round-trip levels of different sizes exactly."""
    from text2video_amd import distributed as D
    lv = [torch.randn(5, 7, 8), torch.randn(3, 4, 8), torch.randn(2, 2, 8)]
    back = D.unpack_state(D.pack_state(lv))
    assert len(back) == 3 and all(torch.equal(a, b) for a, b in zip(lv, back))
    n = 19
    want, _ = _two_scale_generate(_poses("only", n), None, None)
    res = {}
    for name, stitch in (("off", 0), ("full", 1000)):
        d = tmp_path / name
        d.mkdir()
        mp.spawn(_two_scale_stitch_worker, args=(2, _free_port(), n, stitch, str(d)), nprocs=2, join=True)
        res[name] = _collect(d)
        assert set(res[name]) == {("only", t) for t in range(2, n)}
    assert all(torch.equal(res["full"][("only", t)], want[t - 2]) for t in range(2, n))
    assert any(not torch.equal(res["off"][("only", t)], want[t - 2]) for t in range(2, n))      # the seam the pass closes


@pytest.mark.parametrize("seq_lengths,how_many", [({"tmp": 12, "tmp_smooth": 12}, None), ({"only": 21}, None),
                                                  ({"a": 9, "b": 14, "c": 5}, 13)])
def test_multi_rank_frames_equal_single_rank_frames_by_default(tmp_path, seq_lengths, how_many):
    """Default sharding deals whole sequences only: whatever WORLD_SIZE is, every frame is the frame a single
    process generates, and --how_many counts output frames globally in dataset order."""
    mp.spawn(_stitch_worker, args=(2, _free_port(), seq_lengths, (False, 0, 1, how_many), str(tmp_path)), nprocs=2, join=True)
    got = _collect(tmp_path)
    want, budget = {}, how_many
    for seq, n in seq_lengths.items():          # the single-process loop
        f = _fake_generate(_poses(seq, n))
        for j in range(f.shape[0]):
            if budget is not None and budget <= 0:
                break
            want[(seq, 2 + j)] = f[j]
            budget = None if budget is None else budget - 1
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_stitched_chunks_reproduce_the_unsharded_sequence(tmp_path):
    """--shard_chunks cuts the one sequence in two; with the stitch pass re-generating the WHOLE continuation chunk
    from the first chunk's all-gathered tail the result is the unsharded sequence, bit for bit; with a short stitch
    only the first k frames after the cut are the unsharded ones; without it the frames after the cut differ."""
    seq_lengths = {"only": 23}
    want = _fake_generate(_poses("only", 23))
    results = {}
    for name, mode in (("off", (True, 0, 1, None)), ("k3", (True, 3, 1, None)), ("full", (True, 1000, 1, None))):
        d = tmp_path / name
        d.mkdir()
        mp.spawn(_stitch_worker, args=(2, _free_port(), seq_lengths, mode, str(d)), nprocs=2, join=True)
        results[name] = _collect(d)
        assert set(results[name]) == {("only", t) for t in range(2, 23)}
    from text2video_amd.distributed import plan_units
    cut = sorted(fo for p in plan_units(seq_lengths, 2, 3, True) for _, _, _, fo in p)[1]
    for t in range(2, 23):
        w = want[t - 2]
        assert torch.equal(results["full"][("only", t)], w), t
        if t < cut:
            assert torch.equal(results["off"][("only", t)], w) and torch.equal(results["k3"][("only", t)], w)
        elif t < cut + 3:
            assert torch.equal(results["k3"][("only", t)], w)
    assert not torch.equal(results["off"][("only", cut)], want[cut - 2])


def test_plan_units_rules():
    from text2video_amd.distributed import plan_units
    # whole sequences by default, even when that leaves ranks idle
    plan = plan_units({"seq": 514}, 8)
    assert sum(len(p) for p in plan) == 1 and [u for p in plan for u in p] == [("seq", 0, 514, 2)]
    # config 3: --shard_chunks cuts 512 output frames into 8 x 64
    plan = plan_units({"seq": 514}, 8, 3, True)
    assert sorted(e - fo for p in plan for _, _, e, fo in p) == [64] * 8
    # how_many is global and applied in dataset order before the deal
    plan = plan_units({"a": 10, "b": 10}, 2, 3, False, 11)
    assert sorted(u for p in plan for u in p) == [("a", 0, 10, 2), ("b", 0, 5, 2)]
    assert plan_units({"a": 10, "b": 10}, 2, 3, False, 8) == [[("a", 0, 10, 2)], []]


def _buckets_worker(rank, world, port, tmpdir, rs_ag):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), T2V_GRAD_RS_AG="1" if rs_ag else "0")
    from text2video_amd import distributed as D
    from text2video_amd import train as T
    D.init_from_env("gloo")
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (6, 70000, 3, 1 << 18, 17, 40)]
    gb = T.GradBuckets(params, bucket_mb=0.5)
    assert len(gb.bounds) >= 3 and all((hi - lo) % (256 * world) == 0 for lo, hi in gb.bounds)
    # static layout: reverse parameter order, every slot a view of the one flat buffer
    assert gb.members[0][0] == len(params) - 1
    for p, sl in zip(params, gb.slots):
        assert sl.view.shape == p.shape and sl.view.untyped_storage().data_ptr() == gb.flat.untyped_storage().data_ptr()
    for step in range(2):          # the buffers persist across steps
        gb.begin_step()
        uses = [1, 2, 1, 3, 1, 0]  # backward nodes per parameter this step (the last one: never reached)
        for p, u in zip(params, uses):
            for _ in range(u):
                T.expect_gradient(p)
        gb.seal()                  # forward over: parameters without a pending node do not hold their bucket back
        launched_before = gb._launched
        # "backward": last parameters first; a bucket goes out as soon as all of its parameters are complete
        for i in reversed(range(len(params))):
            for k in range(uses[i]):
                T.deliver(T.grad_slot(params[i]), torch.full_like(params[i], float(rank + 1) * (i + 1) + k + step))
        assert launched_before == 0 and gb._launched >= 1          # collectives started from inside the "backward pass"
        gb.absorb([None] * len(params))
        nbytes = gb.finish()
        assert nbytes == 4 * sum(p.numel() for p in params)
        for i, p in enumerate(params):
            if uses[i] == 0:
                assert p.grad is None
                continue
            want = sum(1.5 * (i + 1) + k + step for k in range(uses[i]))      # mean over ranks of the node sum
            assert p.grad.data_ptr() == gb.slots[i].view.data_ptr()            # the gradient IS the bucket slice
            assert torch.allclose(p.grad, torch.full_like(p, want)), (i, float(p.grad.view(-1)[0]), want)
        T.check_presence_across_ranks([gb], "cpu")
    # ranks that disagree on which parameters have gradients are caught
    gb.begin_step()
    for i, p in enumerate(params[:2 + rank]):
        T.expect_gradient(p)
    gb.seal()
    for i, p in enumerate(params[:2 + rank]):
        T.deliver(T.grad_slot(p), torch.ones_like(p))
    gb.absorb([None] * len(params))
    gb.finish()
    try:
        T.check_presence_across_ranks([gb], "cpu")
        raised = False
    except RuntimeError as e:
        raised = "disagree" in str(e)
    assert raised
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rs_ag", [False, True], ids=["all_reduce", "reduce_scatter_all_gather"])
def test_two_rank_gloo_grad_buckets_deliver_launch_in_order_and_average(tmp_path, rs_ag):
    """GradBuckets: persistent flat buckets, gradients delivered in place by the backward nodes, a bucket's collective
    launched when its last gradient lands (in bucket order), all-reduce or reduce-scatter + all-gather, the result
    averaged over the ranks and handed to the optimiser as views; and the guard against ranks whose gradient sets differ."""
    mp.spawn(_buckets_worker, args=(2, _free_port(), str(tmp_path), rs_ag), nprocs=2, join=True)


def test_chunks_per_rank_reproduces_the_finer_plan_on_fewer_ranks():
    """--shard_chunks --chunks_per_rank C: the chunks are those of the world * C rank plan (BASELINE configs[2]: 8 x 64
    frames), every rank holding C consecutive ones -- 1 GPU x 8, 2 x 4, 4 x 2 -- so the frames do not depend on how many
    GPUs the plan runs on."""
    from text2video_amd.distributed import plan_units
    p8 = plan_units({"seq": 514}, 8, 3, shard_chunks=True)
    flat8 = sorted(u for r in p8 for u in r)
    for world, c in ((1, 8), (2, 4), (4, 2)):
        p = plan_units({"seq": 514}, world, 3, shard_chunks=True, chunks_per_rank=c)
        assert len(p) == world and all(len(r) == c for r in p)
        assert sorted(u for r in p for u in r) == flat8
        for r in p:                                   # consecutive chunks per rank
            assert all(r[i][3] + 64 == r[i + 1][3] for i in range(c - 1))
    # two sequences, two ranks, two chunks each: every sequence stays on one rank
    p = plan_units({"a": 10, "b": 8}, 2, 3, shard_chunks=True, chunks_per_rank=2)
    assert [sorted({u[0] for u in r}) for r in p] == [["a"], ["b"]]
    assert plan_units({"a": 10}, 1, 3, shard_chunks=False, chunks_per_rank=4) == [[("a", 0, 10, 2)]]     # only with --shard_chunks


# ---- world 8: the metric's other half ("1/2/4/8 MI355X"; the reference's train recipe is 8-way, README.md:171-176) ----------
# every multi-process test above runs exactly two ranks.  The same code paths with EIGHT live gloo ranks:
# the chunk plan of configs[2], the two-sequence layout of configs[0] (six ranks idle -- they still have to take part in
# every collective), the ragged frame gather, and both forms of the gradient exchange with parameter sizes that are not
# multiples of 8.

WORLD8 = 8


@pytest.mark.parametrize("seq_lengths,mode", [
    ({"seq": 514}, (True, 0, 1, None)),                       # configs[2]: 512 frames -> 8 chunks of 64
    ({"seq": 514}, (True, 3, 1, None)),                       # ... with a 3-frame stitch pass (the 2-frame FIFO all-gather)
    ({"tmp": 87, "tmp_smooth": 87}, (False, 0, 1, None)),     # configs[0]'s layout, whole sequences: 6 idle ranks
    ({"tmp": 87, "tmp_smooth": 87}, (False, 3, 1, None)),     # ... idle ranks inside the tail exchange
    ({"a": 9, "b": 14, "c": 5}, (True, 2, 2, 17)),            # fewer sequences than ranks, how_many cap, two stitch rounds
], ids=["cfg2-chunks", "cfg2-chunks-stitched", "cfg0-whole", "cfg0-whole-stitch-on", "ragged-cap"])
def test_eight_rank_gloo_unit_plans(tmp_path, seq_lengths, mode):
    from text2video_amd.distributed import plan_units
    shard, stitch, rounds, how_many = mode
    mp.spawn(_stitch_worker, args=(WORLD8, _free_port(), seq_lengths, mode, str(tmp_path)), nprocs=WORLD8, join=True)
    got = _collect(tmp_path)                # (asserts that no frame was produced twice)
    plan = plan_units(seq_lengths, WORLD8, 3, shard, how_many)
    assert len(plan) == WORLD8
    # every output frame the single-process loop would write (how_many: global, dataset order) exists exactly once
    want_keys, budget = set(), how_many
    for seq, n in seq_lengths.items():
        for t in range(2, n):
            if budget is not None and budget <= 0:
                break
            want_keys.add((seq, t))
            budget = None if budget is None else budget - 1
    assert set(got) == want_keys
    if shard:
        assert all(len(p) >= 1 for p in plan) or sum(len(p) for p in plan) < WORLD8
    else:
        assert sum(1 for p in plan if p) == len(seq_lengths)              # whole sequences: the other ranks are idle
    # frames: a unit that starts a sequence equals the single-process frames up to the first cut; after a cut, the first
    # `stitch` frames continue from the predecessor's tail, i.e. equal a run over the two chunks joined
    for p in plan:
        for seq, s, e, first_out in p:
            poses = _poses(seq, seq_lengths[seq])
            if s == 0:
                ref = _fake_generate(poses[:e])
                for j in range(ref.shape[0]):
                    assert torch.equal(got[(seq, 2 + j)], ref[j]), (seq, j)
            elif stitch == 0:
                ref = _fake_generate(poses[s:e])                           # a chunk restarts the recurrence
                for j in range(ref.shape[0]):
                    assert torch.equal(got[(seq, first_out + j)], ref[j]), (seq, first_out + j)
    if shard and stitch and seq_lengths == {"seq": 514}:
        # one round: every continuation chunk's first 3 frames come from its predecessor's (unstitched) tail
        chunks = sorted(u for p in plan for u in p)
        for (seq, s0, e0, f0), (_, s1, e1, f1) in zip(chunks, chunks[1:]):
            poses = _poses(seq, 514)
            pre, tail = _fake_generate_from(poses[s0:e0], None, None)
            cont, _ = _fake_generate_from(poses[s1:e1], tail, 3)
            for j in range(3):
                assert torch.equal(got[(seq, f1 + j)], cont[j]), (f1, j)


def _gather8_worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from text2video_amd import distributed as D
    D.init_from_env("gloo")
    counts = [5, 0, 3, 0, 0, 7, 1, 0]                          # ragged, several ranks with nothing at all
    local = torch.full((counts[rank], 2, 3), float(rank)) + torch.arange(counts[rank], dtype=torch.float32).view(-1, 1, 1) / 16
    blocks = D.gather_ragged_frames(local, counts)
    assert [b.shape[0] for b in blocks] == counts
    for r, b in enumerate(blocks):
        for j in range(counts[r]):
            assert torch.equal(b[j], torch.full((2, 3), r + j / 16.0))
    full = D.gather_frames(torch.full((2, 4), float(rank)))
    assert full.shape == (16, 4) and all(torch.equal(full[2 * r:2 * r + 2], torch.full((2, 4), float(r))) for r in range(world))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gloo_ragged_frame_gather(tmp_path):
    mp.spawn(_gather8_worker, args=(WORLD8, _free_port(), str(tmp_path)), nprocs=WORLD8, join=True)


def _buckets8_worker(rank, world, port, tmpdir, rs_ag):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), T2V_GRAD_RS_AG="1" if rs_ag else "0")
    from text2video_amd import distributed as D
    from text2video_amd import train as T
    D.init_from_env("gloo")
    sizes = (7, 70001, 3, (1 << 18) + 5, 17, 41, 1)            # none a multiple of 8
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    gb = T.GradBuckets(params, bucket_mb=0.5)
    assert len(gb.bounds) >= 3 and all((hi - lo) % (256 * world) == 0 for lo, hi in gb.bounds)
    gen = torch.Generator().manual_seed(1234)                  # every rank draws ALL ranks' gradients: the mean is known locally
    for step in range(2):
        per_rank = [[torch.randn(n, generator=gen) for n in sizes] for _ in range(world)]
        gb.begin_step()
        uses = [1, 2, 1, 1, 3, 1, 1]
        for p, u in zip(params, uses):
            for _ in range(u):
                T.expect_gradient(p)
        gb.seal()
        for i in reversed(range(len(params))):
            for k in range(uses[i]):
                T.deliver(T.grad_slot(params[i]), per_rank[rank][i] * (k + 1))
        assert gb._launched >= 1
        gb.absorb([None] * len(params))
        nbytes = gb.finish()
        assert nbytes == 4 * sum(sizes)
        for i, p in enumerate(params):
            scale = sum(k + 1 for k in range(uses[i]))
            want = torch.stack([per_rank[r][i] for r in range(world)]).double().mean(0).float() * scale
            assert p.grad.data_ptr() == gb.slots[i].view.data_ptr()
            assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), (i, (p.grad - want).abs().max().item())
        T.check_presence_across_ranks([gb], "cpu")
    # replicas: the averaged gradients are the same bits on every rank (what keeps 8 optimiser replicas in step)
    digest = torch.tensor([float(gb.flat.double().sum().item())], dtype=torch.float64)
    all_d = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(all_d, digest)
    assert all(torch.equal(all_d[0], d) for d in all_d)
    # plain (non-bucketed) exchange at world 8
    from text2video_amd.train import allreduce_gradients
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 70001, 3)]
    for i, p in enumerate(ps):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    assert allreduce_gradients(ps, bucket_mb=1) == 4 * 70009
    for i, p in enumerate(ps):
        assert torch.allclose(p.grad, torch.full_like(p, (world + 1) / 2.0 * (i + 1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rs_ag", [False, True], ids=["all_reduce", "reduce_scatter_all_gather"])
def test_eight_rank_gloo_grad_buckets_average_equals_mean_of_rank_gradients(tmp_path, rs_ag):
    """GradBuckets at world 8, AVG all-reduce and reduce-scatter + all-gather, parameter sizes that are no multiples of 8 (the
    buckets pad to 256 * world): every parameter's gradient equals the mean over the eight ranks' own gradients, and the
    flat buffers are bit-equal on all ranks afterwards."""
    mp.spawn(_buckets8_worker, args=(WORLD8, _free_port(), str(tmp_path), rs_ag), nprocs=WORLD8, join=True)
