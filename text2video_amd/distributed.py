"""Multi-GPU inference: sequence-chunk data parallelism.

The frame recurrence (frame t consumes generated frames t-1, t-2) makes a sequence strictly serial,
so the unit of independence is the reference's own: a *sequence* (sub-folder; recurrence reset on
change_seq).  `plan_units` deals WHOLE sequences to the ranks by default -- the frames are then the
single-GPU frames, whatever WORLD_SIZE is.  With `shard_chunks` (test.py --shard_chunks; BASELINE
config 3) sequences are cut into contiguous chunks so that every rank has work; a chunk is treated
exactly like a separate sequence folder (zero previous frames + raw-only first frame, as
--no_first_img prescribes): the parity target is the oracle run on the same chunks, and the frames
differ from the unsharded run's near the cuts.  The optional stitch pass narrows that: the ranks
all-gather the FIFO of generated frames each chunk ended with (RCCL over xGMI: the prev-frame
dependency, 2 frames per chunk) and every continuation chunk re-generates its first k frames from
its predecessor's true tail.  One process per GPU; there is no other collective on the data path
besides the optional all-gather of finished uint8 frames to every rank / the writer.
"""
import os

import torch
import torch.distributed as dist


EXIT_RANK_FAILURE = 3          # exit status of a rank that gave up on its peers (timeouts, a peer that never arrived)


class RankFailure(RuntimeError):
    """A peer did not show up / a collective did not complete inside T2V_DIST_TIMEOUT_S."""


def dist_timeout_s():
    """Seconds every rendezvous and every collective may take before it is an error (T2V_DIST_TIMEOUT_S, default 300;
    the libraries' own defaults are 10 min for RCCL and 30 min for gloo: a wedged rank would cost the job's whole lease)."""
    try:
        return max(1.0, float(os.environ.get("T2V_DIST_TIMEOUT_S", "300")))
    except ValueError:
        return 300.0


_STORE = [None]


def _roll_call(store, tag, rank, world, t):
    """sign in under `tag`, wait for everybody; RankFailure naming the absentees after `t` seconds"""
    import datetime
    keys = ["t2v/rdv/%s/%d" % (tag, r) for r in range(world)]
    try:
        store.set(keys[rank], "1")
        store.wait(keys, datetime.timedelta(seconds=t))
    except Exception as e:      # noqa: BLE001 -- DistStoreError (timeout) / a dead store host
        try:
            missing = [r for r in range(world) if not store.check([keys[r]])]
        except Exception:       # noqa: BLE001 -- the store itself is gone: rank 0 hosts it
            missing = [0]
        raise RankFailure("rank(s) %s did not reach '%s' within %.0f s (reported by rank %d of %d; %s)"
                          % (missing, tag, t, rank, world, type(e).__name__)) from e


def init_group(backend, rank, world, device=None):
    """dist.init_process_group with the job's timeout on the rendezvous AND on every later collective (RCCL: the
    watchdog aborts the process when one overruns; gloo: the call raises).  The key-value store is created here, FIRST,
    and a roll call runs on it before the backend's own rendezvous: a rank that never started is reported BY NUMBER on
    every rank that waited for it (the backends only say that "somebody" timed out)."""
    import datetime
    t = dist_timeout_s()
    td = datetime.timedelta(seconds=t)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500"))
    # under torchrun the elastic agent already serves a store on MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE): every rank is
    # a client of it, under the per-attempt prefix torch's own env:// rendezvous uses; started by launch.py (or by hand),
    # rank 0 hosts the store
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    try:
        store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=td, wait_for_workers=False)
        if agent:
            store = dist.PrefixStore("/worker/attempt_%s/t2v" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)
    except Exception as e:      # noqa: BLE001 -- rank 0 (the store's host) never came up, or the port is not reachable
        raise RankFailure("rank %d of %d: no rendezvous store at %s:%d within %.0f s -- rank(s) [0] did not start, or the "
                          "address is unreachable (%s)" % (rank, world, addr, port, t, type(e).__name__)) from e
    _STORE[0] = store
    _roll_call(store, "start", rank, world, t)
    kw = {"timeout": td, "store": store}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    try:
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    except Exception as e:      # noqa: BLE001
        raise RankFailure("rank %d of %d: the process group (%s) did not come up within %.0f s (%s: %s)"
                          % (rank, world, backend, t, type(e).__name__, str(e).splitlines()[0] if str(e) else "")) from e


_RDV_SEQ = {}


def rendezvous(tag, timeout_s=None):
    """Roll call through the job's key-value store (no collective, so it works before the first one and on any
    backend): every rank signs in under `tag`, then waits for all the others.  On a timeout the ranks that did not
    sign in are named -- `RankFailure: rank(s) [5] did not reach 'tails' ...` -- on every rank that waited.  The
    reference's DataParallel re-raises the first worker's exception in the caller
    ($SP/torch/nn/parallel/parallel_apply.py:40-78); a process per GPU needs the store for the same report.  A no-op
    in a single process and on a group that init_group did not create (torchrun-initialised test groups)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or _STORE[0] is None:
        return
    n = _RDV_SEQ[tag] = _RDV_SEQ.get(tag, 0) + 1
    _roll_call(_STORE[0], "%s/%d" % (tag, n), dist.get_rank(), dist.get_world_size(),
               dist_timeout_s() if timeout_s is None else float(timeout_s))


def leave_group():
    """The orderly end of a rank of a multi-rank job: roll call (everybody has finished), then tear the group down.  Leaving
    the interpreter with a live process group aborts now and then in the backends' own threads (`terminate called without an
    active exception`, status -6) -- which the launcher would report as a failed job."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            rendezvous("done")
        dist.destroy_process_group()
        _STORE[0] = None


def fail_loudly(fn, *args, **kw):
    """Run a rank's main function; any exception (a collective's timeout, RankFailure, an error of this rank's own)
    becomes ONE stderr line naming the rank and a non-zero exit status -- through os._exit: the tear-down of a process
    group whose peers are gone can block for its own timeout again.  The launcher (launch.self_launch / torchrun)
    stops the other ranks on the first non-zero status."""
    import sys
    import traceback
    try:
        return fn(*args, **kw)
    except SystemExit:
        raise
    except BaseException as e:      # noqa: BLE001
        if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            raise
        traceback.print_exc()
        print("[t2v] rank %s of %s FAILED: %s: %s" % (os.environ.get("RANK", "?"), os.environ.get("WORLD_SIZE", "?"),
                                                     type(e).__name__, str(e).splitlines()[0] if str(e) else ""),
              file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(EXIT_RANK_FAILURE)


def init_from_env(backend=None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if torch.cuda.is_available():
        # --gpu_ids[local_rank] when the self-launcher passed the list on; with T2V_DIST_BACKEND=gloo more ranks than
        # GPUs are allowed (single-GPU tests): ranks then share devices
        from .launch import local_device_index
        local_rank = local_device_index(local_rank)
    if world > 1:
        from .launch import pin_to_numa_node
        pin_to_numa_node(local_rank)       # before any worker pool is forked: the pose workers inherit the mask
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm.  T2V_DIST_BACKEND=gloo: several ranks on ONE GPU (tests of the multi-rank frame loop
            # on a single-GPU box; RCCL refuses two ranks per device) -- the collectives then stage through host memory
            backend = os.environ.get("T2V_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        device = None
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                raise RankFailure("rank %d wants device %d but this process sees %d GPU(s) (RCCL needs one device per rank; "
                                  "T2V_DIST_BACKEND=gloo lets ranks share a device in tests)"
                                  % (rank, local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        init_group(backend, rank, world, device)
    return rank, local_rank, world


def chunk_bounds(n_frames, n_chunks):
    """Contiguous, near-equal chunks of a pose sequence: [(start, stop)], stop exclusive."""
    base, extra = divmod(n_frames, n_chunks)
    out, s = [], 0
    for c in range(n_chunks):
        e = s + base + (1 if c < extra else 0)
        out.append((s, e))
        s = e
    return out


def assign_chunks(seq_lengths, world, n_frames_G=3):
    """Work units for `world` ranks.  seq_lengths: {seq: number of pose maps}.

    Whole sequences are dealt out first (longest first, to the least loaded rank); if there are fewer
    sequences than ranks the longest ones are cut so every rank has work.  A chunk of a cut sequence
    starts n_frames_G-1 pose maps early so that its first output frame is the frame right after the
    previous chunk's last one (the window needs tG pose maps).
    Returns per rank a list of (seq, pose_start, pose_stop, first_output_index).
    """
    units = [(seq, 0, n) for seq, n in seq_lengths.items() if n >= n_frames_G]
    while 0 < len(units) < world:
        units.sort(key=lambda u: u[2] - u[1], reverse=True)
        seq, s, e = units[0]
        n_out = (e - s) - (n_frames_G - 1)
        if n_out < 2:
            break
        half = n_out // 2 + (n_out % 2)
        mid = s + (n_frames_G - 1) + half          # first output index of the second half
        units = units[1:] + [(seq, s, mid), (seq, mid - (n_frames_G - 1), e)]
    loads = [0] * world
    plan = [[] for _ in range(world)]
    for seq, s, e in sorted(units, key=lambda u: (-(u[2] - u[1]), u[0], u[1])):
        r = loads.index(min(loads))
        plan[r].append((seq, s, e, s + n_frames_G - 1))
        loads[r] += (e - s) - (n_frames_G - 1)
    for p in plan:
        p.sort(key=lambda u: (u[0], u[1]))
    return plan


def plan_units(seq_lengths, world, n_frames_G=3, shard_chunks=False, how_many=None, chunks_per_rank=1):
    """Work units per rank: [(seq, pose_start, pose_stop, first_output_index)].

    chunks_per_rank C > 1 (with shard_chunks): sequences are cut as for world * C ranks and every rank takes C
    consecutive shares -- BASELINE configs[2]'s 8 x 64-frame plan on fewer than 8 GPUs (C = 8 / world), the chunks of a
    rank advancing in lock-step on its GPU (test.py --batch_sequences).  The chunks, and so the frames, are those of the
    world * C rank plan.

    how_many caps the number of OUTPUT frames globally, in dataset order -- what the single-process frame loop's
    `if i >= how_many: break` does -- before anything is dealt out.  Without shard_chunks only whole sequences are
    dealt (longest first, to the least loaded rank): ranks may stay idle, but every frame equals the single-GPU
    frame.  With shard_chunks the longest sequences are cut until every rank has work (assign_chunks)."""
    lengths, budget = {}, how_many
    for seq, n in seq_lengths.items():
        n_out = n - (n_frames_G - 1)
        if n_out <= 0:
            continue
        if budget is not None:
            if budget <= 0:
                break
            n_out = min(n_out, budget)
            budget -= n_out
        lengths[seq] = n_out + (n_frames_G - 1)
    if shard_chunks:
        C = max(1, int(chunks_per_rank))
        fine = assign_chunks(lengths, world * C, n_frames_G)
        if C == 1:
            return fine
        plan = [[] for _ in range(world)]
        for v, units in enumerate(fine):          # virtual rank v -> real rank v // C (contiguous shares)
            plan[v // C] += units
        for p in plan:
            p.sort(key=lambda u: (u[0], u[1]))
        return plan
    loads = [0] * world
    plan = [[] for _ in range(world)]
    for seq, n in sorted(lengths.items(), key=lambda kv: (-kv[1], kv[0])):
        r = loads.index(min(loads))
        plan[r].append((seq, 0, n, n_frames_G - 1))
        loads[r] += n - (n_frames_G - 1)
    for p in plan:
        p.sort(key=lambda u: (u[0], u[1]))
    return plan


def pack_state(levels):
    """The FIFO state of a generator with a spatial pyramid (n_scales_spatial > 1: one [h,w,cs] FIFO per level, finest
    first) as ONE flat fp32 tensor, so that the tail exchange moves it like a single-level tail: a header
    (number of levels, then h, w, cs per level -- exact as floats) followed by the levels' values."""
    head = [float(len(levels))] + [float(v) for t in levels for v in t.shape]
    assert all(t.dim() == 3 for t in levels)
    return torch.cat([torch.tensor(head, dtype=torch.float32, device=levels[0].device)]
                     + [t.reshape(-1).to(torch.float32) for t in levels])


def unpack_state(flat):
    """inverse of pack_state: the list of per-level FIFOs (clones: the caller owns them)"""
    n = int(flat[0].item())
    dims = [int(v) for v in flat[1:1 + 3 * n].tolist()]
    out, o = [], 1 + 3 * n
    for i in range(n):
        h, w, c = dims[3 * i:3 * i + 3]
        out.append(flat[o:o + h * w * c].reshape(h, w, c).clone())
        o += h * w * c
    assert o == flat.numel(), "unpack_state: %d values for a header that describes %d" % (flat.numel(), o)
    return out


def exchange_tails(plan, rank, my_tails):
    """The stitch pass's collective.  my_tails: one tensor per unit of plan[rank] (the FIFO of generated frames the
    unit ended with; one shape everywhere).  All-gathers them (RCCL over xGMI / gloo) and returns
    {(seq, pose_stop): tail} over the units of ALL ranks: the chunk whose first output frame is `pose_stop`
    continues from that state."""
    world = len(plan)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {(u[0], u[2]): t for u, t in zip(plan[rank], my_tails)}
    rendezvous("tails")        # (names a rank that never got here, before the collective that would only time out)
    shapes = [None] * world
    dist.all_gather_object(shapes, [tuple(t.shape) for t in my_tails])
    kinds = {s for per_rank in shapes for s in per_rank}
    if not kinds:      # no rank generated anything
        return {}
    if len(kinds) != 1:
        raise NotImplementedError("stitching needs one frame geometry across all chunks, got %s" % sorted(kinds))
    shape = kinds.pop()
    umax = max(len(p) for p in plan)
    ref = my_tails[0] if my_tails else None
    dev = ref.device if ref is not None else (torch.device("cuda", torch.cuda.current_device())
                                              if dist.get_backend() == "nccl" else torch.device("cpu"))
    block = torch.zeros((umax,) + shape, dtype=torch.float32, device=dev)
    for j, t in enumerate(my_tails):
        block[j] = t
    full = gather_frames(block)
    out = {}
    for r, p in enumerate(plan):
        for j, (seq, s, e, fo) in enumerate(p):
            out[(seq, e)] = full[r * umax + j]
    return out


def run_units(units, plan, rank, generate, stitch=0, rounds=1):
    """Generator-agnostic driver of one rank's units.  generate(unit, start_state, n_frames) -> (frames, tail):
    the unit's first n_frames output frames (all if None) starting from start_state (None: a fresh sequence --
    zero previous frames, raw-only first frame) and the FIFO state after the last of them.
    stitch = k > 0: after the chunk pass the tails are exchanged and every continuation chunk re-generates its
    first k frames from its predecessor's tail (`rounds` times: a tail that was itself re-generated to the end of
    its chunk is exact one round later).  With k >= chunk length and rounds >= world - 1 the result is the
    unsharded sequence.  Returns [frames per unit]."""
    frames, tails = [], []
    for u in units:
        f, t = generate(u, None, None)
        frames.append(f)
        tails.append(t)
    for _ in range(rounds if stitch > 0 else 0):
        known = exchange_tails(plan, rank, tails)
        for j, u in enumerate(units):
            seq, s, e, first_out = u
            pred = known.get((seq, first_out)) if s > 0 else None
            if pred is None:
                continue
            n_out = e - first_out
            k = min(stitch, n_out)
            f, t = generate(u, pred, k)
            frames[j] = torch.cat([f, frames[j][k:]]) if torch.is_tensor(frames[j]) else list(f) + list(frames[j][k:])
            if k == n_out:
                tails[j] = t
    return frames


def gather_frames(local_frames):
    """All-gather equal-shaped per-rank frame blocks [K, ...] -> [world*K, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_frames
    world = dist.get_world_size()
    if local_frames.is_cuda and dist.get_backend() == "gloo":     # gloo moves host memory
        host = local_frames.contiguous().cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(local_frames.device)
    out = torch.empty((world * local_frames.shape[0],) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype,
                      device=local_frames.device)
    dist.all_gather_into_tensor(out, local_frames.contiguous())
    return out


def gather_ragged_frames(local_frames, counts):
    """All-gather blocks whose leading sizes differ per rank (counts[r] frames on rank r): pads to the
    maximum, gathers once, and returns the list of per-rank blocks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_frames]
    kmax = max(counts)
    pad = torch.zeros((kmax,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype, device=local_frames.device)
    pad[:local_frames.shape[0]] = local_frames
    full = gather_frames(pad)
    return [full[r * kmax:r * kmax + counts[r]] for r in range(len(counts))]
