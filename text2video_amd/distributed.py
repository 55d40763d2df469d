"""Multi-GPU inference: sequence-chunk data parallelism.

The frame recurrence (frame t consumes generated frames t-1, t-2) makes a sequence strictly serial,
so the unit of independence is the reference's own: a *sequence* (sub-folder; recurrence reset on
change_seq).  Long sequences are cut into contiguous chunks, each treated exactly like a separate
sequence folder (zero previous frames + raw-only first frame, as --no_first_img prescribes).  One
process per GPU; there is NO collective on the data path -- the only exchange is an all-gather of
the finished uint8 frames (RCCL over xGMI; `gloo` in the CPU tests) to every rank / the writer.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
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


def gather_frames(local_frames):
    """All-gather equal-shaped per-rank frame blocks [K, ...] -> [world*K, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_frames
    world = dist.get_world_size()
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
