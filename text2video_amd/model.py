"""create_model / frame loop of vid2vid's test.py on the MI355X path.

    opt = TestOptions().parse(); run_test(opt)
is what `cd ../vid2vid && python test.py --name P --dataroot datasets/P --dataset_mode pose ...`
(/root/reference/text2video_audio.sh:37-42) executes.
"""
import json
import os
import sys
import time

import numpy as np
from ._xp import torch     # the real torch, or leantorch under vid2vid/test.py's torch-free frame loop

from . import ops
from .generator import GeneratorSpec, HipGenerator, Vid2VidModelG, synthetic_state_dict
from .pose_dataset import PoseDataset
from .visualizer import Visualizer


def generator_specs(opt):
    """One GeneratorSpec per spatial scale (coarsest first), as Vid2VidModelG.initialize builds them."""
    input_nc = opt.label_nc if opt.label_nc != 0 else opt.input_nc
    specs = []
    for s in range(opt.n_scales_spatial):
        specs.append(GeneratorSpec(
            input_nc=input_nc * opt.n_frames_G, prev_nc=(opt.n_frames_G - 1) * opt.output_nc,
            output_nc=opt.output_nc, ngf=opt.ngf // (2 ** s), n_downsample=opt.n_downsample_G,
            n_blocks=opt.n_blocks if s == 0 else opt.n_blocks_local, no_flow=opt.no_flow, norm=opt.norm,
            is_local=s > 0, scale=s))
    return specs


class LeanUnsupported(RuntimeError):
    """the torch-free frame loop (text2video_amd/_xp.py) met something only torch can do; vid2vid/test.py re-runs the
    command with torch"""


def process_start_time():
    """wall-clock time at which this process was created (for the cold-start split: the interpreter start-up and the
    imports lie between it and the first line of run_test); None where /proc is not readable"""
    try:
        with open("/proc/self/stat") as fh:
            ticks = int(fh.read().rsplit(")", 1)[1].split()[19])         # field 22: starttime, clock ticks after boot
        # age = now on the boot clock - start on the boot clock (10 ms ticks); /proc/stat's btime is whole seconds only
        age = time.clock_gettime(time.CLOCK_BOOTTIME) - ticks / os.sysconf("SC_CLK_TCK")
        return time.time() - age if 0 <= age < 86400 else None
    except Exception:
        return None


def load_checkpoint(path):
    """torch.save'd state-dict (legacy pickle stream of torch 0.4.1 included), tensors only.  Zip-format files (what
    save() here and every torch >= 1.6 write) are memory-mapped: the H2D copies then read straight from the page
    cache instead of from a second host copy of the 1.5 GB file."""
    from . import _xp
    try:
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
        except (RuntimeError, ValueError, TypeError):     # legacy (non-zip) stream: cannot be mapped
            sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:      # noqa: BLE001
        if _xp.LEAN:           # a container leantorch.load does not read: vid2vid/test.py starts over with torch
            raise LeanUnsupported("%s: %s: %s" % (path, type(e).__name__, e)) from e
        raise
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue  # BatchNorm runs in train mode at test time: batch statistics only
        out[k] = v.float()
    return out


def create_model(opt, device="cuda:0", marks=None):
    marks = marks if marks is not None else {}
    if opt.fp16:
        print("warning: --fp16 ignored, the MI355X path computes in exact fp32", file=sys.stderr)
    nets = []
    for s, spec in enumerate(generator_specs(opt)):
        path = os.path.join(opt.checkpoints_dir, opt.name, "%s_net_G%d.pth" % (opt.which_epoch, s))
        if os.path.exists(path):
            t0 = time.perf_counter()
            sd = load_checkpoint(path)
            marks["checkpoint_read_s"] = marks.get("checkpoint_read_s", 0.0) + time.perf_counter() - t0
            has_flow = any(k.startswith("model_final_flow") for k in sd)
            if spec.no_flow and has_flow and not getattr(opt, "no_flow_explicit", False):
                # the architecture follows the checkpoint: it was trained with the flow branch
                print("note: %s carries a flow branch -> flow-warp compositor enabled" % path)
                spec.no_flow = False
            elif not spec.no_flow and not has_flow:
                print("note: %s has no flow branch -> running without flow" % path)
                spec.no_flow = True
        elif opt.synthetic_weights is not None:
            print("warning: %s not found -- using seeded random-init weights (seed %d)" % (path, opt.synthetic_weights))
            sd = synthetic_state_dict(spec, opt.synthetic_weights + s, flow_gain=0.1)
        else:
            raise FileNotFoundError("%s not found (pass --synthetic_weights SEED to run without a checkpoint)" % path)
        t0 = time.perf_counter()
        nets.append(HipGenerator(spec, device).load_state_dict(sd))
        torch.cuda.synchronize(device)
        marks["weights_to_device_s"] = marks.get("weights_to_device_s", 0.0) + time.perf_counter() - t0
        if getattr(torch, "LAST_UPLOAD", None):
            marks["upload"] = dict(torch.LAST_UPLOAD)
    return Vid2VidModelG(nets, opt.n_frames_G, opt.output_nc, opt.no_first_img)


def _lib_max_batch():
    from ._lib import MAX_BATCH
    return MAX_BATCH


def _real_A_u8(pose_map_u8):
    """what util.tensor2im(real_A) writes: the pose map after the Normalize / de-normalize round trip"""
    x = (pose_map_u8.astype(np.float32) / 255.0 - 0.5) / 0.5
    return np.clip((x + 1) / 2.0 * 255.0, 0, 255).astype(np.uint8)


def run_test(opt, model=None, device=None, dataset=None):
    """The frame loop.  Returns a dict of counters/timings.  `dataset`: a ready PoseDataset (e.g.
    PoseDataset.from_memory for the in-memory L2 driver) instead of the one scanned from opt.dataroot.

    Under torchrun (WORLD_SIZE > 1) the sequences are dealt to the ranks (distributed.plan_units): whole
    sequences by default, so every frame equals the single-GPU frame; --shard_chunks also cuts sequences into
    chunks (each a fresh recurrence), and --stitch_frames K then re-generates the first K frames of every
    continuation chunk from its predecessor's true last frames, all-gathered over RCCL.  --how_many counts
    output frames globally, as the single-process loop does.  Every rank writes its own frames.

    --batch_sequences N (default 2): a rank with several independent recurrences to generate (the reference always has
    two per utterance: tmp and tmp_smooth) advances N of them in lock-step, one batched generator call per frame
    (t2v_generator_forward_batch); the frames are those of the one-at-a-time loop, only their order of production differs."""
    t_start = time.perf_counter()
    t_proc = process_start_time()
    marks = {"process_to_run_test_s": (time.time() - t_proc) if t_proc is not None else None}
    if device is None:      # a plain single-device run computes on --gpu_ids[0]
        ids = getattr(opt, "gpu_ids", None)
        device = "cuda:%d" % (ids[0] if ids else 0)
    # the rasteriser workers are fresh interpreters (numpy / scipy / PIL imports): started first, they come up while the
    # checkpoint is read
    from .pose_dataset import default_pose_workers
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # a rank of a multi-GPU job: onto the CPUs of its GPU's NUMA node BEFORE the worker pool starts (the workers inherit
        # the mask); a no-op where sysfs does not place the device (launch.pin_to_numa_node)
        from . import launch as _launch
        _launch.pin_to_numa_node(_launch.local_device_index(int(os.environ.get("LOCAL_RANK", "0"))))
    n_workers = opt.pose_workers if getattr(opt, "pose_workers", None) is not None else default_pose_workers()
    if n_workers > 1:
        from .raster_pool import get_pool
        get_pool(n_workers)
    dataset = dataset if dataset is not None else PoseDataset(opt)
    marks["dataset_scan_s"] = time.perf_counter() - t_start
    world = int(os.environ.get("WORLD_SIZE", "1"))
    plan = rank = None
    limit = opt.how_many
    cpr = int(getattr(opt, "chunks_per_rank", 1) or 1)
    if world > 1 or (getattr(opt, "shard_chunks", False) and cpr > 1):
        from . import distributed as D
        rank, local_rank = 0, None
        if world > 1:
            rank, local_rank, world = D.init_from_env()
        plan = D.plan_units(dataset.seq_lengths(), world, opt.n_frames_G, getattr(opt, "shard_chunks", False),
                            None if opt.how_many in (None, float("inf")) else int(opt.how_many), cpr)
        dataset.restrict(plan[rank])
        limit = None            # already applied globally
        if local_rank is not None:
            device = "cuda:%d" % local_rank
    n_lanes = max(1, min(int(getattr(opt, "batch_sequences", 2) or 1), _lib_max_batch()))
    # the first step's pose maps are rasterised by the pool while the model is created (checkpoint -> device)
    first_steps = dataset.iter_lanes(n_lanes, opt.pose_workers, limit=limit)
    primer = None
    if model is None:
        import threading
        primed = {}

        def _prime():
            try:
                primed["step"] = next(first_steps)
            except StopIteration:
                primed["step"] = None
            except BaseException as e:      # noqa: BLE001 -- re-raised in the frame loop's thread
                primed["error"] = e
        primer = threading.Thread(target=_prime, daemon=True)
        primer.start()
        t0 = time.perf_counter()
        model = create_model(opt, device, marks)
        marks["create_model_s"] = time.perf_counter() - t0

    def primed_steps():
        if primer is not None:
            primer.join()
            if "error" in primed:
                raise primed["error"]
            if primed["step"] is None:
                return
            yield primed["step"]
        for st in first_steps:
            yield st
    vis = Visualizer(opt)
    dev = torch.device(device)
    cs = ops.round_up(3 * opt.n_frames_G, 4)
    pinned = {}      # shape -> ring of 3 pinned host buffers (allocating pinned memory per frame costs ~0.2 ms)
    counters = {"n": 0, "t_loop0": 0.0}
    # where the host spends the loop: waiting for the next step's pose maps (rasteriser), uploading them, enqueueing the
    # generator / tensor2im / D2H, and waiting for the previous step's frames (the GPU) + handing them to the JPEG threads
    split = marks.setdefault("loop_split", {"wait_pose_s": 0.0, "upload_s": 0.0, "enqueue_s": 0.0, "finish_s": 0.0})
    tails = {}       # unit index -> FIFO of generated frames the unit ended with (stitch pass)

    class Lane:
        """one recurrence being advanced: its FIFO of generated frames, pose window and resident pose maps"""
        def __init__(self):
            from .generator import Recurrence
            self.rec, self.window, self.dev_maps, self.unit = Recurrence(), None, None, None

    def frame_loop(steps, tails, start_state=None):
        """steps: iterable of [(lane, item), ...] (PoseDataset.iter_lanes): the items of one step belong to independent
        recurrences and go through the generator in ONE batched call per frame geometry."""
        lanes = {}
        pending = []   # (event, pinned uint8 frame, path, real_A) of the previous step: D2H overlaps the next step

        def finish(p):
            ev, host, a_path, real_a = p
            ev.synchronize()
            ops.check_async_errors()      # e.g. a fixed-grid hand-over that timed out while this frame was computed
            vis.save_images({"real_A": real_a, "fake_B": host.numpy()[..., :3].copy()}, a_path)

        def close_unit(L):
            if L.unit is not None and L.rec.prev is not None:
                if len(L.rec.prev) == 1:
                    tails[L.unit] = L.rec.prev[0].clone()
                else:       # a spatial pyramid (n_scales_spatial > 1): every level's FIFO travels, packed into one tensor
                    from . import distributed as D
                    tails[L.unit] = D.pack_state(L.rec.prev)

        steps = iter(steps)
        while True:
            tq = time.perf_counter()
            try:
                step = next(steps)       # (the rasteriser pool is ahead of the loop, or the loop waits here)
            except StopIteration:
                break
            split["wait_pose_s"] += time.perf_counter() - tq
            tq = time.perf_counter()
            groups = {}      # frame geometry -> [(lane object, item)]
            for k, data in step:
                if counters["n"] == 0 and not groups:
                    counters["t_loop0"] = time.perf_counter()
                    marks.setdefault("to_first_step_s", counters["t_loop0"] - t_start)
                L = lanes.setdefault(k, Lane())
                A = data["A"]  # [tG, H, W, 3] uint8
                H, W = A.shape[1], A.shape[2]
                if data["change_seq"] or L.dev_maps is None or L.window.shape[:2] != (H, W):
                    close_unit(L)
                    L.rec.reset()
                    if start_state is not None:      # stitch pass: continue from the predecessor chunk's last frames
                        st0 = start_state[data["unit"]]
                        if st0.dim() == 1:           # (packed pyramid levels: distributed.pack_state)
                            from . import distributed as D
                            L.rec.prev = D.unpack_state(st0)
                        else:
                            L.rec.prev = [st0.clone()]
                    L.window = torch.zeros(H, W, cs, dtype=torch.float32, device=dev)
                    L.dev_maps = [torch.from_numpy(A[f]).to(dev) for f in range(opt.n_frames_G)]
                else:
                    L.dev_maps = L.dev_maps[1:] + [torch.from_numpy(A[-1]).to(dev, non_blocking=True)]
                L.unit = data.get("unit")
                for f in range(opt.n_frames_G):
                    ops.pose_u8_to_f32(L.dev_maps[f], L.window, 3 * f)
                groups.setdefault((H, W), []).append((k, L, data))
            split["upload_s"] += time.perf_counter() - tq
            tq = time.perf_counter()
            now = []
            for (gh, gw), members in groups.items():
                tg = time.perf_counter()      # per group: an earlier group's tensor2im / D2H is not this one's generator time
                if len(members) > 1 and not model.lockstep_pays(gh, gw):      # (same frames either way: one call per sequence)
                    outs = [model.inference_nhwc_batch([L.window], [L.rec])[0] for _, L, _ in members]
                else:
                    outs = model.inference_nhwc_batch([L.window for _, L, _ in members], [L.rec for _, L, _ in members])
                split["generator_s"] = split.get("generator_s", 0.0) + time.perf_counter() - tg
                for (k, L, data), out in zip(members, outs):
                    t1 = time.perf_counter()
                    u8 = ops.tensor2im_u8(out)
                    ring = pinned.get((k,) + tuple(u8.shape))
                    if ring is None:      # (not setdefault: its default would allocate three pinned buffers per frame)
                        ring = pinned[(k,) + tuple(u8.shape)] = \
                            [[torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True) for _ in range(3)], 0]
                    host = ring[0][ring[1] % 3]     # the buffer of this lane's frame n-3: its JPEG copy was taken in finish(n-3)
                    ring[1] += 1
                    t2 = time.perf_counter()
                    host.copy_(u8, non_blocking=True)
                    t3 = time.perf_counter()
                    ev = torch.cuda.Event()
                    ev.record()
                    split["tensor2im_s"] = split.get("tensor2im_s", 0.0) + t2 - t1
                    split["d2h_s"] = split.get("d2h_s", 0.0) + t3 - t2
                    now.append((ev, host, data["A_path"], _real_A_u8(data["A"][-1])))
                    print("process image... %s" % data["A_path"])
                    counters["n"] += 1
            split["enqueue_s"] += time.perf_counter() - tq
            if "first_step_s" not in marks:       # incl. the weight pack / Winograd filter transforms of this geometry
                torch.cuda.synchronize(dev)
                marks["first_step_s"] = time.perf_counter() - counters["t_loop0"]
            tq = time.perf_counter()
            for p in pending:
                finish(p)
            split["finish_s"] += time.perf_counter() - tq
            pending = now
        for p in pending:
            finish(p)
        for L in lanes.values():
            close_unit(L)

    frame_loop(primed_steps(), tails)
    marks["loop_s"] = time.perf_counter() - counters["t_loop0"] if counters["n"] else 0.0
    n_first_pass = counters["n"]
    stitch = int(getattr(opt, "stitch_frames", 0) or 0)
    if plan is not None and stitch > 0:
        # all-gather the chunk tails (2 generated frames per chunk) and re-generate the first `stitch` frames of every
        # continuation chunk from its predecessor's; the JPEGs of those frames are overwritten
        from . import distributed as D
        vis.flush()
        units = plan[rank]
        for _ in range(max(1, int(getattr(opt, "stitch_rounds", 1) or 1))):
            known = D.exchange_tails(plan, rank, [tails[j] for j in range(len(units))])
            redo = [(j, u) for j, u in enumerate(units) if u[1] > 0 and (u[0], u[3]) in known]
            if redo:
                dataset.restrict([u for _, u in redo], first_n=stitch)
                redone = {}      # tails of the re-generated runs, under the restricted numbering
                frame_loop(dataset.iter_lanes(n_lanes, opt.pose_workers), redone,
                           start_state={jj: known[(u[0], u[3])] for jj, (_, u) in enumerate(redo)})
                for jj, (j, u) in enumerate(redo):
                    if stitch >= u[2] - u[3]:      # re-generated to its end: this chunk's tail is the new one
                        tails[j] = redone[jj]
    vis.flush()
    marks["to_last_jpeg_s"] = time.perf_counter() - t_start
    videos = []
    if getattr(opt, "write_video", False):
        # the reference's next stage (image2video*.py, text2video_audio.sh:44) on the frames just written; under
        # --shard_chunks a sequence's frames are spread over the ranks' runs of this loop: mux after all have finished
        if plan is not None and getattr(opt, "shard_chunks", False):
            print("note: --write_video skipped under --shard_chunks (run vid2vid/image2video.py once all ranks are done)")
        else:
            from . import mux
            import glob
            own = None if plan is None else {u[0] for u in plan[rank]}    # the ranks share results_dir: mux only what this one wrote
            for seq_dir in sorted(glob.glob(os.path.join(vis.save_dir, "*"))):
                if own is not None and os.path.basename(seq_dir) not in own:
                    continue
                frames = sorted(glob.glob(os.path.join(seq_dir, "fake_B_*.jpg")))
                if frames and os.path.isdir(seq_dir):
                    out = os.path.join(os.path.dirname(vis.save_dir), "%s_%s.mp4" % (opt.name, os.path.basename(seq_dir)))
                    mux.write_mp4(frames, out, mux.FPS, getattr(opt, "video_audio", None) or None)
                    videos.append(out)
    vis.close()
    t_end = time.perf_counter()
    n = counters["n"]
    stats = {"frames": n_first_pass, "frames_regenerated": n - n_first_pass, "seconds_total": t_end - t_start,
             "fps_loop": n / (t_end - counters["t_loop0"]) if n else 0.0, "results_dir": vis.save_dir, "videos": videos,
             "cold_start": dict({k: (round(v, 4) if isinstance(v, float) else
                                     ({kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v))
                                for k, v in marks.items()},
                                pack_s=round(sum(getattr(net, "pack_seconds", 0.0) for net in model.nets), 4),
                                torch_imported="torch" in sys.modules,
                                mux_s=round(t_end - t_start - marks["to_last_jpeg_s"], 4))}
    if opt.timing_json:
        with open(opt.timing_json, "w") as fh:
            json.dump(stats, fh)
    if plan is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from . import distributed as D
        D.leave_group()
    return stats
