#!/usr/bin/env python
"""Drop-in for `../vid2vid/test.py` as the reference invokes it (text2video_audio.sh:37-42):

    CUDA_VISIBLE_DEVICES=1 python test.py --name fadg0 --dataroot datasets/fadg0 --dataset_mode pose \
        --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 --openpose_only --how_many 1200 \
        --no_first_img --random_drop_prob 0

Reads datasets/<name>/test_openpose/<seq>/*.json (+ test_img/<seq>/*.jpg for size and names), writes
results/<name>/test_latest/<seq>/{real_A,fake_B}_*.jpg.  The generator runs on the MI355X through
libt2v_hip.so; there is no CPU fallback.  `--gpu_ids a,b,...` with more than one device runs one rank per device
(whole sequences per rank; --shard_chunks also cuts sequences), started by this script itself or by torchrun.
`--resident` (or T2V_RESIDENT=1 in the environment of an unchanged text2video_audio.sh): the command becomes a thin client of
a server process that keeps the weights on the GPU between utterances (text2video_amd/resident.py); `--resident_stop` ends it.
A plain single-device run never imports torch: the frame loop takes its device buffers, stream and events from the
library's own host-plumbing entry points (text2video_amd/_xp.py, leantorch.py; T2V_LEAN=0 keeps torch).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


if __name__ == "__main__":
    _args = sys.argv[1:]
    _multi = any(a == "--gpu_ids" and i + 1 < len(_args) and "," in _args[i + 1].strip(",") for i, a in enumerate(_args))
    # (chunked / stitched runs go through text2video_amd/distributed.py, i.e. through torch: neither the torch-free loop nor
    # the resident server -- itself torch-free -- takes them)
    _chunked = any(a in ("--shard_chunks", "--chunks_per_rank", "--stitch_frames", "--stitch_rounds") for a in _args)
    if ("--resident" in _args or "--resident_stop" in _args or os.environ.get("T2V_RESIDENT") == "1") and not _multi \
            and "WORLD_SIZE" not in os.environ and not _chunked:
        # thin client of the resident server (text2video_amd/resident.py): no torch import, no checkpoint load in this process
        from text2video_amd import resident
        _rc = resident.client(_args)
        if _rc is not None:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(_rc)

    # a plain single-device run: the frame loop without torch (the reference starts this script once per utterance, and
    # `import torch` was the largest term of its start-up)
    if not _multi and "WORLD_SIZE" not in os.environ and os.environ.get("T2V_LEAN", "1") != "0" and not _chunked:
        from text2video_amd import _xp
        _xp.use_lean()

from text2video_amd.model import LeanUnsupported, run_test      # noqa: E402
from text2video_amd.options import TestOptions  # noqa: E402

if __name__ == "__main__":
    opt = TestOptions().parse()
    # more than one device in --gpu_ids and no torchrun environment: one rank per listed device (text2video_amd/launch.py);
    # the sequences are dealt to the ranks (run_test), every rank writes its own frames
    from text2video_amd import launch          # noqa: E402
    launch.fan_out_if_needed(len(opt.gpu_ids), opt.gpu_ids)
    try:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # a rank of a multi-GPU job: its exception (a collective's timeout, a peer that never arrived) becomes one stderr
            # line naming the rank and exit status 3; the launcher stops the others (text2video_amd/distributed.py)
            from text2video_amd.distributed import fail_loudly     # noqa: E402
            stats = fail_loudly(run_test, opt)
        else:
            stats = run_test(opt)
    except LeanUnsupported as e:       # e.g. a checkpoint container leantorch.load does not read: start over with torch
        print("note: %s -- running with torch" % e, file=sys.stderr)
        from text2video_amd import raster_pool     # noqa: E402
        raster_pool._close_all()
        sys.stdout.flush()
        sys.stderr.flush()
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, T2V_LEAN="0"))
    print("done: %d frames, %.2f fps in the frame loop -> %s" % (stats["frames"], stats["fps_loop"],
                                                                stats["results_dir"]))
    # The reference starts this script once per utterance (text2video_audio.sh:37-44): everything is on disk now, so the
    # process ends here -- the rasteriser workers are closed, the streams flushed, and the interpreter / HIP runtime
    # tear-down (0.26 s of a 3.3 s run on the MI355X box) is skipped.
    # (ranks of a multi-GPU run leave through the normal tear-down; so does a run under a profiler that writes its output at
    # exit: T2V_NO_FAST_EXIT=1)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("T2V_NO_FAST_EXIT") != "1":
        from text2video_amd import raster_pool     # noqa: E402
        raster_pool._close_all(kill=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
