#!/usr/bin/env python
"""Drop-in for `../vid2vid/test.py` as the reference invokes it (text2video_audio.sh:37-42):

    CUDA_VISIBLE_DEVICES=1 python test.py --name fadg0 --dataroot datasets/fadg0 --dataset_mode pose \
        --input_nc 3 --resize_or_crop scaleHeight --loadSize 512 --openpose_only --how_many 1200 \
        --no_first_img --random_drop_prob 0

Reads datasets/<name>/test_openpose/<seq>/*.json (+ test_img/<seq>/*.jpg for size and names), writes
results/<name>/test_latest/<seq>/{real_A,fake_B}_*.jpg.  The generator runs on the MI355X through
libt2v_hip.so; there is no CPU fallback.  `--gpu_ids a,b,...` with more than one device runs one rank per device
(whole sequences per rank; --shard_chunks also cuts sequences), started by this script itself or by torchrun.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from text2video_amd.model import run_test      # noqa: E402
from text2video_amd.options import TestOptions  # noqa: E402

if __name__ == "__main__":
    opt = TestOptions().parse()
    # more than one device in --gpu_ids and no torchrun environment: one rank per listed device (text2video_amd/launch.py);
    # the sequences are dealt to the ranks (run_test), every rank writes its own frames
    from text2video_amd import launch          # noqa: E402
    launch.fan_out_if_needed(len(opt.gpu_ids), opt.gpu_ids)
    stats = run_test(opt)
    print("done: %d frames, %.2f fps in the frame loop -> %s" % (stats["frames"], stats["fps_loop"],
                                                                stats["results_dir"]))
    # The reference starts this script once per utterance (text2video_audio.sh:37-44): everything is on disk now, so the
    # process ends here -- the rasteriser workers are closed, the streams flushed, and the interpreter / HIP runtime
    # tear-down (0.26 s of a 3.3 s run on the MI355X box) is skipped.
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:      # (ranks of a multi-GPU run leave through the normal tear-down)
        from text2video_amd import raster_pool     # noqa: E402
        raster_pool._close_all()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
