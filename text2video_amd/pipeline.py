"""Text -> frames without the disk round trip between the L2 driver and the generator.

The reference runs `interp_landmarks_motion_*_smooth.py` (writes ~2 x N JSON + 2 x N skeleton JPEGs),
then `test.py` re-reads them [REF text2video_audio.sh:31-42].  Here the driver's sequences stay in memory
and feed the pose dataset directly; frames, names and results layout are those of the two-step form
(tests/test_gpu_e2e.py::test_in_memory_pipeline_equals_file_pipeline).

    python -m text2video_amd.pipeline "<utterance>" <person> [--pinyin] [--l2_root DIR] [test.py flags ...]
`--l2_root` = the reference's Text2Video directory (time stamps, unit tables and key poses are read
relative to it; default "."); checkpoints / results resolve relative to the working directory exactly as
for test.py, so run it from ../vid2vid with `--l2_root ../Text2Video`.
"""
import os
import sys

from . import l2_driver
from .options import TestOptions
from .pose_dataset import PoseDataset


def text_to_frames(text, person, opt, root=".", spec=l2_driver.PHONEME, model=None, device="cuda:0"):
    """-> run_test()'s stats dict; frames are written by the Visualizer as test.py would."""
    from .model import run_test      # (here, not at import: main() chooses the torch-free provider first)
    raw, smooth = l2_driver.synthesize(text, person, root, spec)
    dataset = PoseDataset.from_memory(opt, {"tmp": raw, "tmp_smooth": smooth},
                                      size=l2_driver.canvas_size(spec, person))
    return run_test(opt, model=model, device=device, dataset=dataset)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    spec = l2_driver.PHONEME
    if "--pinyin" in argv:
        argv.remove("--pinyin")
        spec = l2_driver.PINYIN
    root = "."
    if "--l2_root" in argv:
        i = argv.index("--l2_root")
        root = argv[i + 1]
        del argv[i:i + 2]
    if len(argv) < 2:
        raise SystemExit(__doc__)
    text, person, rest = argv[0], argv[1], argv[2:]
    defaults = ["--name", person, "--dataroot", "datasets/%s" % person, "--dataset_mode", "pose", "--input_nc", "3",
                "--resize_or_crop", "scaleHeight", "--loadSize", "512", "--openpose_only", "--how_many", "1200",
                "--no_first_img", "--random_drop_prob", "0"]                      # [REF text2video_audio.sh:42]
    opt = TestOptions().parse(defaults + rest)
    # one process, one device: the frame loop without torch, as vid2vid/test.py runs it (text2video_amd/_xp.py)
    if len(opt.gpu_ids) <= 1 and "WORLD_SIZE" not in os.environ and os.environ.get("T2V_LEAN", "1") != "0" and \
            not getattr(opt, "shard_chunks", False):
        from . import _xp
        _xp.use_lean()
    stats = text_to_frames(text, person, opt, root=root, spec=spec,
                           device="cuda:%d" % (opt.gpu_ids[0] if opt.gpu_ids else 0))
    print("%d frames, %.1f frames/s -> %s" % (stats["frames"], stats["fps_loop"], stats["results_dir"]))


if __name__ == "__main__":
    main()
