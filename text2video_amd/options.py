"""Option surface of vid2vid's test.py / test_fifo.py / train.py as the reference drives it.

Flag names used by the reference (must parse and behave): text2video_audio.sh:42, README.md:171-176,
212-214.  The remaining upstream names and their defaults follow SURVEY.md Appendix B [RECALL];
unknown flags are ignored with a warning so that a fork-specific flag does not abort a run.
"""
import argparse
import os
import sys


def _common(p):
    g = p.add_argument
    g("--name", type=str, default="label2city")
    g("--gpu_ids", type=str, default="0")
    g("--checkpoints_dir", type=str, default="./checkpoints")
    g("--model", type=str, default="vid2vid")
    g("--norm", type=str, default="batch", choices=["batch", "instance"])
    g("--fp16", action="store_true", help="accepted, ignored: the MI355X path computes in exact fp32")
    g("--batchSize", type=int, default=1)
    g("--loadSize", type=int, default=512)
    g("--fineSize", type=int, default=512)
    g("--input_nc", type=int, default=3)
    g("--label_nc", type=int, default=0)
    g("--output_nc", type=int, default=3)
    g("--dataroot", type=str, default="datasets/Cityscapes/")
    g("--dataset_mode", type=str, default="temporal")
    g("--resize_or_crop", type=str, default="scaleWidth")
    g("--serial_batches", action="store_true")
    g("--no_flip", action="store_true")
    g("--nThreads", type=int, default=2)
    g("--netG", type=str, default="composite")
    g("--ngf", type=int, default=128)
    g("--n_blocks", type=int, default=9)
    g("--n_downsample_G", type=int, default=3)
    g("--n_blocks_local", type=int, default=3)
    g("--n_frames_G", type=int, default=3)
    g("--n_scales_spatial", type=int, default=1)
    g("--no_first_img", action="store_true")
    g("--use_single_G", action="store_true")
    g("--fg", action="store_true")
    g("--no_flow", action="store_true")
    g("--use_instance", action="store_true")
    g("--densepose_only", action="store_true")
    g("--openpose_only", action="store_true")
    g("--remove_face_labels", action="store_true")
    g("--basic_point_only", action="store_true")
    g("--random_drop_prob", type=float, default=0.2)
    g("--n_gpus_gen", type=int, default=-1)
    g("--debug", action="store_true")
    # extensions of this implementation (not upstream)
    g("--synthetic_weights", type=int, default=None, metavar="SEED",
      help="no checkpoint: run with seeded random-init weights (plumbing / benchmarks only)")
    g("--no_pose_crop", action="store_true", help="keep the full width instead of upstream's central-width crop")
    g("--fast_pose", action="store_true", help="closed-form segment fit in the rasteriser (not bit-identical)")
    g("--no_hand_discs", action="store_true", help="do not draw the two radius-8 hand discs")
    g("--pose_workers", type=int, default=None, help="processes rasterising pose maps ahead of the GPU")
    g("--timing_json", type=str, default=None, help="write fps / per-stage timing to this file")
    g("--resident", action="store_true", help="test.py: run through the resident server (weights stay on the GPU between "
      "calls; started on first use, text2video_amd/resident.py); T2V_RESIDENT=1 does the same")
    g("--resident_idle_s", type=float, default=600.0, help="the resident server leaves after this long without a request")
    g("--resident_stop", action="store_true", help="test.py: stop the resident server of this device selection")
    g("--write_video", action="store_true", help="after the frame loop, mux every sequence's frames into "
      "results/<name>/<name>_<seq>.mp4 at 25 fps (the reference's image2video*.py stage; text2video_amd/mux.py)")
    g("--video_audio", type=str, default=None, help="--write_video: .wav / .mp3 sound track")
    g("--batch_sequences", type=int, default=2, metavar="N", help="advance up to N independent sequences (or chunks) of this "
      "rank in lock-step, one batched generator call per frame; frames are identical to N=1")
    g("--shard_chunks", action="store_true", help="multi-GPU test.py: also cut sequences into chunks so that every rank "
      "has work (each chunk restarts the recurrence; default: whole sequences only, frames identical to 1 GPU)")
    g("--chunks_per_rank", type=int, default=1, metavar="C", help="with --shard_chunks: cut as for C x the number of ranks "
      "and give every rank C chunks (BASELINE configs[2]'s 8 x 64 plan on fewer than 8 GPUs); a rank advances its chunks in "
      "lock-step (--batch_sequences)")
    g("--stitch_frames", type=int, default=0, metavar="K", help="with --shard_chunks: re-generate the first K frames of "
      "every continuation chunk from its predecessor's last frames (all-gathered over RCCL)")
    g("--stitch_rounds", type=int, default=1, help="repetitions of the stitch pass")


class BaseOptions:
    is_train = False

    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        _common(self.parser)
        self.extra()

    def extra(self):
        pass

    def parse(self, argv=None, save=False):
        opt, unknown = self.parser.parse_known_args(argv)
        if unknown:
            print("warning: ignoring unknown options %s" % unknown, file=sys.stderr)
        opt.isTrain = self.is_train
        opt.gpu_ids = [int(i) for i in str(opt.gpu_ids).split(",") if i.strip() != "" and int(i) >= 0]
        # CUDA_VISIBLE_DEVICES from the reference's shell is honoured by the ROCm runtime as well
        # (HIP reads CUDA_VISIBLE_DEVICES when HIP_VISIBLE_DEVICES is unset): nothing to translate.
        # Inference: whether --openpose_only implies no_flow upstream is a recollection; it is the default
        # here, and a checkpoint that carries a flow branch overrides it (create_model).  Training follows the
        # explicit flag only: the reference's recipe (README.md:171-176) passes --openpose_only without --no_flow,
        # and north_star names the flow-warp compositor as part of the generator.
        opt.no_flow_explicit = bool(opt.no_flow)
        if opt.openpose_only and not self.is_train:
            opt.no_flow = True
        if save:
            d = os.path.join(opt.checkpoints_dir, opt.name)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "opt.txt"), "w") as fh:
                for k, v in sorted(vars(opt).items()):
                    fh.write("%s: %s\n" % (k, v))
        return opt


class TestOptions(BaseOptions):
    def extra(self):
        g = self.parser.add_argument
        g("--ntest", type=int, default=float("inf"))
        g("--results_dir", type=str, default="./results/")
        g("--phase", type=str, default="test")
        g("--which_epoch", type=str, default="latest")
        g("--how_many", type=int, default=300)
        g("--use_real_img", action="store_true")
        g("--start_frame", type=int, default=0)
        g("--fifo", type=str, default=None, help="test_fifo.py: named pipe to read requests from")

    def parse(self, argv=None, save=False):
        opt = super().parse(argv, save)
        # test.py forces these upstream
        opt.nThreads = 1
        opt.batchSize = 1
        opt.serial_batches = True
        opt.no_flip = True
        return opt


class TrainOptions(BaseOptions):
    is_train = True

    def extra(self):
        g = self.parser.add_argument
        g("--display_freq", type=int, default=100)
        g("--print_freq", type=int, default=100)
        g("--save_latest_freq", type=int, default=1000)
        g("--save_epoch_freq", type=int, default=1)
        g("--continue_train", action="store_true")
        g("--load_pretrain", type=str, default="")
        g("--which_epoch", type=str, default="latest")
        g("--phase", type=str, default="train")
        g("--niter", type=int, default=10)
        g("--niter_decay", type=int, default=10)
        g("--beta1", type=float, default=0.5)
        g("--lr", type=float, default=0.0002)
        g("--TTUR", action="store_true")
        g("--pool_size", type=int, default=1)
        g("--num_D", type=int, default=1)
        g("--n_layers_D", type=int, default=3)
        g("--ndf", type=int, default=64)
        g("--lambda_feat", type=float, default=10.0)
        g("--lambda_F", type=float, default=10.0)
        g("--lambda_T", type=float, default=10.0)
        g("--no_ganFeat", action="store_true")
        g("--no_vgg", action="store_true")
        g("--vgg_weights", type=str, default="", help="torchvision vgg19 state dict (.pth) for the perceptual loss; the "
          "reference lets torchvision download it, which this tree cannot")
        g("--vgg_random_init", action="store_true", help="run the VGG loss path on seeded random weights (timing / tests)")
        g("--no_lsgan", action="store_true")
        g("--n_frames_D", type=int, default=3)
        g("--n_scales_temporal", type=int, default=2)
        g("--max_frames_per_gpu", type=int, default=1)
        g("--max_frames_backpropagate", type=int, default=1)
        g("--max_t_step", type=int, default=1)
        g("--n_frames_total", type=int, default=30)
        g("--niter_step", type=int, default=5)
        g("--niter_fix_global", type=int, default=0)
        g("--add_face_disc", action="store_true")
        g("--synthetic_data", action="store_true", help="train on seeded synthetic sequences (no dataset needed)")
