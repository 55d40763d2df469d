#!/usr/bin/env python
"""Drop-in for `../vid2vid/train.py` with the reference's flag surface (README.md:171-176):

    python train.py --name pose2body_512p --dataroot datasets/pose --dataset_mode pose --input_nc 3 --num_D 2 \
        --resize_or_crop randomScaleHeight_and_scaledCrop --loadSize 544 --fineSize 512 \
        --gpu_ids 0,1,2,3,4,5,6,7 --batchSize 8 --max_frames_per_gpu 2 --niter 500 --niter_decay 5 --no_first_img \
        --n_frames_total 12 --max_t_step 4 --add_face_disc --openpose_only [--synthetic_data]

One process per GPU: the reference fanned out over --gpu_ids inside one process with nn.DataParallel
(torch/nn/parallel/data_parallel.py:116-137); here this script starts one rank per listed device itself
(text2video_amd/launch.py; `python -m torch.distributed.run --nproc-per-node 8 train.py ...` works as well), every
rank owns one clip of the batch and gradients are all-reduced over RCCL.  Built: the train step (generator,
multiscale + face + temporal discriminators, LSGAN + feature matching + VGG19 perceptual loss, Adam), the
real-data loader (train_openpose / train_img), the epoch schedule with learning-rate decay, periodic
checkpoints and --continue_train; --synthetic_data runs it without a dataset.  FlowNet2-based losses are not
built.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from text2video_amd.options import TrainOptions  # noqa: E402
from text2video_amd.train import run_train       # noqa: E402

if __name__ == "__main__":
    opt = TrainOptions().parse(save=False)
    from text2video_amd import launch            # noqa: E402
    launch.fan_out_if_needed(len(opt.gpu_ids), opt.gpu_ids)
    from text2video_amd.distributed import fail_loudly      # noqa: E402
    stats = fail_loudly(run_train, opt)      # (a rank's exception: one line naming the rank, exit status 3; single runs re-raise)
    print("done: %d steps, median %.1f ms/step on %d GPU(s)" % (stats["steps"], stats["ms_per_step"], stats["world"]))
