"""CPU oracle for the vid2vid pose->RGB frame-synthesis path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (text2video_amd/, vid2vid/) never does: it fails loudly when
the HIP library is missing.

PARITY UNPINNED (generator): the generator's source lives in the un-vendored, un-pinned
github.com/sibozhang/vid2vid (fork of NVIDIA/vid2vid); /root/reference holds neither that
source, nor a checkpoint, nor any test/golden vector for it.  The
restatement in generator_ref.py therefore follows SURVEY.md Appendix A (recollection of the
public upstream architecture) with operator semantics pinned to the torch-0.4.1 sources that
ARE vendored in the reference (cited per function).

PINNED to outputs of the reference's own files (golden vectors under tests/golden/, each with the script that
imported the reference file from where it lies):
  * optim_ref.py (Adam)            <- $SP/torch/optim/adam.py                      make_adam_golden.py -> adam041.npz
  * generator_ref.VGG19Features    <- $SP/torchvision/models/vgg.py                make_vgg_golden.py  -> vgg19_taps.npz
  * ToTensor / Normalize / NEAREST resize, default conv init (host side of row a2, seeded weights)
                                   <- $SP/torchvision/transforms/functional.py, $SP/torch/nn/modules/conv.py
                                                                                   make_transforms_golden.py -> transforms.npz
  * the pose rasteriser and both L2 interpolation drivers (product code in text2video_amd/keypoints.py, l2_driver.py,
    which have no oracle twin: they are compared directly with captured reference outputs)
                                   <- keypoint2img.py, interp_landmarks_motion*.py  make_host_goldens.py
"""
