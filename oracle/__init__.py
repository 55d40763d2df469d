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

PINNED (host side): keypoints_ref.py restates keypoint2img.py / the L2 interpolation driver and
is checked against golden vectors captured by importing the reference's own Python files
(tests/golden/make_host_goldens.py).
"""
