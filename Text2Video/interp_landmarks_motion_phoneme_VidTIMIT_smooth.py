#!/usr/bin/env python
"""Drop-in for the reference's interp_landmarks_motion_phoneme_VidTIMIT_smooth.py
(`python interp_landmarks_motion_phoneme_VidTIMIT_smooth.py "<utterance>" <person>`, run from the
Text2Video directory [REF text2video_audio.sh:31]): same inputs, same files under
../vid2vid/datasets/<person>/test_{openpose,img}/{tmp,tmp_smooth}/, bit-identical key points."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import l2_driver  # noqa: E402

if __name__ == "__main__":
    l2_driver.main(spec=l2_driver.PHONEME)
