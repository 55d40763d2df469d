#!/usr/bin/env python
"""Drop-in for the reference's interp_landmarks_motion.py (Mandarin pinyin driver,
`python interp_landmarks_motion.py <utterance> <person>` [REF text2video_tts_chinese.sh:28])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2video_amd import l2_driver  # noqa: E402

if __name__ == "__main__":
    l2_driver.main(spec=l2_driver.PINYIN)
