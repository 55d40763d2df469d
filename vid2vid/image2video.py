"""Drop-in for the reference's mux step (`python image2video_real_audio_text2video.py "$1" $2`, text2video_audio.sh:44;
`python image2video.py "$1" $2`, text2video_tts.sh:48): frames written by test.py -> results/<person>/<person>_<test>.mp4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from text2video_amd.mux import image2video
    for path, info in image2video(sys.argv[1], sys.argv[2]):
        print("wrote %s: %s" % (path, info))
