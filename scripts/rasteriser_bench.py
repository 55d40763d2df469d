import time, os, sys, glob
sys.path.insert(0, os.getcwd())
from text2video_amd.keypoints import read_keypoints
files = sorted(glob.glob("tests/golden/keypoints_fadg0/sa1_*.json"))
for exact in (True, False):
    t0 = time.perf_counter(); n = 0
    for _ in range(5):
        for f in files:
            read_keypoints(f, (512, 384), exact_fit=exact); n += 1
    dt = time.perf_counter() - t0
    print("exact_fit=%s: %.1f ms/frame on one core (%d frames)" % (exact, 1e3 * dt / n, n))
