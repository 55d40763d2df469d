"""Two more utterances through the reference's own L2 drivers (a second one per driver), captured as golden vectors
like make_host_goldens.py does for the first pair -- runs ONLY in the build container (needs /root/reference):

  tests/golden/l2_driver_sheslipped.npz          interp_landmarks_motion_phoneme_VidTIMIT_smooth.py "she slipped ..." fadg0
  tests/golden/l2_driver_pinyin_jintiantianqi.npz  interp_landmarks_motion.py "今天天气好极了不冷不" henan
  tests/golden/l2_inputs/                          + the reference DATA files these two runs read (time stamps, key-pose JSONs)

Usage:  python tests/golden/make_host_goldens_more.py
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_host_goldens as M  # noqa: E402

CASES = [("interp_landmarks_motion_phoneme_VidTIMIT_smooth.py", "she slipped on the floor", "fadg0", ["*phoneme_data"],
          "l2_driver_sheslipped.npz", "PHONEME"),
         ("interp_landmarks_motion.py", "今天天气好极了不冷不", "henan", ["*pinyin_data", "dict_henan.txt"],
          "l2_driver_pinyin_jintiantianqi.npz", "PINYIN")]

if __name__ == "__main__":
    M._stub_modules()
    from text2video_amd import l2_driver as L
    for script, text, person, links, out_name, spec_name in CASES:
        spec = getattr(L, spec_name)
        work, out = M._run_reference_driver(script, text, person, links)
        gold = {}
        for seq, files in out.items():
            gold[seq] = np.stack([M._vec(f) for f in files])
            gold[seq + "_names"] = np.array([os.path.basename(f) for f in files])
        np.savez_compressed(os.path.join(HERE, out_name), **gold)
        print(out_name, gold["tmp"].shape, gold["tmp_smooth"].shape)
        shutil.rmtree(work)
        bank = L.KeyPoseBank(M.REF, person, spec)
        L.synthesize(text, person, M.REF, spec, bank=bank)
        tsp = L.timestamps_path(M.REF, person, text, spec)
        n = 0
        for src in [tsp] + [os.path.join(bank.dir, f) for f in bank.touched]:
            to = os.path.join(HERE, "l2_inputs", os.path.relpath(src, M.REF))
            if not os.path.exists(to):
                os.makedirs(os.path.dirname(to), exist_ok=True)
                shutil.copyfile(src, to)
                n += 1
        print("  l2 inputs: %d key-pose files touched, %d new files copied" % (len(bank.touched), n))
