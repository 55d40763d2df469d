"""Capture golden vectors for the HOST side of the frame path from the reference's own Python.

Runs ONLY in the build container (needs /root/reference); the outputs are committed as small
data fixtures so the GPU box never needs the reference:

  tests/golden/keypoints_fadg0/sa1_XXX_keypoints.json   inputs (reference data files, verbatim)
  tests/golden/pose_maps_fadg0.npz                       expected uint8 maps of
        keypoint2img.read_keypoints(json, (512,384))     [REF keypoint2img.py:70-90]
        with cv2.circle stubbed out, i.e. WITHOUT the two radius-8 discs of keypoint2img.py:159-160
        (their rasterisation rule is OpenCV's and cannot be reproduced without OpenCV)
  tests/golden/l2_driver_Shehadyour.npz                  the 87 tmp + 87 tmp_smooth pose vectors the
        reference L2 driver (interp_landmarks_motion_phoneme_VidTIMIT_smooth.py) writes for config 1
  tests/golden/l2_driver_pinyin_nihaoa.npz               the same for interp_landmarks_motion.py (pinyin
        driver) on the 16-frame utterance input_timestamp/henan/<ni hao a>.txt
  tests/golden/l2_inputs/                                the reference DATA files those two runs read
        (time stamps, unit tables, key-pose JSONs), verbatim, in the reference's directory layout

Usage:  python tests/golden/make_host_goldens.py
"""
import glob
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
KP_DIR = os.path.join(REF, "*phoneme_data", "VidTIMIT", "fadg0", "keypoints_fadg0")
FRAMES = [0, 1, 17, 38, 64, 101]


def _stub_modules():
    cv2 = types.ModuleType("cv2")
    cv2.circle = lambda *a, **k: None
    cv2.imwrite = lambda *a, **k: True
    sys.modules["cv2"] = cv2
    mp = types.ModuleType("moviepy")
    mpe = types.ModuleType("moviepy.editor")
    sys.modules["moviepy"] = mp
    sys.modules["moviepy.editor"] = mpe
    sp = os.path.join(REF, "venv_vid2vid", "lib", "python3.7", "site-packages", "zhon")
    if os.path.isdir(sp):  # only zhon: nothing else from the py3.7 venv may shadow this env
        d = tempfile.mkdtemp()
        os.symlink(sp, os.path.join(d, "zhon"))
        sys.path.append(d)


def pose_maps():
    sys.path.insert(0, REF)
    import keypoint2img  # the reference's rasteriser
    os.makedirs(os.path.join(HERE, "keypoints_fadg0"), exist_ok=True)
    maps = []
    for f in FRAMES:
        name = "sa1_%03d_keypoints.json" % f
        src = os.path.join(KP_DIR, name)
        shutil.copyfile(src, os.path.join(HERE, "keypoints_fadg0", name))
        maps.append(keypoint2img.read_keypoints(src, (512, 384)))
    # a nested-list face (what the L2 smoother writes) and a partially-invalid pose as extra cases
    with open(os.path.join(KP_DIR, "sa1_005_keypoints.json")) as fh:
        d = json.load(fh)
    d["people"][0]["face_keypoints_2d"] = [d["people"][0]["face_keypoints_2d"]]
    d["people"][0]["pose_keypoints_2d"][2] = 0.0  # nose confidence 0 -> edge [0,1] dropped
    extra = os.path.join(HERE, "keypoints_fadg0", "synthetic_nested_face.json")
    with open(extra, "w") as fh:
        json.dump(d, fh)
    maps.append(keypoint2img.read_keypoints(extra, (512, 384)))
    names = ["sa1_%03d_keypoints.json" % f for f in FRAMES] + ["synthetic_nested_face.json"]
    np.savez_compressed(os.path.join(HERE, "pose_maps_fadg0.npz"), maps=np.stack(maps), names=np.array(names))
    print("pose maps:", np.stack(maps).shape, [int((m != 0).any(2).sum()) for m in maps])


def l2_driver():
    work = tempfile.mkdtemp()
    t2v = os.path.join(work, "Text2Video")
    os.makedirs(t2v)
    for name in ["interp_landmarks_motion_phoneme_VidTIMIT_smooth.py", "keypoint2img.py", "input_timestamp",
                 "*phoneme_data"]:
        os.symlink(os.path.join(REF, name), os.path.join(t2v, name))
    for sub in ["test_openpose/tmp", "test_openpose/tmp_smooth", "test_img/tmp", "test_img/tmp_smooth"]:
        os.makedirs(os.path.join(work, "vid2vid", "datasets", "fadg0", sub))
    cwd = os.getcwd()
    os.chdir(t2v)
    sys.path.insert(0, t2v)
    argv = sys.argv
    sys.argv = ["interp", "She had your dark suit in greasy wash water all year.", "fadg0"]
    try:
        src = open("interp_landmarks_motion_phoneme_VidTIMIT_smooth.py").read()
        exec(compile(src, "interp_landmarks_motion_phoneme_VidTIMIT_smooth.py", "exec"), {"__name__": "__main__"})
    finally:
        sys.argv = argv
        os.chdir(cwd)

    def vec(path):
        p = json.load(open(path))["people"][0]
        return np.concatenate([np.asarray(p["pose_keypoints_2d"], float).ravel(),
                               np.asarray(p["face_keypoints_2d"], float).ravel()])

    out = {}
    for seq in ["tmp", "tmp_smooth"]:
        files = sorted(glob.glob(os.path.join(work, "vid2vid", "datasets", "fadg0", "test_openpose", seq, "*.json")))
        out[seq] = np.stack([vec(f) for f in files])
        out[seq + "_names"] = np.array([os.path.basename(f) for f in files])
    np.savez_compressed(os.path.join(HERE, "l2_driver_Shehadyour.npz"), **out)
    print("l2 driver:", out["tmp"].shape, out["tmp_smooth"].shape)
    # also keep 8 frames of the L2 output as a tiny ready-made dataset for the test.py plumbing test
    ds = os.path.join(HERE, "dataset_fadg0_l2")
    for seq in ["tmp", "tmp_smooth"]:
        os.makedirs(os.path.join(ds, "test_openpose", seq), exist_ok=True)
        files = sorted(glob.glob(os.path.join(work, "vid2vid", "datasets", "fadg0", "test_openpose", seq, "*.json")))
        for f in files[:6]:
            shutil.copyfile(f, os.path.join(ds, "test_openpose", seq, os.path.basename(f)))
    shutil.rmtree(work)


def _run_reference_driver(script, utterance, person, data_links):
    """exec one of the reference's L2 driver scripts in a scratch copy of its directory layout;
    -> {seq: [json paths]} of what it wrote under ../vid2vid/datasets/<person>/test_openpose/."""
    work = tempfile.mkdtemp()
    t2v = os.path.join(work, "Text2Video")
    os.makedirs(t2v)
    for name in [script, "keypoint2img.py", "input_timestamp"] + data_links:
        os.symlink(os.path.join(REF, name), os.path.join(t2v, name))
    for sub in ["test_openpose/tmp", "test_openpose/tmp_smooth", "test_img/tmp", "test_img/tmp_smooth"]:
        os.makedirs(os.path.join(work, "vid2vid", "datasets", person, sub))
    cwd, argv = os.getcwd(), sys.argv
    os.chdir(t2v)
    sys.path.insert(0, t2v)
    sys.argv = ["interp", utterance, person]
    try:
        exec(compile(open(script).read(), script, "exec"), {"__name__": "__main__"})
    finally:
        sys.argv = argv
        os.chdir(cwd)
    out = {seq: sorted(glob.glob(os.path.join(work, "vid2vid", "datasets", person, "test_openpose", seq, "*.json")))
           for seq in ["tmp", "tmp_smooth"]}
    return work, out


def _vec(path):
    p = json.load(open(path))["people"][0]
    return np.concatenate([np.asarray(p["pose_keypoints_2d"], float).ravel(),
                           np.asarray(p["face_keypoints_2d"], float).ravel()])


def l2_inputs_and_pinyin():
    """(1) golden for the pinyin driver (interp_landmarks_motion.py) on a 16-frame utterance;
    (2) the INPUT data files both L2 goldens depend on (time stamps, unit tables, the key-pose JSONs the
    drivers touch), copied verbatim in the reference's directory layout under tests/golden/l2_inputs/ so
    that the parity test of text2video_amd/l2_driver.py runs without /root/reference."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from text2video_amd import l2_driver as L
    work, out = _run_reference_driver("interp_landmarks_motion.py", "你好啊", "henan", ["*pinyin_data", "dict_henan.txt"])
    gold = {}
    for seq, files in out.items():
        gold[seq] = np.stack([_vec(f) for f in files])
        gold[seq + "_names"] = np.array([os.path.basename(f) for f in files])
    np.savez_compressed(os.path.join(HERE, "l2_driver_pinyin_nihaoa.npz"), **gold)
    print("l2 pinyin driver:", gold["tmp"].shape, gold["tmp_smooth"].shape)
    shutil.rmtree(work)
    dst = os.path.join(HERE, "l2_inputs")
    cases = [("She had your dark suit in greasy wash water all year.", "fadg0", L.PHONEME),
             ("你好啊", "henan", L.PINYIN)]
    for text, person, spec in cases:
        bank = L.KeyPoseBank(REF, person, spec)
        L.synthesize(text, person, REF, spec, bank=bank)
        tsp = L.timestamps_path(REF, person, text, spec)
        tables = [tsp, os.path.join(REF, "*phoneme_data", "VidTIMIT", "%s.txt" % person) if spec.kind == "phoneme"
                  else os.path.join(REF, "dict_%s.txt" % person)]
        for src in tables + [os.path.join(bank.dir, n) for n in bank.touched]:
            to = os.path.join(dst, os.path.relpath(src, REF))
            os.makedirs(os.path.dirname(to), exist_ok=True)
            shutil.copyfile(src, to)
        print("l2 inputs:", person, len(bank.touched), "key-pose files")


if __name__ == "__main__":
    _stub_modules()
    pose_maps()
    l2_driver()
    l2_inputs_and_pinyin()
