"""L2 driver against golden vectors captured from the reference's own scripts
(tests/golden/make_host_goldens.py) on the reference's own data files (tests/golden/l2_inputs/)."""
import json
import os

import numpy as np
import pytest

from text2video_amd import l2_driver as L

HERE = os.path.dirname(os.path.abspath(__file__))
INPUTS = os.path.join(HERE, "golden", "l2_inputs")
CASES = [("She had your dark suit in greasy wash water all year.", "fadg0", L.PHONEME, "l2_driver_Shehadyour.npz", 87),
         ("你好啊", "henan", L.PINYIN, "l2_driver_pinyin_nihaoa.npz", 16),
         # a second utterance per driver (tests/golden/make_host_goldens_more.py): other key poses, long gaps, skips
         ("she slipped on the floor", "fadg0", L.PHONEME, "l2_driver_sheslipped.npz", None),
         ("今天天气好极了不冷不", "henan", L.PINYIN, "l2_driver_pinyin_jintiantianqi.npz", 150)]


@pytest.mark.parametrize("case", CASES, ids=["phoneme_fadg0", "pinyin_henan", "phoneme_fadg0_sheslipped", "pinyin_henan_weather"])
def test_sequences_match_reference_bit_for_bit(case):
    text, person, spec, gold, n = case
    g = np.load(os.path.join(HERE, "golden", gold))
    raw, smooth = L.synthesize(text, person, INPUTS, spec)
    a = np.stack([L.pose_vector(j) for j in raw])
    b = np.stack([L.pose_vector(j) for j in smooth])
    n = g["tmp"].shape[0] if n is None else n
    assert a.shape == b.shape == (n, 285) and n >= 16
    assert np.array_equal(a, g["tmp"]), np.abs(a - g["tmp"]).max()
    assert np.array_equal(b, g["tmp_smooth"]), np.abs(b - g["tmp_smooth"]).max()


def test_file_contract(tmp_path):
    """run() lays the dataset out as the reference does, and the pose dataset can consume it."""
    from text2video_amd.keypoints import read_keypoints
    text, person, spec, gold, n = CASES[0]
    L.run(text, person, root=INPUTS, spec=spec, dataset_root=str(tmp_path), log=lambda *a: None)
    g = np.load(os.path.join(HERE, "golden", gold))
    for seq, key in (("tmp", "tmp"), ("tmp_smooth", "tmp_smooth")):
        d = tmp_path / person / "test_openpose" / seq
        names = sorted(os.listdir(d))
        assert names == list(g[key + "_names"])
        vec = np.stack([L.pose_vector(json.load(open(d / f))) for f in names])
        assert np.array_equal(vec, g[key])
        imgs = sorted(os.listdir(tmp_path / person / "test_img" / seq))
        assert len(imgs) == n and imgs[0] == ("0000.jpg" if seq == "tmp" else "smooth_0000.jpg")
    js = json.load(open(tmp_path / person / "test_openpose" / "tmp_smooth" / "smooth_00010.json"))
    face = js["people"][0]["face_keypoints_2d"]
    assert isinstance(face[0], list) and len(face[0]) == 210          # nested one level, as the reference writes it
    img = read_keypoints(str(tmp_path / person / "test_openpose" / "tmp_smooth" / "smooth_00010.json"), (512, 384))
    assert img.shape == (384, 512, 3) and img.any()
    assert np.array_equal(img, read_keypoints(js, (512, 384)))        # parsed dict == file


def test_forked_writers_produce_the_same_files(tmp_path):
    """run() hands the per-frame files (JSON + skeleton JPEG) to forked writer processes; every file equals the in-process
    writer's, byte for byte."""
    text, person, spec, gold, n = CASES[0]
    a, b = tmp_path / "one", tmp_path / "forked"
    L.run(text, person, root=INPUTS, spec=spec, dataset_root=str(a), log=lambda *x: None, workers=1)
    L.run(text, person, root=INPUTS, spec=spec, dataset_root=str(b), log=lambda *x: None, workers=4)
    count = 0
    for dirpath, _, files in os.walk(a):
        for f in files:
            pa = os.path.join(dirpath, f)
            pb = os.path.join(str(b), os.path.relpath(pa, str(a)))
            assert open(pa, "rb").read() == open(pb, "rb").read(), pa
            count += 1
    assert count == 4 * n and count == sum(len(fs) for _, _, fs in os.walk(b))


def test_key_interval_rules():
    ts = [[0, "a"], [2, "b"], [5, "c"], [9, "d"], [10, "e"]]
    # phoneme driver: gap >= 4 keeps the next key; closer keys are jumped over; the last pair is always taken
    assert L.key_intervals(ts, L.PHONEME) == [(0, 2), (2, 3), (3, 4)]
    # pinyin driver: gap > 3
    assert L.key_intervals(ts, L.PINYIN) == [(0, 2), (2, 3), (3, 4)]
    ts2 = [[0, "a"], [4, "b"], [7, "c"], [11, "d"]]
    assert L.key_intervals(ts2, L.PHONEME) == [(0, 1), (1, 3)]        # 4 >= 4 kept; 7 skipped
    assert L.key_intervals(ts2, L.PINYIN) == [(0, 1), (1, 3)]         # 4 > 3 kept
    ts3 = [[0, "a"], [3, "b"], [6, "c"]]
    assert L.key_intervals(ts3, L.PHONEME) == [(0, 2)]                # 3 < 4: jump to the one after
    assert L.key_intervals(ts3 + [[7, "d"]], L.PHONEME) == [(0, 2), (2, 3)]
    assert L.key_intervals([[0, "a"], [1, "b"]], L.PHONEME) == [(0, 1)]
    assert L.key_intervals([[5, "a"]], L.PHONEME) == []


def test_utterance_key():
    assert L.utterance_key("She had your dark suit in greasy wash water all year.", L.PHONEME) == "Shehadyour"
    assert L.utterance_key("今天天气好极了，不冷。", L.PINYIN) == "今天天气好极了不冷"
    assert L.utterance_key("你好啊", L.PINYIN) == "你好啊"
    assert L.utterance_key("a b", L.PINYIN) == "a b"                   # the pinyin driver keeps spaces
    zhon = "/root/reference/venv_vid2vid/lib/python3.7/site-packages/zhon/hanzi.py"
    if os.path.exists(zhon):                                          # build container only: the table itself
        ns = {}
        exec(compile(open(zhon, encoding="utf-8").read(), zhon, "exec"), ns)
        assert set(ns["punctuation"]) == set(L.HANZI_PUNCTUATION)


def test_long_gap_structure_and_smoothing_properties():
    text, person, spec, _, _ = CASES[0]
    bank = L.KeyPoseBank(INPUTS, person, spec)
    ts = L.read_table(L.timestamps_path(INPUTS, person, text, spec))
    frames = L.build_sequence(ts, bank, spec)
    assert sorted(frames) == list(range(int(ts[-1][0]) + 1))
    # the frames before the first key time hold the first key pose
    clip, f = bank.unit[ts[0][1]]
    for t in range(int(ts[0][0])):
        assert frames[t] == bank.pose(clip, f)
    # a synthetic long gap: motion_width+1 frames of each key-pose motion verbatim, linear blend between
    ts_long = [[0, ts[1][1]], [20, ts[2][1]]]
    fr = L.build_sequence(ts_long, bank, spec)
    c1, f1 = bank.unit[ts_long[0][1]]
    c2, f2 = bank.unit[ts_long[1][1]]
    for k in range(4):
        assert fr[k] == bank.pose(c1, f1 + k) and fr[20 - k] == bank.pose(c2, f2 - k)
    a, b, mid = L.pose_vector(fr[3]), L.pose_vector(fr[17]), L.pose_vector(fr[10])
    assert np.allclose(mid, a * 0.5 + b * 0.5, rtol=0, atol=1e-12)
    # smoothing a constant sequence leaves it unchanged (weights are normalised, mouth offset is zero)
    const = [bank.pose(c1, f1) for _ in range(12)]
    sm = L.smooth_sequence(const, spec)
    for js in sm:
        assert np.allclose(L.pose_vector(js), L.pose_vector(const[0]), rtol=0, atol=1e-9)
    assert L.pose_vector(const[0]).shape == (285,) and const[0] == bank.pose(c1, f1)   # input not modified
