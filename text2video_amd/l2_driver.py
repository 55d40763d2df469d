"""L2 driver: phoneme / pinyin time stamps -> key-pose lookup -> interpolated + smoothed OpenPose sequences.

SURVEY.md section 8f rank 3: the caller on the input side of the frame path.  Restates what the
reference's `interp_landmarks_motion_phoneme_VidTIMIT_smooth.py` (English, VidTIMIT key poses;
[REF :45-267]) and `interp_landmarks_motion.py` (Mandarin pinyin; same algorithm, different tables and
thresholds) compute, as functions over in-memory sequences instead of a script over ~600 JSON files:

  time stamps  [(frame, unit)]                       [REF :45, input_timestamp/<person>/phones/<key>.txt]
  unit table   unit -> (clip, frame) of its key pose [REF :47-57, *phoneme_data/VidTIMIT/<person>.txt]
  key poses    OpenPose JSON per (clip, frame)        [REF :26, keypoints_<person>/]

  build_sequence():  between consecutive key times the 25-point pose and 70-point face are either a
      cross-fade of the two key-pose *motions* (short gaps, [REF :150-172]) or 3 frames of each motion
      joined by a linear blend (long gaps, [REF :175-202]); key times closer than `min_key_dist` are
      skipped [REF :128-145].
  smooth_sequence(): causal-recursive 8-tap smoothing of face + pose with the mouth (points 48..67)
      re-attached rigidly at the smoothed mouth centre [REF :218-262].

Everything is float64 with the reference's operation order, so the result is bit-identical to the
JSON the reference writes (tests/test_cpu_l2_driver.py against tests/golden/l2_driver_*.npz, captured
from the reference itself by tests/golden/make_host_goldens.py).

`run()` keeps the reference's file contract (`../vid2vid/datasets/<person>/test_{openpose,img}/{tmp,tmp_smooth}`)
so that `text2video_audio.sh` works unchanged; `synthesize()` is the in-memory variant: no disk round
trip between the driver and the pose dataset.
"""
import copy
import json
import os
from dataclasses import dataclass

import numpy as np

# zhon.hanzi.punctuation (the reference strips these from the utterance to form the time-stamp file name,
# [REF :19-23]); CJK punctuation only -- ASCII punctuation stays.
HANZI_PUNCTUATION = ("＂＃＄％＆＇（）＊＋，－／：；＜＝＞＠［＼］＾＿｀｛｜｝～｟｠｢｣､　、〃〈〉《》「」『』【】〔〕〖〗〘〙〚〛〜〝〞〟"
                     "〰〾〿–—‘’‛“”„‟…‧﹏﹑﹔·！？｡。")

FACE, POSE = "face_keypoints_2d", "pose_keypoints_2d"


@dataclass(frozen=True)
class L2Spec:
    """The two reference drivers differ only in these."""
    kind: str                   # "phoneme" (VidTIMIT) | "pinyin"
    min_key_dist: int           # key times closer than this are skipped ...
    inclusive: bool             # ... `>=` (phoneme driver) or `>` (pinyin driver)
    motion_width: int = 3
    transition_width: int = 5
    smooth_width: int = 4


PHONEME = L2Spec("phoneme", 4, True)      # [REF interp_landmarks_motion_phoneme_VidTIMIT_smooth.py:70-76]
PINYIN = L2Spec("pinyin", 3, False)       # [REF interp_landmarks_motion.py:55-60]

_PINYIN_CANVAS = {"xuesong": (1280, 720), "henan": (1920, 1080)}   # [REF interp_landmarks_motion.py:63-68]


def canvas_size(spec, person):
    """(width, height) the skeleton images are rasterised on."""
    return (512, 384) if spec.kind == "phoneme" else _PINYIN_CANVAS.get(person, (1280, 720))


def utterance_key(text, spec):
    """Time-stamp file stem: first 10 characters of the utterance without CJK punctuation (and, for the
    phoneme driver, without spaces)."""
    s = text.replace(" ", "") if spec.kind == "phoneme" else text
    s = "".join(ch for ch in s if ch not in HANZI_PUNCTUATION)
    return s[:10]


def read_table(path):
    """whitespace-separated rows of strings (what np.genfromtxt(dtype=str) yields, minus numpy)."""
    rows = []
    with open(path, encoding="utf-8") as fh:
        for line in fh:
            parts = line.split()
            if parts and not parts[0].startswith("#"):
                rows.append(parts)
    return rows


class KeyPoseBank:
    """unit -> key pose, and (clip, frame) -> OpenPose JSON.  `root` is the reference's working directory."""

    def __init__(self, root, person, spec, loader=None):
        self.spec, self.person = spec, person
        self._cache = {}
        self.touched = []                      # files read, in first-use order (fixture capture)
        if spec.kind == "phoneme":
            self.dir = os.path.join(root, "*phoneme_data", "VidTIMIT", person, "keypoints_%s" % person)
            rows = read_table(os.path.join(root, "*phoneme_data", "VidTIMIT", "%s.txt" % person))
            self.unit = {r[0]: (r[1], int(r[2])) for r in rows}          # AA0 sa1 038
        else:
            self.dir = os.path.join(root, "*pinyin_data", person, "keypoints_%s" % person)
            rows = read_table(os.path.join(root, "dict_%s.txt" % person))
            self.unit = {r[0]: ("", int(r[1])) for r in rows}            # ba 50
        self._loader = loader or self._load_file

    def file_name(self, clip, frame):
        return ("%s_%03d_keypoints.json" % (clip, frame)) if self.spec.kind == "phoneme" else "%05d_keypoints.json" % frame

    def _load_file(self, name):
        with open(os.path.join(self.dir, name)) as fh:
            return json.load(fh)

    def pose(self, clip, frame):
        """a fresh copy of the key-pose JSON (callers edit it)."""
        name = self.file_name(clip, frame)
        if name not in self._cache:
            self._cache[name] = self._loader(name)
            self.touched.append(name)
        return copy.deepcopy(self._cache[name])


def timestamps_path(root, person, text, spec):
    key = utterance_key(text, spec)
    sub = ("phones", "%s.txt" % key) if spec.kind == "phoneme" else ("%s.txt" % key,)
    return os.path.join(root, "input_timestamp", person, *sub)


def _blend(a, wa, b, wb):
    # x1*w1 + x2*w2 element by element, float64: the reference's list comprehension [REF :96,:100]
    return (np.asarray(a, np.float64) * wa + np.asarray(b, np.float64) * wb).tolist()


def _person(js):
    return js["people"][0]


def key_intervals(ts, spec):
    """Pairs of key indices (i, j) the sequence is built between [REF :120-145]: the next key time, unless
    it is too close -- then the one after it (the last pair is always taken)."""
    out, i, n = [], 0, len(ts)
    while i < n - 1:
        gap = int(ts[i + 1][0]) - int(ts[i][0])
        far = gap >= spec.min_key_dist if spec.inclusive else gap > spec.min_key_dist
        if far:
            out.append((i, i + 1))
            i += 1
        elif i == n - 2:
            out.append((i, i + 1))
            i += 2
        else:
            out.append((i, i + 2))
            i += 2
    return out


def build_sequence(ts, bank, spec):
    """-> {frame index: OpenPose JSON dict} of the raw ("tmp") sequence."""
    mw = spec.motion_width
    first_t, last_t = int(ts[0][0]), int(ts[-1][0])
    frames = {}
    lead = bank.pose(*bank.unit[ts[0][1]])
    for t in range(first_t):                                       # hold the first key pose [REF :81-88]
        frames[t] = copy.deepcopy(lead)
    # the short-gap branch writes into ONE template (the first key pose's JSON): hands etc. stay the
    # first key pose's, only face and pose are replaced [REF :116-117,:163-171]
    template = copy.deepcopy(lead)
    for i, j in key_intervals(ts, spec):
        t1, t2 = int(ts[i][0]), int(ts[j][0])
        clip1, f1 = bank.unit[ts[i][1]]
        clip2, f2 = bank.unit[ts[j][1]]
        span = float(t2 - t1)
        if span - 1 < 2 * mw + spec.transition_width:
            # cross-fade of the two key-pose motions over the whole gap
            for t in range(t1, t2 + 1):
                w2 = float(t - t1) / span
                w1 = 1.0 - w2
                a, b = _person(bank.pose(clip1, f1 + t - t1)), _person(bank.pose(clip2, f2 + t - t2))
                tp = _person(template)
                tp[FACE] = _blend(a[FACE], w1, b[FACE], w2)
                tp[POSE] = _blend(a[POSE], w1, b[POSE], w2)
                frames[t] = copy.deepcopy(template)
        else:
            # mw+1 frames of each motion verbatim, linear blend of their inner ends in between
            for t in range(t1, t1 + mw + 1):
                head = bank.pose(clip1, f1 + t - t1)
                frames[t] = head
            for t in range(t2, t2 - mw - 1, -1):
                tail = bank.pose(clip2, f2 + t - t2)
                frames[t] = tail
            inner = t2 - mw - (t1 + mw)
            for t in range(t1 + mw + 1, t2 - mw):
                w2 = float(t - (t1 + mw)) / float(inner)
                w1 = 1.0 - w2
                mid = copy.deepcopy(head)
                mp = _person(mid)
                mp[FACE] = _blend(_person(head)[FACE], w1, _person(tail)[FACE], w2)
                mp[POSE] = _blend(_person(head)[POSE], w1, _person(tail)[POSE], w2)
                frames[t] = mid
    # [REF :204-210] would hold the last key pose for range(last+1, total) with total == last: empty
    assert sorted(frames) == list(range(last_t + 1)), "L2 driver: gap in the frame sequence"
    return frames


def _mouth_centre(face):
    return np.average(np.asarray(face, np.float64).reshape(70, 3)[48:60, :], axis=0)


def smooth_sequence(frames, spec):
    """frames: list of OpenPose JSON dicts (time order).  -> new list ("tmp_smooth").
    Window s in [-w, w) with weights 1/(|s|+1); frames behind the current one are the already smoothed
    ones (the reference smooths its list in place), so the filter is recursive."""
    seq = copy.deepcopy(frames)
    sw, n = spec.smooth_width, len(seq)
    for t in range(n):
        sum_w = 0.0
        sum_fc = np.zeros((1, 210), dtype=np.float64)
        sum_ps = np.zeros((1, 75), dtype=np.float64)
        for s in range(-sw, sw):
            k = t + s
            if 0 <= k < n:
                wt = 1.0 / (abs(s) + 1.0)
                sum_fc += np.asarray(_person(seq[k])[FACE], np.float64) * wt
                sum_ps += np.asarray(_person(seq[k])[POSE], np.float64) * wt
                sum_w += wt
        ave_fc, ave_ps = sum_fc / sum_w, sum_ps / sum_w
        own = np.asarray(_person(seq[t])[FACE], np.float64).reshape(-1)
        shift = _mouth_centre(ave_fc) - _mouth_centre(own)
        for p in range(48, 68):                                   # the mouth keeps its shape, moves as a block
            own[p * 3] = own[p * 3] + shift[0]
            own[p * 3 + 1] = own[p * 3 + 1] + shift[1]
        ave_fc[0, 48 * 3:68 * 3] = own[48 * 3:68 * 3]
        _person(seq[t])[FACE] = ave_fc.tolist()                   # nested one level, like the reference's files
        _person(seq[t])[POSE] = ave_ps.tolist()
    return seq


def pose_vector(js):
    """[pose 75 | face 210] of one frame as float64."""
    p = _person(js)
    return np.concatenate([np.asarray(p[POSE], np.float64).ravel(), np.asarray(p[FACE], np.float64).ravel()])


def synthesize(text, person, root=".", spec=PHONEME, ts=None, bank=None):
    """In-memory L2: -> (tmp frames, tmp_smooth frames), lists of OpenPose JSON dicts."""
    ts = ts if ts is not None else read_table(timestamps_path(root, person, text, spec))
    bank = bank or KeyPoseBank(root, person, spec)
    frames = build_sequence(ts, bank, spec)
    raw = [frames[t] for t in sorted(frames)]
    return raw, smooth_sequence(raw, spec)


def _write_frame(job):
    """one frame's two files: the OpenPose JSON and (write_images) its skeleton JPEG"""
    pose_path, img_path, js, size = job
    with open(pose_path, "w") as fh:
        json.dump(js, fh)
    if img_path is not None:
        from PIL import Image
        from .keypoints import read_keypoints
        img = read_keypoints(js, size)
        # cv2.imwrite stores the array as BGR; with --no_first_img only the image SIZE is used downstream
        Image.fromarray(np.ascontiguousarray(img[..., ::-1])).save(img_path)


def run(text, person, root=".", spec=PHONEME, dataset_root=None, write_images=True, log=print, workers=None):
    """The reference's file contract: JSON (+ skeleton JPEG) per frame under
    <dataset_root>/<person>/test_{openpose,img}/{tmp,tmp_smooth}/ [REF :28-37,:213-267].
    The frames' files are independent of each other (6 ms of rasterising + JPEG + JSON each, 2 x 87 of them for the configs[0]
    utterance = 1.1 s of this script's 1.4 s): they are written by `workers` forked processes (default: up to 8; this process
    holds no HIP runtime, so a fork costs nothing).  Same files, same bytes (tests/test_cpu_l2_driver.py)."""
    dataset_root = dataset_root or os.path.join(root, "..", "vid2vid", "datasets")
    base = os.path.join(dataset_root, person)
    raw, smooth = synthesize(text, person, root, spec)
    log("total_frame_num %d" % (len(raw) - 1))
    size = canvas_size(spec, person)
    digits = 4 if spec.kind == "phoneme" else 5
    jobs = []
    for seq, frames, stem in (("tmp", raw, ""), ("tmp_smooth", smooth, "smooth_")):
        pose_dir = os.path.join(base, "test_openpose", seq)
        img_dir = os.path.join(base, "test_img", seq)
        os.makedirs(pose_dir, exist_ok=True)
        os.makedirs(img_dir, exist_ok=True)
        for t, js in enumerate(frames):
            jobs.append((os.path.join(pose_dir, "%s%05d.json" % (stem, t)),
                         os.path.join(img_dir, "%s%s.jpg" % (stem, str(t).zfill(digits))) if write_images else None, js, size))
    if workers is None:
        try:
            workers = min(8, len(os.sched_getaffinity(0)))
        except (AttributeError, OSError):
            workers = 1
    _t = __import__("sys").modules.get("torch")
    if _t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized():
        workers = 1      # a live HIP runtime in this process: a fork would copy (and tear down) its mappings per worker
    if workers > 1 and len(jobs) >= 4 * workers:
        import multiprocessing as mp
        from . import keypoints
        keypoints._minpack()                  # resolved once here, inherited by the forks
        keypoints._host_lib()
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_write_frame, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
    else:
        for job in jobs:
            _write_frame(job)
    return raw, smooth


def main(argv=None, spec=PHONEME):
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        raise SystemExit("usage: <utterance> <person>")
    run(argv[0], argv[1], root=".", spec=spec)


if __name__ == "__main__":
    main()
