"""Pose dataset of the inference path (section 8a row a2).

Directory contract written by the reference's L2 driver
(/root/reference/interp_landmarks_motion_phoneme_VidTIMIT_smooth.py:31-37,86,220,260,266):
    <dataroot>/test_openpose/<seq>/*.json      OpenPose keypoints, one file per frame
    <dataroot>/test_img/<seq>/*.jpg            only their SIZE and NAMES matter with --no_first_img
Per item (one output frame): rasterise the newest JSON (keypoints.read_keypoints), NEAREST-resize to
the `scaleHeight` geometry, central-width crop, keep a sliding window of n_frames_G maps.  Maps stay
uint8 HWC: ToTensor + Normalize(.5,.5) run on the GPU (t2v_pose_u8_to_f32).
"""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
from PIL import Image

from . import keypoints

IMG_EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".tiff")


def _seq_dirs(root):
    if not os.path.isdir(root):
        return {}
    out = {}
    for d in sorted(os.listdir(root)):
        p = os.path.join(root, d)
        if os.path.isdir(p):
            files = sorted(f for f in os.listdir(p) if not f.startswith("."))
            if files:
                out[d] = [os.path.join(p, f) for f in files]
    return out


def get_img_params(opt, size):
    """(w, h) of the source -> (new_w, new_h); upstream `get_img_params` for the modes on the path."""
    w, h = size
    mode = opt.resize_or_crop
    if "scaleHeight" in mode:
        new_h, new_w = opt.loadSize, opt.loadSize * w // h
    elif "scaleWidth" in mode:
        new_w, new_h = opt.loadSize, opt.loadSize * h // w
    elif "resize" in mode:
        new_w = new_h = opt.loadSize
    else:
        new_w, new_h = w, h
    return int(round(new_w / 4)) * 4, int(round(new_h / 4)) * 4


def central_crop_cols(w):
    """upstream PoseDataset.crop: keep 2*(int(0.25*w)//32*32) central columns"""
    bs = int(w * 0.25) // 32 * 32
    return w // 2 - bs, w // 2 + bs


def _render_job(job):
    """top-level (picklable) worker: rasterise + resize + crop one pose JSON -> uint8 [H,W,3]"""
    (path, size, new_size, crop, remove_face_labels, basic_point_only, exact_fit, hand_discs) = job
    m = keypoints.read_keypoints(path, size, 0, remove_face_labels, basic_point_only, exact_fit=exact_fit,
                                 hand_discs=hand_discs)
    if new_size != size:
        m = np.asarray(Image.fromarray(m).resize(new_size, Image.NEAREST))
    if crop:
        c0, c1 = central_crop_cols(m.shape[1])
        m = m[:, c0:c1]
    return np.ascontiguousarray(m)


class PoseDataset:
    def __init__(self, opt):
        self.opt = opt
        self.tG = opt.n_frames_G
        phase = getattr(opt, "phase", "test")
        self.op = _seq_dirs(os.path.join(opt.dataroot, phase + "_openpose"))
        self.img = _seq_dirs(os.path.join(opt.dataroot, phase + "_img"))
        if not self.op:
            raise FileNotFoundError("no pose sequences under %s" % os.path.join(opt.dataroot, phase + "_openpose"))
        self.items = []  # (seq, index of newest frame)
        for seq, files in self.op.items():
            files = [f for f in files if f.endswith(".json")]
            self.op[seq] = files
            imgs = [f for f in self.img.get(seq, []) if f.lower().endswith(IMG_EXT)]
            if imgs and len(imgs) != len(files):
                raise ValueError("sequence %s: %d pose files vs %d images" % (seq, len(files), len(imgs)))
            self.img[seq] = imgs
            for i in range(self.tG - 1 + getattr(opt, "start_frame", 0), len(files)):
                self.items.append((seq, i))
        self._window = None
        self._window_key = None
        self._sizes = {}

    @classmethod
    def from_memory(cls, opt, sequences, size=(512, 384), names=None):
        """Dataset over in-memory OpenPose frames: {sequence name: [parsed JSON dict per frame]} -- what
        l2_driver.synthesize() returns -- rasterised on a `size` = (w, h) canvas.  Same windows, same
        recurrence resets and same output naming as the directory form (names default to the files the
        reference's L2 driver would have written)."""
        self = cls.__new__(cls)
        self.opt, self.tG = opt, opt.n_frames_G
        self.op = {seq: list(frames) for seq, frames in sequences.items()}
        self.img = {seq: [] for seq in self.op}
        stem = {"tmp": "%04d.jpg", "tmp_smooth": "smooth_%04d.jpg"}   # the skeleton images the L2 driver writes
        self.names = {seq: list((names or {}).get(seq) or [os.path.join(seq, stem.get(seq, "%04d.jpg") % i)
                                                           for i in range(len(fr))])
                      for seq, fr in self.op.items()}
        self.items = [(seq, i) for seq, fr in self.op.items()
                      for i in range(self.tG - 1 + getattr(opt, "start_frame", 0), len(fr))]
        self._window = self._window_key = None
        self._sizes = {seq: tuple(size) for seq in self.op}
        return self

    def _name(self, seq, i):
        if getattr(self, "names", None):
            return self.names[seq][i]
        return self.img[seq][i] if self.img[seq] else self.op[seq][i]

    def __len__(self):
        return len(self.items)

    def seq_lengths(self):
        """{sequence: number of pose maps} -- input of distributed.assign_chunks"""
        return {seq: len(files) for seq, files in self.op.items()}

    def restrict(self, units):
        """Keep only the output frames of this rank's work units [(seq, pose_start, pose_stop,
        first_output_index)] (distributed.assign_chunks).  Every unit starts a fresh recurrence."""
        self.items = []
        self._unit_starts = set()
        for seq, s, e, first_out in units:
            self._unit_starts.add((seq, first_out))
            self.items += [(seq, i) for i in range(first_out, e)]
        self._window = self._window_key = None

    def _size(self, seq):
        if self.img[seq]:
            with Image.open(self.img[seq][0]) as im:
                return im.size
        return (512, 384)  # the reference L2 driver's canvas (interp_...smooth.py:72-73)

    def _job(self, seq, i):
        opt = self.opt
        if seq not in self._sizes:
            self._sizes[seq] = self._size(seq)
        size = self._sizes[seq]
        return (self.op[seq][i], size, get_img_params(opt, size), not opt.no_pose_crop, opt.remove_face_labels,
                opt.basic_point_only, not opt.fast_pose, not opt.no_hand_discs)

    def _pose_map(self, seq, i):
        return _render_job(self._job(seq, i))

    def iter_prefetch(self, workers=None, ahead=None, limit=None):
        """Same items as iteration, with the pose maps rasterised ahead of time by a process pool.
        The reference rasterises inside the (single) data-loader worker at ~65 ms/frame (SURVEY 8f
        rank 1), which would cap the pipeline at ~15 frames/s; every pose map is independent, so
        `workers` processes keep the host ahead of the GPU while the maps stay bit-identical."""
        n_items = len(self.items) if limit is None else min(limit, len(self.items))
        if workers is None:
            workers = max(1, min(16, (os.cpu_count() or 2) - 1))
        if workers <= 1 or n_items == 0:
            for idx in range(n_items):
                yield self[idx]
            return
        ahead = ahead or 4 * workers
        need, seen = [], set()      # pose maps in first-use order
        for idx in range(n_items):
            seq, i = self.items[idx]
            for j in range(i - self.tG + 1, i + 1):
                if (seq, j) not in seen:
                    seen.add((seq, j))
                    need.append((seq, j))
        with ProcessPoolExecutor(max_workers=workers) as pool:
            futures, cache, nxt = {}, {}, 0

            def pump():
                nonlocal nxt
                while nxt < len(need) and len(futures) < ahead:
                    futures[need[nxt]] = pool.submit(_render_job, self._job(*need[nxt]))
                    nxt += 1

            pump()
            for idx in range(n_items):
                seq, i = self.items[idx]
                win = []
                for j in range(i - self.tG + 1, i + 1):
                    if (seq, j) not in cache:
                        cache[(seq, j)] = futures.pop((seq, j)).result()
                        pump()
                    win.append(cache[(seq, j)])
                for key in [k for k in cache if k[0] != seq or k[1] < i - self.tG + 2]:
                    del cache[key]
                change_seq = idx == 0 or self.items[idx - 1][0] != seq or \
                    (seq, i) in getattr(self, "_unit_starts", ())
                yield {"A": np.stack(win), "A_path": self._name(seq, i), "seq": seq, "change_seq": change_seq}

    def __getitem__(self, idx):
        seq, i = self.items[idx]
        change_seq = idx == 0 or self.items[idx - 1][0] != seq or (seq, i) in getattr(self, "_unit_starts", ())
        if not change_seq and self._window_key == (seq, i - 1):
            self._window = self._window[1:] + [self._pose_map(seq, i)]
        else:
            self._window = [self._pose_map(seq, j) for j in range(i - self.tG + 1, i + 1)]
        self._window_key = (seq, i)
        return {"A": np.stack(self._window), "A_path": self._name(seq, i), "seq": seq, "change_seq": change_seq}

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
