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

    def _pose_map(self, seq, i):
        opt = self.opt
        size = self._size(seq)
        m = keypoints.read_keypoints(self.op[seq][i], size, 0, opt.remove_face_labels, opt.basic_point_only,
                                     exact_fit=not opt.fast_pose, hand_discs=not opt.no_hand_discs)
        nw, nh = get_img_params(opt, size)
        if (nw, nh) != size:
            m = np.asarray(Image.fromarray(m).resize((nw, nh), Image.NEAREST))
        if not opt.no_pose_crop:
            c0, c1 = central_crop_cols(m.shape[1])
            m = m[:, c0:c1]
        return np.ascontiguousarray(m)

    def __getitem__(self, idx):
        seq, i = self.items[idx]
        change_seq = idx == 0 or self.items[idx - 1][0] != seq or (seq, i) in getattr(self, "_unit_starts", ())
        if not change_seq and self._window_key == (seq, i - 1):
            self._window = self._window[1:] + [self._pose_map(seq, i)]
        else:
            self._window = [self._pose_map(seq, j) for j in range(i - self.tG + 1, i + 1)]
        self._window_key = (seq, i)
        name_src = self.img[seq][i] if self.img[seq] else self.op[seq][i]
        return {"A": np.stack(self._window), "A_path": name_src, "seq": seq, "change_seq": change_seq}

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
