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


def default_pose_workers():
    """Rasteriser processes per frame loop.  With the C stamping loops and the direct MINPACK call a pose map costs
    ~2.3 ms of one host core in the bit-exact mode (26 ms in round 1): two workers already keep up with 150 frames/s;
    four leave headroom and still let eight ranks share a host."""
    return max(1, min(4, (os.cpu_count() or 2) - 1))


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

    def restrict(self, units, first_n=None):
        """Keep only the output frames of this rank's work units [(seq, pose_start, pose_stop,
        first_output_index)] (distributed.plan_units); first_n: only the first n output frames of each (the stitch
        pass).  Every unit starts a fresh recurrence; items report the index of their unit."""
        self.items = []
        self._unit_starts = set()
        self._unit_of = {}
        for u, (seq, s, e, first_out) in enumerate(units):
            self._unit_starts.add((seq, first_out))
            stop = e if first_n is None else min(e, first_out + first_n)
            for i in range(first_out, stop):
                self.items.append((seq, i))
                self._unit_of[(seq, i)] = u
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
        src = self.op[seq][i]
        if isinstance(src, str):
            src = os.path.abspath(src)      # the rasteriser workers do not share this process's working directory
        return (src, size, get_img_params(opt, size), not opt.no_pose_crop, opt.remove_face_labels,
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
            workers = default_pose_workers()
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
        # fresh worker interpreters, not forks of this (HIP-initialised) process: raster_pool.py
        from .raster_pool import get_pool
        pool = get_pool(workers)
        futures, cache, nxt = {}, {}, 0

        def pump():
            nonlocal nxt
            while nxt < len(need) and len(futures) < ahead:
                futures[need[nxt]] = pool.submit(self._job(*need[nxt]))
                nxt += 1

        pump()
        try:
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
                yield {"A": np.stack(win), "A_path": self._name(seq, i), "seq": seq, "change_seq": change_seq,
                       "unit": getattr(self, "_unit_of", {}).get((seq, i))}
        finally:
            for f in futures.values():      # an early exit (--how_many, an error): let the in-flight jobs drain
                f.cancel()

    def lane_plan(self, n_lanes, limit=None):
        """The items cut into recurrences (maximal runs of consecutive frames of one unit / sequence) and the
        recurrences dealt to `n_lanes` lanes that advance in lock-step: longest first to the least loaded lane,
        dataset order kept inside a lane.  Returns [[item index, ...] per lane].  `limit` keeps the first `limit` items
        in dataset order (what the single-lane loop's --how_many does) before anything is dealt."""
        n_items = len(self.items) if limit is None else min(limit, len(self.items))
        runs = []
        for idx in range(n_items):
            seq, i = self.items[idx]
            new = idx == 0 or self.items[idx - 1] != (seq, i - 1) or (seq, i) in getattr(self, "_unit_starts", ())
            if new:
                runs.append([])
            runs[-1].append(idx)
        n_lanes = max(1, min(n_lanes, len(runs)))
        loads, lanes = [0] * n_lanes, [[] for _ in range(n_lanes)]
        for r in sorted(range(len(runs)), key=lambda r: (-len(runs[r]), r)):
            k = loads.index(min(loads))
            lanes[k].append(r)
            loads[k] += len(runs[r])
        return [[idx for r in sorted(lane) for idx in runs[r]] for lane in lanes]

    def iter_lanes(self, n_lanes, workers=None, ahead=None, limit=None):
        """Lock-step iteration for N independent recurrences per step (t2v_generator_forward_batch): yields, per step, a
        list of (lane, item) with at most one item per lane -- the same items, windows and names as iter_prefetch,
        only their order differs.  change_seq marks the first frame of every recurrence."""
        plan = self.lane_plan(n_lanes, limit)
        if workers is None:
            workers = default_pose_workers()
        steps = max((len(p) for p in plan), default=0)
        order = [(k, plan[k][t]) for t in range(steps) for k in range(len(plan)) if t < len(plan[k])]
        first_of_run = set()
        for lane in plan:
            for j, idx in enumerate(lane):
                seq, i = self.items[idx]
                if j == 0 or self.items[lane[j - 1]] != (seq, i - 1) or (seq, i) in getattr(self, "_unit_starts", ()):
                    first_of_run.add(idx)
        pool = futures = None
        need, nxt = [], 0
        if workers > 1 and order:
            ahead = ahead or 4 * workers
            seen = set()
            for k, idx in order:        # pose maps in first-use order, per lane (chunks of one sequence overlap by tG-1 maps)
                seq, i = self.items[idx]
                for j in range(i - self.tG + 1, i + 1):
                    if (k, seq, j) not in seen:
                        seen.add((k, seq, j))
                        need.append((k, seq, j))
            from .raster_pool import get_pool
            pool, futures = get_pool(workers), {}

        def pump():
            nonlocal nxt
            while pool is not None and nxt < len(need) and len(futures) < ahead:
                futures[need[nxt]] = pool.submit(self._job(*need[nxt][1:]))
                nxt += 1

        caches = [dict() for _ in plan]
        pump()
        try:
            for t in range(steps):
                out = []
                for k in range(len(plan)):
                    if t >= len(plan[k]):
                        continue
                    idx = plan[k][t]
                    seq, i = self.items[idx]
                    cache, win = caches[k], []
                    for j in range(i - self.tG + 1, i + 1):
                        if (seq, j) not in cache:
                            if pool is not None:
                                cache[(seq, j)] = futures.pop((k, seq, j)).result()
                                pump()
                            else:
                                cache[(seq, j)] = self._pose_map(seq, j)
                        win.append(cache[(seq, j)])
                    for key in [q for q in cache if q[0] != seq or q[1] < i - self.tG + 2]:
                        del cache[key]
                    out.append((k, {"A": np.stack(win), "A_path": self._name(seq, i), "seq": seq,
                                    "change_seq": idx in first_of_run,
                                    "unit": getattr(self, "_unit_of", {}).get((seq, i))}))
                yield out
        finally:
            if futures:
                for f in futures.values():      # an early exit: let the in-flight jobs drain
                    f.cancel()

    def __getitem__(self, idx):
        seq, i = self.items[idx]
        change_seq = idx == 0 or self.items[idx - 1][0] != seq or (seq, i) in getattr(self, "_unit_starts", ())
        if not change_seq and self._window_key == (seq, i - 1):
            self._window = self._window[1:] + [self._pose_map(seq, i)]
        else:
            self._window = [self._pose_map(seq, j) for j in range(i - self.tG + 1, i + 1)]
        self._window_key = (seq, i)
        return {"A": np.stack(self._window), "A_path": self._name(seq, i), "seq": seq, "change_seq": change_seq,
                "unit": getattr(self, "_unit_of", {}).get((seq, i))}

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


# ------------------------------------------------------------------------------------------------
# Training side (README.md:171-176 recipe: --dataset_mode pose --resize_or_crop
# randomScaleHeight_and_scaledCrop --loadSize 544 --fineSize 512 --n_frames_total 12 --max_t_step 4
# --random_drop_prob 0).  The sampling rules are upstream vid2vid's base_dataset.get_img_params /
# get_video_params [RECALL: un-vendored, SURVEY App. F]; nothing in the reference tree pins them.
# ------------------------------------------------------------------------------------------------
def make_power_2(n, base=32):
    return max(base, int(round(n / base)) * base)


def get_train_img_params(opt, size, rng):
    """One random scale + crop for a whole clip.  -> dict(new_size=(w,h), crop_size=(w,h), crop_pos=(x,y))."""
    w, h = size
    mode = opt.resize_or_crop
    new_w, new_h = w, h
    if "resize" in mode:
        new_h = new_w = opt.loadSize
    elif "randomScaleWidth" in mode:
        new_w = int(rng.integers(opt.fineSize, opt.loadSize + 1))
        new_h = new_w * h // w
    elif "randomScaleHeight" in mode:
        new_h = int(rng.integers(opt.fineSize, opt.loadSize + 1))
        new_w = new_h * w // h
    elif "scaleWidth" in mode:
        new_w, new_h = opt.loadSize, opt.loadSize * h // w
    elif "scaleHeight" in mode:
        new_h, new_w = opt.loadSize, opt.loadSize * w // h
    new_w, new_h = int(round(new_w / 4)) * 4, int(round(new_h / 4)) * 4
    crop_x = crop_y = crop_w = crop_h = 0
    if "crop" in mode or "scaledCrop" in mode:
        if "scaledCrop" in mode:
            if "Width" in mode:
                crop_w, crop_h = opt.fineSize, opt.fineSize * h // w
            else:
                crop_h, crop_w = opt.fineSize, opt.fineSize * w // h
        else:
            crop_w = crop_h = opt.fineSize
        crop_w, crop_h = min(make_power_2(crop_w), new_w // 4 * 4), min(make_power_2(crop_h), new_h // 4 * 4)
        x_span = (new_w - crop_w) // 2
        crop_x = int(max(0, min(x_span * 2, int(rng.standard_normal() * x_span / 3 + x_span))))
        crop_y = int(rng.integers(0, min(max(0, new_h - crop_h), new_h // 8) + 1))
    else:
        new_w, new_h = make_power_2(new_w), make_power_2(new_h)
        crop_w, crop_h = new_w, new_h
    return {"new_size": (new_w, new_h), "crop_size": (crop_w, crop_h), "crop_pos": (crop_x, crop_y)}


def get_video_params(opt, n_frames_total, seq_len, rng):
    """-> (n_frames incl. the tG-1 warm-up frames, start index, frame step)."""
    tG = opt.n_frames_G
    n = min(n_frames_total, seq_len - tG + 1) + tG - 1
    max_t_step = max(1, min(opt.max_t_step, (seq_len - 1) // max(1, n - 1)))
    t_step = int(rng.integers(1, max_t_step + 1))
    offset_max = max(1, seq_len - (n - 1) * t_step)
    return n, int(rng.integers(0, offset_max)), t_step


class TrainPoseDataset:
    """<dataroot>/train_openpose/<seq>/*.json + <dataroot>/train_img/<seq>/*.jpg (one image per JSON).
    sample(i) -> a clip: pose maps A and frames B as uint8 [T,H,W,3] with ONE augmentation for the clip."""

    def __init__(self, opt, seed=0):
        self.opt = opt
        self.op = {s: [f for f in fs if f.endswith(".json")]
                   for s, fs in _seq_dirs(os.path.join(opt.dataroot, "train_openpose")).items()}
        self.img = {s: [f for f in fs if f.lower().endswith(IMG_EXT)]
                    for s, fs in _seq_dirs(os.path.join(opt.dataroot, "train_img")).items()}
        self.seqs = [s for s in self.op if self.op[s]]
        if not self.seqs:
            raise FileNotFoundError("no training sequences under %s" % os.path.join(opt.dataroot, "train_openpose"))
        for s in self.seqs:
            if len(self.img.get(s, [])) != len(self.op[s]):
                raise ValueError("sequence %s: %d pose files vs %d images" % (s, len(self.op[s]), len(self.img.get(s, []))))
            if len(self.op[s]) < opt.n_frames_G:
                raise ValueError("sequence %s is shorter than n_frames_G=%d" % (s, opt.n_frames_G))
        self.rng = np.random.default_rng(seed)
        self.n_frames_total = opt.n_frames_total

    def __len__(self):
        return len(self.seqs)

    def update_training_batch(self, ratio):
        """upstream doubles the clip length every --niter_step epochs: n_frames_total * 2**ratio (capped)."""
        self.n_frames_total = min(getattr(self.opt, "max_frames_total", 10 ** 9), self.opt.n_frames_total * (2 ** ratio))

    def sample(self, index):
        opt, rng = self.opt, self.rng
        seq = self.seqs[index % len(self.seqs)]
        n, start, step = get_video_params(opt, self.n_frames_total, len(self.op[seq]), rng)
        with Image.open(self.img[seq][0]) as im:
            size = im.size
        prm = get_train_img_params(opt, size, rng)
        (nw, nh), (cw, ch), (cx, cy) = prm["new_size"], prm["crop_size"], prm["crop_pos"]
        A, B = [], []
        for i in range(n):
            t = start + i * step
            a = keypoints.read_keypoints(self.op[seq][t], size, opt.random_drop_prob, opt.remove_face_labels,
                                         opt.basic_point_only, exact_fit=not opt.fast_pose,
                                         hand_discs=not opt.no_hand_discs, rng=rng)
            a = Image.fromarray(a).resize((nw, nh), Image.NEAREST)
            with Image.open(self.img[seq][t]) as im:
                b = im.convert("RGB").resize((nw, nh), Image.BICUBIC)
            box = (cx, cy, cx + cw, cy + ch)
            A.append(np.asarray(a.crop(box)))
            B.append(np.asarray(b.crop(box)))
        return {"A": np.stack(A), "B": np.stack(B), "seq": seq, "start": start, "t_step": step, "params": prm}
