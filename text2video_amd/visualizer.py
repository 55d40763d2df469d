"""Result writer with the reference's output layout (SURVEY.md App. F `Visualizer.save_images`):
    <results_dir>/<name>/<phase>_<which_epoch>/<seq>/<label>_<basename of the test_img file>.jpg
which is what text2video_audio.sh:39-40 cleans and image2video*.py globs (`fake_B_*.jpg`).
JPEG encoding runs on a small thread pool so it overlaps the GPU work of the next frames."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image


def tensor2im_np(x_chw):
    """util.tensor2im: float CHW in [-1,1] -> uint8 HWC (truncating cast, like numpy astype)."""
    a = (np.transpose(np.asarray(x_chw, np.float32), (1, 2, 0)) + 1) / 2.0 * 255.0
    return np.clip(a, 0, 255).astype(np.uint8)


class Visualizer:
    def __init__(self, opt, workers=4):
        self.save_dir = os.path.join(opt.results_dir, opt.name, "%s_%s" % (opt.phase, opt.which_epoch))
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.pending = []
        self._made = set()

    @staticmethod
    def _write(path, arr):
        Image.fromarray(arr).save(path)

    def save_images(self, visuals, a_path):
        """visuals: {label: uint8 HWC array}.  Returns the written paths."""
        sub = os.path.basename(os.path.dirname(a_path))
        name = os.path.splitext(os.path.basename(a_path))[0]
        d = os.path.join(self.save_dir, sub)
        if d not in self._made:
            os.makedirs(d, exist_ok=True)
            self._made.add(d)
        out = []
        for label, arr in visuals.items():
            path = os.path.join(d, "%s_%s.jpg" % (label, name))
            self.pending.append(self.pool.submit(self._write, path, np.ascontiguousarray(arr)))
            out.append(path)
        return out

    def flush(self):
        for f in self.pending:
            f.result()
        self.pending = []

    def close(self):
        """flush and release the encoder threads (a resident server builds one Visualizer per request)"""
        self.flush()
        self.pool.shutdown(wait=True)
