"""Golden maps for the training-time options of the reference rasteriser (random_drop_prob > 0, remove_face_labels):
/root/reference/keypoint2img.py:113-123 jitters the head key points and the face and drops limbs / hands / the face
with draws from the GLOBAL np.random stream.  Runs only in the build container (imports the reference from where it
lies, cv2.circle stubbed as in make_host_goldens.py); the committed fixture holds inputs' names, seeds and the
expected uint8 maps.

Usage: python tests/golden/make_jitter_golden.py
"""
import glob
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

cv2 = types.ModuleType("cv2")
cv2.circle = lambda *a, **k: None
cv2.imwrite = lambda *a, **k: True
sys.modules["cv2"] = cv2
sys.path.insert(0, REF)
import keypoint2img  # noqa: E402

cases, maps = [], []
for frame in (0, 17):
    name = "sa1_%03d_keypoints.json" % frame
    src = os.path.join(HERE, "keypoints_fadg0", name)
    assert os.path.exists(src), src
    for seed in (1, 2):
        for prob, remove in ((0.3, True), (0.3, False), (0.0, True)):
            np.random.seed(seed)
            maps.append(keypoint2img.read_keypoints(src, (512, 384), prob, remove))
            cases.append((name, seed, prob, int(remove)))
np.savez_compressed(os.path.join(HERE, "pose_maps_jitter.npz"), maps=np.stack(maps),
                    names=np.array([c[0] for c in cases]), seeds=np.array([c[1] for c in cases]),
                    probs=np.array([c[2] for c in cases]), remove=np.array([c[3] for c in cases]))
print("jitter goldens:", np.stack(maps).shape, [int((m != 0).any(2).sum()) for m in maps])

# the remaining options of read_keypoints: basic_point_only (no hands / face), other canvas sizes (key points are NOT
# rescaled: a smaller canvas clips the skeleton, a larger one leaves it in the corner -- keypoint2img.py:70-90)
opt_maps, opt_cases = [], []
for frame, size, bpo in ((17, (512, 384), True), (0, (512, 384), True), (17, (256, 192), False), (17, (1280, 720), False),
                         (0, (320, 512), False)):
    name = "sa1_%03d_keypoints.json" % frame
    opt_maps.append(keypoint2img.read_keypoints(os.path.join(HERE, "keypoints_fadg0", name), size, 0, False, bpo))
    opt_cases.append((name, size[0], size[1], int(bpo)))
np.savez_compressed(os.path.join(HERE, "pose_maps_options.npz"), names=np.array([c[0] for c in opt_cases]),
                    sizes=np.array([[c[1], c[2]] for c in opt_cases]), basic=np.array([c[3] for c in opt_cases]),
                    **{"map%d" % i: m for i, m in enumerate(opt_maps)})
print("option goldens:", [(c, int((m != 0).any(2).sum())) for c, m in zip(opt_cases, opt_maps)])
