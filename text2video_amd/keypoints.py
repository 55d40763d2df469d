"""OpenPose-JSON -> pose-map rasteriser used by the pose dataset (host side, numpy).

Counterpart of the routine the reference's pose dataset calls to build the generator input
(/root/reference/keypoint2img.py:70-210 `read_keypoints`; semantics catalogued in SURVEY.md
Appendix D).  Written from that specification, not from the file: the behaviour that has to be
reproduced bit for bit is

  * BODY_25 pose (10 limbs, head and legs not drawn), 70-point face (8 parts walked in two-point
    steps), 2x21-point hands (5 fingers x 4 segments); validity thresholds 0.01 / 0.1 / 0.01 applied
    per point (pose) or per part (face, finger); a point at x == 0 disables its segment;
  * every segment is a 2-point straight line sampled at int(|delta|) positions along its major
    axis, coordinates truncated toward zero;
  * the line is the Levenberg-Marquardt fit scipy.optimize.curve_fit returns, NOT the closed form:
    LM stops ~1e-5 short of the exact line and after truncation that moves pixels in ~5 % of the
    frames (SURVEY App. D "Exactness note"), so parity mode calls the same SciPy routine
    (`exact_fit=True`); `exact_fit=False` is the fast closed-form path;
  * stamping: offsets range(-bw, bw) (asymmetric), 6x6 for pose/hand, 4x4 white for face;
    "black -> colour, else average" is decided per offset for the WHOLE segment at once;
    round end caps (i*i+j*j < 4*bw*bw) for pose/hand segments only;
  * people are accumulated with uint8 wrap-around addition;
  * two filled radius-8 discs at hand point 9 of each hand ((0,0) when there are no hands), drawn
    with an OpenCV-style midpoint circle -- the one part that cannot be checked here (no OpenCV in
    the container; the golden maps were captured with cv2.circle stubbed out), `hand_discs=True`.
"""
import json
import warnings

import numpy as np

POSE_LIMBS = ((0, 1), (1, 8), (1, 2), (2, 3), (3, 4), (1, 5), (5, 6), (6, 7), (8, 9), (8, 12))
POSE_RGB = ((153, 0, 51), (153, 0, 0), (153, 51, 0), (153, 102, 0), (153, 153, 0), (102, 153, 0), (51, 153, 0),
            (0, 153, 0), (0, 153, 51), (0, 153, 102))
FINGERS = tuple(tuple([0] + list(range(4 * f + 1, 4 * f + 5))) for f in range(5))
FINGER_RGB = ((204, 0, 0), (163, 204, 0), (0, 204, 82), (0, 82, 204), (163, 0, 204))
FACE_PARTS = (
    (tuple(range(0, 17)),),                                   # jaw line
    (tuple(range(17, 22)),), (tuple(range(22, 27)),),         # eyebrows
    (tuple(range(27, 31)), tuple(range(31, 36))),             # nose
    ((36, 37, 38, 39), (39, 40, 41, 36)),                     # left eye
    ((42, 43, 44, 45), (45, 46, 47, 42)),                     # right eye
    (tuple(range(48, 55)), (54, 55, 56, 57, 58, 59, 48)),     # outer mouth
    (tuple(range(60, 65)), (64, 65, 66, 67, 60)),             # inner mouth
)
NOSE_NECK_RGB = POSE_RGB[0]  # the colour --add_face_disc keys its face crop on


def _affine(x, a, b):
    return a * x + b


_EPS = float(np.finfo(np.float64).eps)
_MINPACK = False     # not looked up yet


def _minpack():
    """MINPACK's lmdif, the routine scipy.optimize.curve_fit(method='lm') ends up in.  Imported on first use: the frame
    loop's own process rasterises nothing when the worker pool does (SciPy costs ~0.2 s of its start-up)."""
    global _MINPACK
    if _MINPACK is False:
        _MINPACK = _load_minpack_extension()
        if _MINPACK is None:
            try:
                from scipy.optimize import _minpack as m
                _MINPACK = m
            except ImportError:      # pragma: no cover -- SciPy layouts without the private module: curve_fit itself
                _MINPACK = None
    return _MINPACK


def _load_minpack_extension():
    """scipy/optimize/_minpack.*.so loaded by file, without `import scipy.optimize`: the extension needs libm only, while the
    package import around it costs 0.2-0.35 s -- per rasteriser worker, on the critical path of the one-shot command now
    that the frame loop's process starts without torch (the workers' first maps were what the loop waited for).  The same
    shared object, so the same bits.  None where SciPy is laid out differently (the caller then imports the package)."""
    try:
        import importlib.machinery
        import importlib.util
        import os
        spec = importlib.util.find_spec("scipy")
        if spec is None or not spec.origin:
            return None
        d = os.path.join(os.path.dirname(spec.origin), "optimize")
        for f in sorted(os.listdir(d)):
            if f.startswith("_minpack.") and f.endswith((".so", ".pyd")):
                loader = importlib.machinery.ExtensionFileLoader("_minpack", os.path.join(d, f))
                m = importlib.util.module_from_spec(importlib.util.spec_from_loader("_minpack", loader))
                loader.exec_module(m)
                return m if hasattr(m, "_lmdif") else None
    except Exception:      # noqa: BLE001 -- any surprise: the package import is the fallback
        return None
    return None


def _fit_line(u, v, exact_fit):
    if exact_fit:
        if _minpack() is not None:
            # curve_fit(_affine, u, v) = leastsq(p -> _affine(u, *p) - v, p0 = ones(2)) with leastsq's defaults
            # (ftol = xtol = 1.49012e-8, gtol = 0, maxfev = 200*(n+1), epsfcn = eps, factor = 100, no scaling), which
            # calls _minpack._lmdif; calling it directly returns the SAME bits (tests/test_cpu_oracle_and_host.py
            # compares both on the golden frames and on random segments) without curve_fit's argument checking,
            # memoiser and covariance estimate: 0.04 ms instead of 0.28 ms per fit, 66 fits per frame
            try:
                r = _MINPACK._lmdif(lambda p: (p[0] * u + p[1]) - v, np.ones(2), (), 1, 1.49012e-8, 1.49012e-8, 0.0, 600,
                                    _EPS, 100, None)
                return r[0][0], r[0][1]
            except (AttributeError, TypeError):      # private signature changed: fall back to the public routine
                pass
        from scipy.optimize import curve_fit
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            (a, b), _ = curve_fit(_affine, u, v)
        return a, b
    a = (v[1] - v[0]) / (u[1] - u[0])
    return a, v[0] - a * u[0]


def trace_segment(x, y, exact_fit=True):
    """Integer pixel coordinates (xs, ys) of the 2-point segment; may be empty."""
    x = np.asarray(x, float)
    y = np.asarray(y, float)
    if abs(x[0] - x[1]) < abs(y[0] - y[1]):
        ys, xs = trace_segment(y, x, exact_fit)
        return xs, ys
    n = int(abs(x[1] - x[0]))
    if n <= 0:
        return np.empty(0, int), np.empty(0, int)
    a, b = _fit_line(x, y, exact_fit)
    lo, hi = (x[0], x[1]) if x[0] <= x[1] else (x[1], x[0])
    u = np.linspace(lo, hi, n)
    return u.astype(int), _affine(u, a, b).astype(int)


def _blend(img, yy, xx, rgb):
    cur = img[yy, xx]
    if (cur == 0).all():
        img[yy, xx] = rgb
    else:
        img[yy, xx] = ((cur.astype(float) + rgb) / 2).astype(np.uint8)


_HOST = None


def _host_lib():
    """lib/libt2v_host.so (csrc/raster_host.c, built by `make -C text2video_amd/csrc`): the stamping loops in C.
    False when it is not there -- the numpy form below is the same arithmetic."""
    global _HOST
    if _HOST is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libt2v_host.so")
        _HOST = False
        if os.path.exists(path):
            try:
                lib = ctypes.CDLL(path)
                lib.t2v_raster_stamp.restype = None
                lib.t2v_raster_stamp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
                if lib.t2v_raster_abi() == 1:
                    _HOST = lib
            except (OSError, AttributeError):
                _HOST = False
    return _HOST


def stamp(img, xs, ys, bw, rgb, caps):
    if xs.size == 0:
        return
    h, w = img.shape[:2]
    rgb = np.asarray(rgb, float)
    lib = _host_lib()
    if lib and img.flags.c_contiguous and img.dtype == np.uint8:
        xs64, ys64 = np.ascontiguousarray(xs, np.int64), np.ascontiguousarray(ys, np.int64)
        lib.t2v_raster_stamp(img.ctypes.data, h, w, xs64.ctypes.data, ys64.ctypes.data, int(xs64.size), int(bw),
                             np.ascontiguousarray(rgb, np.float64).ctypes.data, int(bool(caps)))
        return
    for oy in range(-bw, bw):
        yy = np.clip(ys + oy, 0, h - 1)
        for ox in range(-bw, bw):
            _blend(img, yy, np.clip(xs + ox, 0, w - 1), rgb)
    if caps:
        ex, ey = xs[[0, -1]], ys[[0, -1]]
        for oy in range(-2 * bw, 2 * bw):
            for ox in range(-2 * bw, 2 * bw):
                if oy * oy + ox * ox < 4 * bw * bw:
                    _blend(img, np.clip(ey + oy, 0, h - 1), np.clip(ex + ox, 0, w - 1), rgb)


def filled_disc(img, cx, cy, radius, rgb):
    """cv2.circle(img, (cx, cy), radius, rgb, -1) as the reference calls it (keypoint2img.py:159-160).  With
    thickness < 0, the default LINE_8 and shift 0, cv::circle runs drawing.cpp's integer `Circle(img, center, radius,
    color, fill = true)`: the midpoint walk err / dx / dy / plus / minus below, painting per step the horizontal spans
    rows cy -+ dy over [cx - dx, cx + dx] and rows cy -+ dx over [cx - dy, cx + dy], clipped to the image
    [RECALL of OpenCV's modules/imgproc/src/drawing.cpp -- OpenCV is not in this image, so the rule cannot be
    run against the real library; the goldens were captured with the two discs stubbed out]."""
    h, w = img.shape[:2]

    def span(y, x0, x1):
        if 0 <= y < h:
            x0, x1 = max(x0, 0), min(x1, w - 1)
            if x0 <= x1:
                img[y, x0:x1 + 1] = rgb

    err, dx, dy, plus, minus = 0, radius, 0, 1, 2 * radius - 1
    while dx >= dy:
        span(cy - dy, cx - dx, cx + dx)
        span(cy + dy, cx - dx, cx + dx)
        span(cy - dx, cx - dy, cx + dy)
        span(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2


def _valid_xy(pts, kind):
    """(n,3) -> (n,2) with invalid points at (0,0)."""
    out = np.zeros((pts.shape[0], 2))
    if kind == "pose":
        ok = pts[:, 2] > 0.01
        out[ok] = pts[ok, :2]
    elif kind == "face":
        for part in FACE_PARTS:
            for idx in part:
                idx = list(idx)
                if (pts[idx, 2] > 0.1).all():
                    out[idx] = pts[idx, :2]
    else:
        for finger in FINGERS:
            idx = list(finger)
            if (pts[idx, 2] > 0.01).all():
                out[idx] = pts[idx, :2]
    return out


def _draw_chain(img, pts, idx, bw, rgb, caps, exact_fit):
    """consecutive two-point segments along the index list"""
    for i in range(0, max(1, len(idx) - 1)):
        sub = list(idx[i:i + 2])
        x, y = pts[sub, 0], pts[sub, 1]
        if len(sub) == 2 and not (x == 0).any():
            xs, ys = trace_segment(x, y, exact_fit)
            stamp(img, xs, ys, bw, rgb, caps)


def render_person(person, size, basic_point_only=False, exact_fit=True, hand_discs=True, drop_prob=0.0, rng=None,
                  remove_face_labels=False):
    w, h = size
    img = np.zeros((h, w, 3), np.uint8)
    pose = _valid_xy(np.asarray(person["pose_keypoints_2d"], float).reshape(25, 3), "pose")
    face = _valid_xy(np.asarray(person["face_keypoints_2d"], float).reshape(70, 3), "face")
    if len(person.get("hand_left_keypoints_2d", [])) == 0:
        hands = [np.zeros((21, 2)), np.zeros((21, 2))]
    else:
        hands = [_valid_xy(np.asarray(person[k], float).reshape(21, 3), "hand")
                 for k in ("hand_left_keypoints_2d", "hand_right_keypoints_2d")]

    if drop_prob > 0 and remove_face_labels:
        # training-time jitter of the head key points and of the whole face (/root/reference/keypoint2img.py:119-123):
        # same draws in the same order -- randn(5,2), randn(), randn() -- so a RandomState seeded like the reference's
        # global np.random reproduces its maps.  Invalid points sit at (0,0) and are moved like the others, as there.
        pose[[0, 15, 16, 17, 18], :] += 5 * rng.standard_normal((5, 2))
        face[:, 0] += 2 * rng.standard_normal()
        face[:, 1] += 2 * rng.standard_normal()

    def keep():
        return drop_prob <= 0 or rng.random() > drop_prob

    for limb, rgb in zip(POSE_LIMBS, POSE_RGB):
        if keep():
            _draw_chain(img, pose, limb, 3, rgb, True, exact_fit)
    if not basic_point_only:
        for hand in hands:
            if keep():
                for finger, rgb in zip(FINGERS, FINGER_RGB):
                    _draw_chain(img, hand, finger, 3, rgb, True, exact_fit)
        if keep():
            for part in FACE_PARTS:
                for idx in part:
                    _draw_chain(img, face, idx, 2, (255, 255, 255), False, exact_fit)
    if hand_discs:
        filled_disc(img, int(hands[0][9, 0]), int(hands[0][9, 1]), 8, (0, 255, 0))
        filled_disc(img, int(hands[1][9, 0]), int(hands[1][9, 1]), 8, (255, 0, 0))
    return img


def read_keypoints(json_input, size, random_drop_prob=0, remove_face_labels=False, basic_point_only=False,
                   exact_fit=True, hand_discs=True, rng=None):
    """Same call signature as the reference routine (+ keyword extras).  size = (w, h).
    Returns uint8 [h, w, 3].  `json_input` may also be an already parsed OpenPose dict (in-memory L2 driver)."""
    if isinstance(json_input, dict):
        people = json_input["people"]
    else:
        with open(json_input, encoding="utf-8") as fh:
            people = json.load(fh)["people"]
    w, h = size
    canvas = np.zeros((h, w, 3), np.uint8)
    if random_drop_prob > 0 and rng is None:
        rng = np.random.default_rng()
    for person in people:
        canvas += render_person(person, size, basic_point_only, exact_fit, hand_discs, random_drop_prob, rng,
                                remove_face_labels)
    return canvas
