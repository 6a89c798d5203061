"""Oracle (a1, a5): pre-processing prologue and the glue between detector and uplift net.

Reference map:
  Resize (cv2.resize, default INTER_LINEAR on uint8)  balldetection/transforms.py:17-52
  NormalizeImage (/255, ImageNet mean/std, float64)   balldetection/transforms.py:379-402
  triple concat, HWC->CHW, astype(float32)            interface.py:104-112
  filter_trajectory_ball                              inference/utils.py:70-102
  _uplifting_transform                                inference/utils.py:268-309

PARITY UNPINNED for ``resize_linear_u8``: OpenCV (opencv-python==4.10.0.84, requirements.txt:6)
is not importable in the build container, so the function restates OpenCV's published
fixed-point INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp: 11-bit coefficients,
``saturate_cast<short>(c*2048)``, vertical pass ``(((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2``)
and is only checked for self-consistency (identity at equal size, monotone, exact on constants).
Everything else here is pinned by goldens generated from the reference source.
"""
import numpy as np

WIDTH, HEIGHT = 1920, 1080          # helper_balldetection.py:12
MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_taps(src_n, dst_n):
    """OpenCV linear tap table for one axis: (i0, i1, c0, c1) with int16 coefficients."""
    scale = src_n / dst_n
    d = np.arange(dst_n)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= src_n - 1
    f[hi], s[hi] = 0, src_n - 1
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)
    c0 = np.rint((np.float32(1) - f) * np.float32(COEF_SCALE)).astype(np.int64)
    return s, np.minimum(s + 1, src_n - 1), c0, c1


def _axis_taps_v(src_n, dst_n):
    """Vertical taps: OpenCV clamps the row index only (no coefficient reset)."""
    scale = src_n / dst_n
    d = np.arange(dst_n)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)
    c0 = np.rint((np.float32(1) - f) * np.float32(COEF_SCALE)).astype(np.int64)
    return np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1), c0, c1


def resize_linear_u8(img, dst_w, dst_h):
    """(H,W,C) uint8 -> (dst_h,dst_w,C) uint8, OpenCV INTER_LINEAR fixed-point semantics.  UNPINNED."""
    img = np.asarray(img)
    h, w, _ = img.shape
    if (w, h) == (dst_w, dst_h):
        return img.copy()
    x0, x1, a0, a1 = _axis_taps(w, dst_w)
    y0, y1, b0, b1 = _axis_taps_v(h, dst_h)
    s = img.astype(np.int64)
    hor = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]          # (H,dst_w,C), scaled by 2048
    top, bot = hor[y0], hor[y1]
    out = (((b0[:, None, None] * (top >> 4)) >> 16) + ((b1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def normalize_image(img_u8):
    """transforms.py:388-401: float64 (x/255 - mean)/std applied to channels in the given order."""
    return (img_u8 / 255.0 - MEAN) / STD


def triple_to_tensor(prev, cur, nxt, res_wh):
    """interface.py:104-112 -> (9,H,W) float32 (channel order as given, i.e. BGR for the hub surface)."""
    w, h = res_wh
    ims = [normalize_image(resize_linear_u8(i, w, h)) for i in (prev, cur, nxt)]
    return np.concatenate(ims, axis=2).transpose(2, 0, 1).astype(np.float32)


def filter_trajectory_ball(p1, p2, fps):
    """inference/utils.py:70-102."""
    p1, p2 = np.asarray(p1), np.asarray(p2)
    fps = float(fps)
    diff = np.linalg.norm(p1[:, :2] - p2[:, :2], axis=1)
    keep = [t for t in range(p1.shape[0]) if not (diff[t] > 20 or p1[t, 2] != 1 or p2[t, 2] != 1)]
    pos = np.array([p1[t] for t in keep])[:, :2]           # raises IndexError on empty, like the reference (:98)
    return pos, np.array(keep), np.array([float(t / fps) for t in keep])


def uplifting_transform(ball, table, times, seq_len=50):
    """inference/utils.py:268-309 -> ball (1,L,2), table (1,13,3), times (1,L), mask (1,L), all float32 numpy."""
    ball = (np.asarray(ball, dtype=np.float64) / np.array([WIDTH, HEIGHT])).astype(np.float32)[None]
    table = np.array(table, dtype=np.float64)
    table[:, 0] /= WIDTH
    table[:, 1] /= HEIGHT
    table = table.astype(np.float32)[None]
    tp = ball.shape[1]
    if tp < seq_len:
        b = np.zeros((1, seq_len, 2), np.float32)
        b[:, :tp] = ball
        t = np.zeros((1, seq_len), np.float32)
        t[:, :tp] = np.asarray(times, dtype=np.float32)[None]
        m = np.zeros((1, seq_len), np.float32)
        m[:, :tp] = 1.0
        return b, table, t, m
    return ball[:, :seq_len], table, np.asarray(times[:seq_len], dtype=np.float32)[None], np.ones((1, seq_len), np.float32)


def _largest_cluster_centroid(detections, eps=10, min_samples=5):
    """inference/utils.py:184-232: centroid of the largest DBSCAN cluster (scikit-learn, as in the reference)."""
    from collections import Counter
    from sklearn.cluster import DBSCAN
    detections = np.asarray(detections)
    if detections.shape[0] < min_samples:
        return np.mean(detections, axis=0) if detections.shape[0] > 0 else None
    labels = DBSCAN(eps=eps, min_samples=min_samples).fit(detections).labels_
    valid = [lb for lb in labels if lb != -1]
    if len(valid) == 0:
        return np.mean(detections, axis=0)
    largest = Counter(valid).most_common(1)[0][0]
    return np.mean(detections[labels == largest], axis=0)


def filter_trajectory_table(p1, p2):
    """inference/utils.py:137-180: per keypoint, frames where both detectors see it within 10 px -> DBSCAN centroid;
    fewer than 3 such frames -> (-1, -1, invisible)."""
    p1, p2 = np.asarray(p1), np.asarray(p2)
    out = []
    for n in range(p1.shape[1]):
        xs, ys = [], []
        for t in range(p1.shape[0]):
            if p1[t, n, 2] == 1 and p2[t, n, 2] == 1 and np.linalg.norm([p1[t, n, 0] - p2[t, n, 0], p1[t, n, 1] - p2[t, n, 1]]) < 10:
                xs.append(p1[t, n, 0]); ys.append(p1[t, n, 1])
        if len(xs) < 3:
            out.append([-1, -1, 0])
            continue
        c = _largest_cluster_centroid(np.stack([xs, ys], axis=1), eps=10, min_samples=3)
        out.append([c[0], c[1], 1] if c is not None else [-1, -1, 0])
    return np.array(out)
