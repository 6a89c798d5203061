"""Synthetic inputs for benchmarks, tests and golden fixtures (no datasets offline).

Frames:        uint8 (N,H,W,3) BGR, fixed low-frequency background per clip + per-frame noise,
               a white Gaussian blob moving along a projected parabola.
Trajectories:  smooth normalised 2-D tracks + 13 projected table keypoints + timestamps + mask,
               in the input format of the uplift network (reference uplifting/model.py:529-536).
"""
import numpy as np

# 13 table keypoints in world coordinates (reference uplifting/helper.py:32-50)
_TL, _TW, _TH = 2.74, 1.525, 0.76
TABLE_POINTS = np.array([
    [-_TL / 2, _TW / 2, _TH], [-_TL / 2, -_TW / 2, _TH], [0.0, _TW / 2, _TH], [0.0, -_TW / 2, _TH],
    [_TL / 2, _TW / 2, _TH], [_TL / 2, -_TW / 2, _TH], [0.0, _TW / 2 + 0.1525, _TH], [0.0, -(_TW / 2 + 0.1525), _TH],
    [0.0, 0.0, _TH], [0.0, _TW / 2 + 0.1525, _TH + 0.1525], [0.0, -(_TW / 2 + 0.1525), _TH + 0.1525],
    [-_TL / 2, 0, _TH], [_TL / 2, 0, _TH]])


def blob_track(n, h, w, seed=0, margin=24):
    """Pixel-space ball centres (n,2) [x,y] along a parabola; centres sit 0.2 px off a pixel centre
    so that the brightest pixel is unique."""
    rng = np.random.default_rng(seed + 7919)
    t = np.linspace(0.0, 1.0, n)
    x0, x1 = rng.uniform(margin, w * 0.3), rng.uniform(w * 0.7, w - margin)
    y0 = rng.uniform(h * 0.45, h * 0.7)
    amp = rng.uniform(h * 0.15, h * 0.35)
    x = x0 + (x1 - x0) * t
    y = y0 - amp * 4 * t * (1 - t)
    x = np.clip(np.floor(x), margin, w - margin) + 0.2
    y = np.clip(np.floor(y), margin, h - margin) + 0.2
    return np.stack([x, y], 1)


def synth_frames(n, h=720, w=1280, seed=0, sigma=2.0, track=None):
    """(n,h,w,3) uint8 clip and the (n,2) blob centres."""
    rng = np.random.default_rng(seed)
    gy, gx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing='ij')
    bg = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        ph = rng.uniform(0, 2 * np.pi, 4)
        fr = rng.uniform(0.5, 3.0, 4)
        bg[..., c] = 70 + 25 * np.sin(2 * np.pi * fr[0] * gx + ph[0]) * np.cos(2 * np.pi * fr[1] * gy + ph[1]) \
            + 15 * np.sin(2 * np.pi * fr[2] * (gx + gy) + ph[2])
    if track is None:
        track = blob_track(n, h, w, seed)
    frames = np.empty((n, h, w, 3), np.uint8)
    r = int(np.ceil(4 * sigma))
    for i in range(n):
        f = bg + rng.normal(0, 2.0, (h, w, 1)).astype(np.float32)
        cx, cy = track[i]
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        yy, xx = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing='ij')
        g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sigma ** 2)).astype(np.float32)
        f[y0:y1, x0:x1] += (255.0 - f[y0:y1, x0:x1]) * g[..., None]
        frames[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return frames, track


def hard_clip(n, h=720, w=1280, seed=0, sigma=3.5, gain=1.5):
    """A clip whose heatmaps have NEAR-TIES at the top: a wide blob (sigma 3-4 px) under a brightness gain of 1.3-1.6 saturates at
    255 over dozens of pixels, so the detector's heatmap has a flat top (reference top-2 margins of 1e-4 .. 1e-2 on the planted
    weights, against 0.2 on `synth_frames` defaults).  The content of tools/soak_audit.py and of tests/golden/wasb_hard.npz."""
    frames, track = synth_frames(n, h, w, seed=seed, sigma=sigma)
    return np.clip(np.rint(frames.astype(np.float32) * np.float32(gain)), 0, 255).astype(np.uint8), track


def _random_camera(rng):
    """Pinhole camera looking at the table (ranges in the spirit of reference uplifting/data.py:60-64)."""
    dist = rng.uniform(6.0, 12.0)
    az = rng.uniform(-0.5, 0.5)
    el = rng.uniform(0.15, 0.5)
    c = np.array([dist * np.cos(el) * np.sin(az), -dist * np.cos(el) * np.cos(az), dist * np.sin(el) + _TH])
    fwd = np.array([0, 0, _TH]) - c
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])
    f = rng.uniform(1800, 3200)
    return R, c, f


def synth_trajectories(b, t, seed=0, pad=1, fps_choices=(50.0, 60.0, 120.0)):
    """-> ball (b,t+pad,2), table (b,13,3), mask (b,t+pad), times (b,t+pad), float32, normalised coords."""
    rng = np.random.default_rng(seed)
    L = t + pad
    ball = np.zeros((b, L, 2), np.float32)
    table = np.zeros((b, 13, 3), np.float32)
    mask = np.zeros((b, L), np.float32)
    times = np.zeros((b, L), np.float32)
    for i in range(b):
        R, c, f = _random_camera(rng)
        pc = (TABLE_POINTS - c) @ R.T
        uv = pc[:, :2] / pc[:, 2:3] * f + np.array([960.0, 540.0])
        table[i, :, 0] = uv[:, 0] / 1920.0
        table[i, :, 1] = uv[:, 1] / 1080.0
        table[i, :, 2] = (rng.uniform(size=13) < 0.95).astype(np.float32)
        start = rng.uniform(0.2, 0.8, 2)
        steps = rng.normal(0, 0.01, (t, 2)) + rng.normal(0, 0.004, 2)
        ball[i, :t] = np.clip(start + np.cumsum(steps, 0), 0.0, 1.0)
        fps = fps_choices[int(rng.integers(len(fps_choices)))]
        times[i, :t] = (np.arange(t) / fps).astype(np.float32)
        mask[i, :t] = 1.0
    return ball, table, mask, times
