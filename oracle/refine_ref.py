"""Oracle (a3/a4): heatmap argmax + 3x3 window + bounded 2-D Gaussian fit.

Restates ``balldetection/helper_balldetection.py:29-110`` (ball variant) and
``tabledetection/helper_tabledetection.py:50-156`` (table variant) in numpy + the same
third-party optimiser the reference calls (``scipy.optimize.minimize(method='L-BFGS-B')``,
scipy pinned 1.15.2 in requirements.txt:11, 1.15.3 here).

Differences between the variants (all restated below):
  ball  : sigma bounds [0.5, 50]; visibility = 1 iff peak > -inf (always)        (:13, :82, :85)
  table : sigma bounds [0.5, 3], loss clamps sigma >= 0.5; visibility compared against
          threshold 0.1 but then overwritten with 1                                (:80-81, :119, :142)
"""
import numpy as np
from scipy.optimize import minimize

BALL, TABLE = 0, 1
_XY = np.stack(np.meshgrid(np.arange(3), np.arange(3), indexing='ij')[::-1]).reshape(2, 9).astype(np.float64)
# _XY[0] = x (fast axis), _XY[1] = y  -- helper_balldetection.py:67-68


def argmax_window(heat):
    """heat (N,H,W) fp32 -> idx (N,) int64 (first max, like torch.argmax), win (N,3,3) fp32 zero padded.
    helper_balldetection.py:50-64."""
    heat = np.asarray(heat, dtype=np.float32)
    n, h, w = heat.shape
    idx = heat.reshape(n, -1).argmax(axis=1).astype(np.int64)
    pad = np.zeros((n, h + 2, w + 2), np.float32)
    pad[:, 1:-1, 1:-1] = heat
    y, x = idx // w, idx % w
    win = np.stack([pad[i, y[i]:y[i] + 3, x[i]:x[i] + 3] for i in range(n)]) if n else np.zeros((0, 3, 3), np.float32)
    return idx, win


def gaussian_loss(params, window_flat, clamp_sigma):
    """helper_balldetection.py:70-74 / helper_tabledetection.py:77-83."""
    x0, y0, sx, sy = params
    if clamp_sigma:
        sx, sy = max(0.5, sx), max(0.5, sy)
    g = np.exp(-((_XY[0] - x0) ** 2 / (2 * sx ** 2) + (_XY[1] - y0) ** 2 / (2 * sy ** 2)))
    return np.mean((g - window_flat) ** 2)


def fit_window(win, variant):
    """One 3x3 window -> (x_offset, y_offset, success).  helper_balldetection.py:84-94,
    helper_tabledetection.py:118-134."""
    win = np.asarray(win, dtype=np.float32)
    flat = win.flatten()
    init = np.array([1, 1, 1.0, 1.0], dtype=np.float32)
    smax = 50 if variant == BALL else 3
    bounds = [(0, 3), (0, 3), (0.5, smax), (0.5, smax)]
    res = minimize(lambda p: gaussian_loss(p, flat, variant == TABLE), init, method='L-BFGS-B', bounds=bounds)
    if res.success:
        return float(res.x[0]), float(res.x[1]), True
    # fallback = mean position of the window maximum (table variant :130-134; the ball variant's
    # fallback :92-94 calls .float() on an ndarray and would raise -- treated as the same intent)
    yy, xx = np.where(win == win.max())
    return float(np.mean(xx)), float(np.mean(yy)), False


def extract_position(heat, image_width, image_height, variant):
    """heat (B,C,H,W) or (B,H,W) -> (B,C,3) float64 [x, y, vis] in image_width x image_height pixels."""
    heat = np.asarray(heat, dtype=np.float32)
    if heat.ndim == 3:
        heat = heat[:, None]
    if heat.ndim != 4:
        raise ValueError('Heatmaps must have shape (B, C, H, W)')
    b, c, h, w = heat.shape
    idx, win = argmax_window(heat.reshape(b * c, h, w))
    out = np.zeros((b * c, 3))
    for i in range(b * c):
        xo, yo, _ = fit_window(win[i], variant)
        # index is converted through float32 in the reference (x_max[b].float(), :96)
        out[i, 0] = np.float64(np.float32(idx[i] % w)) - 1 + xo
        out[i, 1] = np.float64(np.float32(idx[i] // w)) - 1 + yo
        out[i, 2] = 1.0
    out[:, 0] = (out[:, 0] + 0.5) * (image_width / w) - 0.5     # :101-108 / :145-154
    out[:, 1] = (out[:, 1] + 0.5) * (image_height / h) - 0.5
    return out.reshape(b, c, 3)


def extract_position_ball(heat, image_width, image_height):
    """Ball variant returns (B,3) (helper_balldetection.py:110)."""
    heat = np.asarray(heat)
    if heat.ndim == 4:
        heat = heat[:, 0]
    if heat.ndim != 3:
        raise ValueError('Heatmaps must have shape (B, H, W)')
    return extract_position(heat, image_width, image_height, BALL)[:, 0]


def extract_position_table(heat, image_width, image_height):
    heat = np.asarray(heat)
    if heat.ndim != 4:
        raise ValueError('Heatmaps must have shape (B, C, H, W)')
    return extract_position(heat, image_width, image_height, TABLE)
