"""Evaluation-path callers of the detector and the uplift net: counterparts of the reference's
``inference/utils.py`` ``process_trajectory_ball`` (:36-67) and ``process_trajectory_uplifting`` (:235-265), the two
functions the paper scripts (``inference/inference_combined.py:65-175``) drive per trajectory.

Same arguments and return values.  What differs underneath: the detector returns the fused device-side
(argmax, 3x3 window) of every heatmap instead of the heatmaps, the **ball**-variant Gaussian refine
(helper_balldetection.py:29-110, the variant this path uses, utils.py:59) runs on the device, and the micro-batch is the
handle's ``max_batch`` instead of 4 frames (frames are independent, the results do not depend on the grouping).
``move_weights`` is accepted and ignored: the handle's weights stay resident in HBM (3 MB + 8 MB)."""
import numpy as np
import torch

from . import _lib, refine, uplift

HEIGHT, WIDTH = 1080, 1920          # inference/utils.py:25 (from helper_balldetection)


def process_trajectory_ball(ball_model, images, move_weights=True):
    """images: (1, T, C, H, W) float32 pre-processed triples of one trajectory (torch, host or device).
    Returns pred_positions (T, 3) float64 [x, y, visibility] in 1920x1080 px."""
    if images.dim() != 5 or images.shape[0] != 1:
        raise ValueError('expected images of shape (1, T, C, H, W)')
    flat = images[0]
    t = flat.shape[0]
    if t == 0:
        return np.zeros((0, 3))
    out = []
    for start in range(0, t, ball_model.max_batch):
        x = flat[start:start + ball_model.max_batch]
        _, idx, win = ball_model.forward(x, want_heatmap=False, want_peaks=True)
        xyv = refine.refine_windows_device(idx, win, ball_model.H, ball_model.W, WIDTH, HEIGHT, _lib.REFINE_BALL)
        out.append(xyv.cpu().numpy())
    return np.concatenate(out, axis=0)


def process_trajectory_uplifting(uplifting_model, predictions_ball, predictions_table, times, mask, transform_mode, move_weights=True):
    """predictions_ball (1,L,2), predictions_table (1,13,3), times (1,L), mask (1,L) as `_uplifting_transform` builds them.
    Returns (pred_spin (3,) numpy, pred_positions_3d (T',3) numpy)."""
    pred_spin, pos3d = uplifting_model(predictions_ball, predictions_table, mask, times)
    if transform_mode == 'global':
        pred_spin = uplift.transform_rotationaxes(pred_spin, pos3d)
    t_prime = int(torch.as_tensor(mask).sum().item())
    return pred_spin[0].cpu().numpy(), pos3d[0, :t_prime, :].cpu().numpy()
