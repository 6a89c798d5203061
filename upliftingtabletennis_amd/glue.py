"""a5: host-side glue between the detector and the uplift network (tiny, per-trajectory, stays on the host as in
the reference): ``filter_trajectory_ball`` (inference/utils.py:70-102) and ``_uplifting_transform`` (:268-309)."""
import numpy as np
import torch

HEIGHT, WIDTH = 1080, 1920          # balldetection/helper_balldetection.py:12
BALL_VISIBLE = 1
SEQ_LEN = 50                        # inference/utils.py:294


def filter_trajectory_ball(pred_positions1, pred_positions2, fps):
    """Keep frames where both detectors see the ball and agree within 20 px.
    Returns (positions (T',2), valid indices (T',), times (T',) = index / fps)."""
    threshold = 20
    pred_positions1, pred_positions2 = np.asarray(pred_positions1), np.asarray(pred_positions2)
    fps = float(fps)
    diff = np.linalg.norm(pred_positions1[:, :2] - pred_positions2[:, :2], axis=1)
    keep = [t for t in range(pred_positions1.shape[0])
            if not (diff[t] > threshold or pred_positions1[t, 2] != BALL_VISIBLE or pred_positions2[t, 2] != BALL_VISIBLE)]
    valid = np.array([pred_positions1[t] for t in keep])[:, :2]
    return valid, np.array(keep), np.array([float(t / fps) for t in keep])


def _uplifting_transform(ball_coords, table_coords, times, seq_len=SEQ_LEN):
    """Normalise by (1920,1080), pad / truncate to seq_len, build the mask.
    Returns torch float32 tensors ball (1,L,2), table (1,13,3), times (1,L), mask (1,L)."""
    ball = torch.tensor(np.asarray(ball_coords, dtype=np.float64) / np.array([WIDTH, HEIGHT]), dtype=torch.float32).unsqueeze(0)
    table = np.array(table_coords, dtype=np.float64)
    table[:, 0] = table[:, 0] / WIDTH
    table[:, 1] = table[:, 1] / HEIGHT
    table = torch.tensor(table, dtype=torch.float32).unsqueeze(0)
    t_prime = ball.shape[1]
    if t_prime < seq_len:
        tmp = torch.zeros((1, seq_len, 2), dtype=torch.float32)
        tmp[:, :t_prime, :] = ball
        ball = tmp
        tmp = torch.zeros((1, seq_len), dtype=torch.float32)
        tmp[:, :t_prime] = torch.tensor(np.asarray(times), dtype=torch.float32).unsqueeze(0)
        times = tmp
        mask = torch.zeros((1, seq_len), dtype=torch.float32)
        mask[:, :t_prime] = 1.0
    else:
        ball = ball[:, :seq_len, :]
        times = torch.tensor(np.asarray(times)[:seq_len], dtype=torch.float32).unsqueeze(0)
        mask = torch.ones((1, seq_len), dtype=torch.float32)
    return ball, table, times, mask
