"""a5: host-side glue between the detectors and the uplift network (tiny, per-trajectory, stays on the host as in
the reference): ``filter_trajectory_ball`` (inference/utils.py:70-102), ``_uplifting_transform`` (:268-309) and, for
the table detector (f1), ``filter_trajectory_table`` with its DBSCAN clustering (:137-232)."""
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


KEYPOINT_VISIBLE, KEYPOINT_INVISIBLE = 1, 0


def _filter_keypoints_with_dbscan(detections, eps=10, min_samples=5):
    """Centroid of the largest DBSCAN cluster of one keypoint's detections over time (inference/utils.py:184-232)."""
    from collections import Counter
    from sklearn.cluster import DBSCAN
    detections = np.asarray(detections)
    if detections.shape[0] < min_samples:
        return np.mean(detections, axis=0) if detections.shape[0] > 0 else None
    labels = DBSCAN(eps=eps, min_samples=min_samples).fit(detections).labels_
    valid = [l for l in labels if l != -1]
    if not valid:
        return np.mean(detections, axis=0)
    largest = Counter(valid).most_common(1)[0][0]
    return np.mean(detections[labels == largest], axis=0)


def filter_trajectory_table(pred_positions1, pred_positions2):
    """(T,13,3) x 2 -> (13,3): keep frames where both detectors see a keypoint within 10 px, cluster, take the centroid
    (inference/utils.py:137-180)."""
    threshold = 10
    p1, p2 = np.asarray(pred_positions1), np.asarray(pred_positions2)
    out = []
    for n in range(p1.shape[1]):
        vx, vy = [], []
        for t in range(p1.shape[0]):
            if p1[t, n, 2] == KEYPOINT_VISIBLE and p2[t, n, 2] == KEYPOINT_VISIBLE:
                if np.linalg.norm([p1[t, n, 0] - p2[t, n, 0], p1[t, n, 1] - p2[t, n, 1]]) < threshold:
                    vx.append(p1[t, n, 0]); vy.append(p1[t, n, 1])
        if len(vx) < 3:
            out.append([-1, -1, KEYPOINT_INVISIBLE])
        else:
            pt = _filter_keypoints_with_dbscan(np.stack([vx, vy], axis=1), eps=10, min_samples=3)
            out.append([pt[0], pt[1], KEYPOINT_VISIBLE] if pt is not None else [-1, -1, KEYPOINT_INVISIBLE])
    return np.array(out)
