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


def _dbscan_labels(points, eps, min_samples):
    """DBSCAN labels of a few dozen 2-D points, equal to what scikit-learn's DBSCAN (which the reference calls,
    inference/utils.py:213) assigns: neighbourhoods are the points within `eps` (Euclidean, the point itself included), core
    points have at least `min_samples` neighbours, clusters are the connected components of the core points, numbered by the
    lowest point index they contain (sklearn grows them one at a time in index order), a border point belongs to the
    lowest-numbered cluster with a core point in its neighbourhood (the first one whose expansion reaches it), noise is -1.
    Plain numpy on an (N,N) distance table -- the 13 keypoint tracks of a clip take about a millisecond instead of the 5 ms of 13
    estimator constructions and tree builds; checked against sklearn on thousands of random sets (tests/test_cabi.py)."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    d2 = (pts[:, None, 0] - pts[None, :, 0]) ** 2 + (pts[:, None, 1] - pts[None, :, 1]) ** 2
    near = d2 <= float(eps) * float(eps)
    core = near.sum(1) >= min_samples
    big = n
    comp = np.where(core, np.arange(n), big)            # component id = lowest core index reachable through core points
    link = near & core[None, :]                          # j is a core neighbour of i
    while True:
        reach = np.where(link, comp[None, :], big).min(1)
        new = np.where(core, np.minimum(comp, reach), big)
        if np.array_equal(new, comp):
            break
        comp = new
    lab = np.where(core, comp, np.where(link, comp[None, :], big).min(1))     # border points: lowest adjacent component
    ids = np.unique(lab[lab < big])                      # sorted lowest indices -> cluster numbers 0, 1, ...
    out = np.full(n, -1, dtype=np.int64)
    if ids.size:
        out[lab < big] = np.searchsorted(ids, lab[lab < big])
    return out


def _filter_keypoints_with_dbscan(detections, eps=10, min_samples=5):
    """Centroid of the largest DBSCAN cluster of one keypoint's detections over time (inference/utils.py:184-232)."""
    detections = np.asarray(detections)
    if detections.shape[0] < min_samples:
        return np.mean(detections, axis=0) if detections.shape[0] > 0 else None
    labels = _dbscan_labels(detections, eps, min_samples)
    valid = labels[labels != -1]
    if valid.size == 0:
        return np.mean(detections, axis=0)
    # Counter(valid).most_common(1): the most frequent label, ties -> the one met first in point order
    counts = np.bincount(valid)
    best = counts.max()
    largest = next(int(l) for l in valid if counts[l] == best)
    return np.mean(detections[labels == largest], axis=0)


def filter_trajectory_table(pred_positions1, pred_positions2):
    """(T,13,3) x 2 -> (13,3): keep frames where both detectors see a keypoint within 10 px, cluster, take the centroid
    (inference/utils.py:137-180).  The per-frame agreement test is evaluated for all frames and keypoints at once; the handful
    of distances within 1e-9 of the threshold are decided by the reference's own expression (np.linalg.norm of the pair)."""
    threshold = 10
    p1, p2 = np.asarray(pred_positions1), np.asarray(pred_positions2)
    dx, dy = p1[:, :, 0] - p2[:, :, 0], p1[:, :, 1] - p2[:, :, 1]
    dist = np.sqrt(dx * dx + dy * dy)
    both = (p1[:, :, 2] == KEYPOINT_VISIBLE) & (p2[:, :, 2] == KEYPOINT_VISIBLE)
    keep = both & (dist < threshold)
    for t, n in zip(*np.nonzero(both & (np.abs(dist - threshold) <= 1e-9))):
        keep[t, n] = np.linalg.norm([dx[t, n], dy[t, n]]) < threshold
    out = []
    for n in range(p1.shape[1]):
        sel = keep[:, n]
        if int(sel.sum()) < 3:
            out.append([-1, -1, KEYPOINT_INVISIBLE])
        else:
            pt = _filter_keypoints_with_dbscan(p1[sel, n, :2], eps=10, min_samples=3)
            out.append([pt[0], pt[1], KEYPOINT_VISIBLE] if pt is not None else [-1, -1, KEYPOINT_INVISIBLE])
    return np.array(out)
