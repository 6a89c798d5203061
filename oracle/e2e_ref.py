"""Oracle (test infrastructure): the whole hot path chained the way the reference's hub surface chains it, built from the
other oracle pieces.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference map (interface.py):
  BallDetector.predict            :93-120   per triple: transform, concat prev|cur|next, CHW float32, model, table-variant extract
  TableDetector.predict           :148-172  per frame: transform, CHW float32, model, table-variant extract
  TableTennisPipeline.predict     :265-289  triples from consecutive frames, both filters, _uplifting_transform, uplift
  UpliftingModel.predict_without_normalization :221-247  model, transform_rotationaxes ('global'), crop to T'
Without the un-vendored SegFormer++ detectors each in-tree detector stands in for both sides of its agreement filter
(DESIGN.md 1).  Pinned by tests/golden/e2e.npz, which tools/make_goldens.py produced by running the reference's own modules in
this order (tests/test_oracle_golden.py::test_e2e_oracle_matches_reference_chain).
"""
import numpy as np
import torch

from . import glue_ref, refine_ref, uplift_ref, wasb_ref


def ball_positions(frames, sd_ball, res_wh, batch=1):
    """(N,h,w,3) uint8 BGR frames -> ((N-2,3) float64 [x,y,vis] in 1920x1080 px, (N-2,) argmax indices)."""
    pos, idx = [], []
    n = len(frames)
    for b0 in range(1, n - 1, batch):
        xs = np.stack([glue_ref.triple_to_tensor(frames[i - 1], frames[i], frames[i + 1], res_wh) for i in range(b0, min(b0 + batch, n - 1))])
        heat = wasb_ref.wasb_forward(xs, sd_ball).numpy()
        pos.append(refine_ref.extract_position_table(heat, glue_ref.WIDTH, glue_ref.HEIGHT).reshape(-1, 3))
        idx.append(heat.reshape(heat.shape[0], -1).argmax(1))
    return np.concatenate(pos), np.concatenate(idx)


def table_keypoints(frames, sd_table, res_wh):
    """(N,h,w,3) uint8 -> ((N,13,3) float64, (N,13) argmax indices)."""
    w, h = res_wh
    kps, idx = [], []
    for f in frames:
        x = glue_ref.normalize_image(glue_ref.resize_linear_u8(f, w, h)).transpose(2, 0, 1).astype(np.float32)[None]
        with torch.no_grad():
            heat = wasb_ref.hrnet_forward(torch.from_numpy(x), sd_table)[0].numpy()
        kps.append(refine_ref.extract_position_table(heat, glue_ref.WIDTH, glue_ref.HEIGHT))
        idx.append(heat.reshape(13, -1).argmax(1))
    return np.concatenate(kps), np.stack(idx)


def uplift_from_detections(positions, table_kp, fps, sd_up, seq_len=50):
    """Detections (T,3) + filtered table keypoints (13,3) -> (spin_local (3,), pos3d (T',3), intermediates)."""
    filtered, valid, times_ball = glue_ref.filter_trajectory_ball(positions, positions, fps)
    ball, table, times, mask = glue_ref.uplifting_transform(filtered, table_kp, times_ball, seq_len)
    rot, pos = uplift_ref.uplift_forward(ball, table, mask, times, sd_up)
    spin = uplift_ref.transform_rotationaxes(rot, pos)
    tp = int(mask.sum())
    return spin[0].numpy(), pos[0, :tp].numpy(), dict(filtered=filtered, valid_idx=valid, times_ball=times_ball, u_ball=ball, u_table=table,
                                                      u_times=times, u_mask=mask, rot=rot.numpy())


def full_pipeline(frames, fps, sd_ball, sd_table, sd_up, res_wh, table_kp=None):
    """TableTennisPipeline.predict on the oracle: -> (spin (3,), pos3d (T',3), dict of intermediates)."""
    pos, idx = ball_positions(frames, sd_ball, res_wh)
    inter = dict(ball_positions=pos, ball_argmax=idx)
    if table_kp is None:
        kp, tidx = table_keypoints(frames, sd_table, res_wh)
        table_kp = glue_ref.filter_trajectory_table(kp, kp)
        inter.update(table_keypoints=kp, table_argmax=tidx)
    inter['filtered_table'] = np.asarray(table_kp)
    spin, pos3d, more = uplift_from_detections(pos, table_kp, fps, sd_up)
    inter.update(more)
    return spin, pos3d, inter
