#!/usr/bin/env python3
"""Wall-clock split of one bench step (256 triples): CNN+argmax, window fit, host glue, uplift."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import pipeline, synth, weights, _lib, refine
dev = torch.device('cuda:0')
w = pipeline.StreamWorker(dev, weights.random_wasb_state_dict(0, planted=True), weights.random_uplift_state_dict(0, 'large'), max_triples=256)
base, _ = synth.synth_frames(34, 720, 1280, seed=0)
frames = torch.from_numpy(np.concatenate([base] * 8)[:258]).to(dev)
table = np.array(synth.synth_trajectories(1, 4, seed=0)[1][0], dtype=np.float64)
table[:, 0] *= 1920; table[:, 1] *= 1080
def sync(): torch.cuda.synchronize()
for it in range(3):
    sync(); t0 = time.perf_counter()
    _, idx, win = w.net.forward_frames(frames, want_heatmap=False); sync(); t1 = time.perf_counter()
    xyv = refine.refine_windows_device(idx, win, w.net_h, w.net_w, 1920, 1080, _lib.REFINE_TABLE); sync(); t2 = time.perf_counter()
    pos = xyv.cpu().numpy(); t3 = time.perf_counter()
    balls, tables, times, masks = [], [], [], []
    for s in range(0, pos.shape[0], w.traj_len):
        seg = pos[s:s + w.traj_len]
        filt, _, t = w._glue.filter_trajectory_ball(seg, seg, 60.0)
        b, tb, tm, mk = w._glue._uplifting_transform(filt, table, t, w.seq_len)
        balls.append(b); tables.append(tb); times.append(tm); masks.append(mk)
    t4 = time.perf_counter()
    mask = torch.cat(masks)
    rot, p3 = w.up(torch.cat(balls), torch.cat(tables), mask, torch.cat(times)); sync(); t5 = time.perf_counter()
    spin = w._uplift.transform_rotationaxes(rot, p3); sync(); t6 = time.perf_counter()
    print('cnn %.2f ms  fit %.2f  d2h %.2f  glue %.2f  uplift %.2f  axes %.2f  total %.2f' % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)))
# PCIe-inclusive note for DESIGN.md: host -> device copy of the 258-frame uint8 clip (pinned and pageable)
host = frames.cpu()
pinned = host.pin_memory()
for name, src in (('pinned', pinned), ('pageable', host)):
    sync(); t0 = time.perf_counter()
    for _ in range(3):
        d = src.to(dev, non_blocking=True); sync()
    dt = (time.perf_counter() - t0) / 3
    print('H2D %s: %.1f MB in %.2f ms -> %.1f GB/s' % (name, src.numel() / 1e6, dt * 1e3, src.numel() / dt / 1e9))
