#!/usr/bin/env python3
"""f1: throughput of the 13-keypoint table detector (3-in / 13-out HRNet + 13-channel refine) on resident 1280x720 frames."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import refine, synth, wasb, weights, _lib
n = int(os.environ.get('TTUP_TABLE_FRAMES', '64'))
frames, _ = synth.synth_frames(10, 720, 1280, seed=0)
fr = torch.from_numpy(np.concatenate([frames] * ((n + 9) // 10))[:n]).cuda()
net = wasb.get_table_model('hrnet', resolution=(1280, 704), state_dict=weights.random_wasb_state_dict(1, in_ch=3, head_out=13), max_batch=n, dtype='bf16')
def step():
    heat, idx, win = net.forward_frames(fr, want_heatmap=False)
    return refine.refine_windows_device(idx.reshape(-1), win.reshape(-1, 9), 704, 1280, 1920, 1080, _lib.REFINE_TABLE)
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print('table detector: %d frames in %.1f ms -> %.0f frames/s (13 keypoints refined per frame; out %s)' % (n, dt * 1e3, n / dt, tuple(out.shape)))
