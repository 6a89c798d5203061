#!/usr/bin/env python3
"""f4: camera calibrations per second on the device (one workgroup per camera: DLT + 100 RANSAC subsets + final refinement)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import calib, synth
rng = np.random.default_rng(0)
B = int(os.environ.get('TTUP_CALIB_B', '512'))
kps = []
for i in range(B):
    R, c, f = synth._random_camera(rng)
    Mext = np.eye(4); Mext[:3, :3] = R; Mext[:3, 3] = -R @ c
    Mint = np.array([[f, 0, 960.0, 0], [0, f, 540.0, 0], [0, 0, 1, 0]])
    uv = calib.reproject(synth.TABLE_POINTS, Mint, Mext) + rng.normal(0, 0.7, (13, 2))
    kps.append(np.concatenate([uv, np.ones((13, 1))], axis=1))
kps = np.stack(kps)
calib.calibrate_cameras(kps[:4])
torch.cuda.synchronize()
for b in (1, B):
    t0 = time.perf_counter()
    mint, mext, ninl = calib.calibrate_cameras(kps[:b])
    dt = time.perf_counter() - t0
    print('%d camera(s): %.2f ms -> %.0f calibrations/s (mean inliers %.1f)' % (b, dt * 1e3, b / dt, ninl.mean()))
