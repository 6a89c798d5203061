#!/usr/bin/env python3
"""BASELINE config 1 on the GPU alone: hubconf.full_pipeline().predict on a 48-frame 1280x720 host clip (bench.py's hub_clip_fps)."""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import hubconf  # noqa: E402
from upliftingtabletennis_amd import synth  # noqa: E402

n = int(os.environ.get('TTUP_HUB_FRAMES', '48'))
frames, _ = synth.synth_frames(n, 720, 1280, seed=0)
images = [f for f in frames]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hub = hubconf.full_pipeline()
for _ in range(2):
    hub.predict(images, 60.0)
torch.cuda.synchronize()
reps = int(os.environ.get('TTUP_HUB_REPS', '8'))
t0 = time.perf_counter()
for _ in range(reps):
    spin, pos = hub.predict(images, 60.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print('hub predict, %d frames: %.2f ms per clip = %.1f frames/s' % (n, dt * 1e3, n / dt))
# long clip through the same overlapped clip path (predict() itself raises the reference's ValueError beyond 49 detections): detections
# of all frames + keypoint filter + ball filter, uplift on the first 49 (bench.py's hub_clip_fps_256)
nl = int(os.environ.get('TTUP_HUB_LONG', '256'))
if nl > 0:
    from upliftingtabletennis_amd import glue
    long_images = [f for f in np.concatenate([frames] * ((nl + n - 1) // n))[:nl]]

    def long_clip():
        pos, kp = hub._clip_detections(long_images, want_table=True, table_consumer=lambda k: hub.table_detector_aux.filter_trajectory(k, k))
        filt, _, tb = hub.ball_detector.filter_trajectory(pos, pos, 60.0)
        bc, tc, tm, mk = glue._uplifting_transform(filt[:49], np.asarray(kp, dtype=np.float64), tb[:49])
        return hub.uplifting_model.predict_without_normalization(bc, tc, mk, tm)
    long_clip()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        long_clip()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('hub clip path, %d frames (chunk %d): %.2f ms per clip = %.1f frames/s' % (nl, hub.CHUNK_LONG if nl >= 4 * hub.CHUNK else hub.CHUNK, dt * 1e3, nl / dt))
