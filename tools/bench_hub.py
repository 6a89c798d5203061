#!/usr/bin/env python3
"""BASELINE config 1 on the GPU: the hub entry point `full_pipeline` on one 48-frame 1280x720 clip (host numpy frames in,
spin + 3D positions out), wall clock including the host->device upload of the frames."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')      # no trained checkpoints offline
import hubconf
from upliftingtabletennis_amd import synth
frames, _ = synth.synth_frames(48, 720, 1280, seed=0)      # 46 detections: the reference caps a rally at 50 tokens and needs one padded slot
images = [f for f in frames]
pipe = hubconf.full_pipeline()
pipe.predict(images[:8], 60.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    spin, pos3d = pipe.predict(images, 60.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print('full_pipeline.predict on %d frames: %.1f ms -> %.0f frames/s (table detection on every frame + ball detection + uplift; pos3d %s)'
      % (len(images), dt * 1e3, len(images) / dt, tuple(pos3d.shape)))
