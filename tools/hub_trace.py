#!/usr/bin/env python3
"""Host-side time line of the hub pipeline's overlapped clip path on a 48-frame clip: when each stage of `_clip_detections` is
reached (ms since the call), averaged over a few calls, plus the uplift tail."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import hubconf
from upliftingtabletennis_amd import synth
n = int(os.environ.get('TTUP_HUB_FRAMES', '48'))
images = [f for f in synth.synth_frames(n, 720, 1280, seed=0)[0]]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hub = hubconf.full_pipeline()
for _ in range(3):
    hub.predict(images, 60.0)
acc, tot = {}, []
for _ in range(6):
    hub._trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hub.predict(images, 60.0)
    torch.cuda.synchronize()
    tot.append((time.perf_counter() - t0) * 1e3)
    for k, v in hub._trace:
        acc.setdefault(k, []).append(v)
for k, v in acc.items():
    print('%-45s %7.2f ms' % (k, np.mean(v)))
print('%-45s %7.2f ms  (%.0f frames/s)' % ('predict returns', np.mean(tot), n / np.mean(tot) * 1e3))
