#!/usr/bin/env python3
"""Probe of the hub surface on clip lengths around its chunk boundaries: `full_pipeline().predict` through the overlapped clip path
against the serial path (TTUP_HUB_SERIAL=1, read per call) on the same frames -- spin and 3-D positions must be identical (same
kernels per frame, same host glue), whatever the split into chunks, first / last partial chunks and micro-batches."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import hubconf
from upliftingtabletennis_amd import synth
lengths = [int(v) for v in os.environ.get('TTUP_PROBE_LENGTHS', '3,4,9,23,24,25,26,47,49,63,71,72,73,96,97,127,129,130').split(',')]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hub = hubconf.full_pipeline()
bad = 0
for n in lengths:
    images = [f for f in synth.synth_frames(n, 720, 1280, seed=100 + n)[0]]
    res = {}
    for mode in ('0', '1'):
        os.environ['TTUP_HUB_SERIAL'] = mode
        try:
            spin, pos = hub.predict(images, 60.0)
            res[mode] = (spin.detach().cpu().numpy().copy(), np.asarray(pos).copy())
        except Exception as e:          # the reference raises on degenerate clips (e.g. a mask without a zero): both modes must agree on that too
            res[mode] = ('error', type(e).__name__, str(e)[:80])
    a, b = res['0'], res['1']
    # the detections of the overlapped clip path against the detectors' own clip calls (also where `predict` raises for both)
    pos_o, kp_o = hub._clip_detections(images, want_table=True)
    pos_s, kp_s = hub.ball_detector.predict_clip(images), hub.table_detector.predict_keypoints(images)
    det_same = np.array_equal(np.asarray(pos_o), np.asarray(pos_s)) and np.array_equal(np.asarray(kp_o), np.asarray(kp_s))
    if not det_same:
        print('%4d frames: DETECTIONS differ: ball %.3g px, table %.3g px' % (n, float(np.abs(np.asarray(pos_o) - np.asarray(pos_s)).max()), float(np.abs(np.asarray(kp_o) - np.asarray(kp_s)).max())))
        bad += 1
    if isinstance(a[0], str) or isinstance(b[0], str):
        same = isinstance(a[0], str) and isinstance(b[0], str) and a[1] == b[1]
        print('%4d frames: predict raises %s in both modes: %s; detections identical %s' % (n, a[1] if isinstance(a[0], str) else 'nothing', same, det_same), flush=True)
        bad += not same
        continue
    same = a[1].shape == b[1].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    d = float(np.abs(a[1] - b[1]).max()) if a[1].shape == b[1].shape and a[1].size else float('nan')
    print('%4d frames: %d positions, predict identical %s (max |dpos| %.3g), detections identical %s' % (n, a[1].shape[0], same, d, det_same), flush=True)
    bad += not same
print('mismatching lengths: %d' % bad)
