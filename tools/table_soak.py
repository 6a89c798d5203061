#!/usr/bin/env python3
"""Soak of the certified argmax of the table detector's 13 heads: TTUP_SOAK_CLIPS clips of 16 frames of changing content (blob size,
brightness, background per clip) through `TableDetector('hrnet')._certified_peaks` with the audits on; all 13 argmax indices of every
frame are compared with the full-frame fp32 path.  TTUP_TABLE_NOISE_WEIGHTS=1 runs the pure-noise stand-in weights (near-ties
everywhere).  Prints one JSON line."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
from upliftingtabletennis_amd import synth, wasb
from upliftingtabletennis_amd.interface import TableDetector
N = int(os.environ.get('TTUP_SOAK_CLIPS', '16'))
det = TableDetector('hrnet', max_batch=16)
det.AUDIT_EVERY = 16
m = det.model
f32 = m._make(dtype='f32')
w, h = det.model_resolution
rng = np.random.default_rng(77)
mism = total = 0
t0 = time.time()
eps0 = None
for c in range(N):
    frames = synth.hard_clip(16, 720, 1280, seed=900 + c, sigma=float(rng.uniform(1.2, 4.0)), gain=float(rng.uniform(0.6, 1.6)))[0]
    kp = det.predict_keypoints(list(frames))          # calibrates / audits / widens like the hub surface does
    if eps0 is None:
        eps0 = float(m.eps)
    fr = torch.from_numpy(frames).cuda()
    idx, win = det._certified_peaks(fr)
    x = wasb.preprocess_frames(fr, (w, h))
    ref = torch.cat([wasb.WASBNet.forward(f32, x[t:t + 1], want_heatmap=False, want_peaks=True)[1] for t in range(16)])
    mism += int((idx != ref).sum()); total += int(idx.numel())
a = m.audit_state
cs = m.certify_stats()
print(json.dumps({'tool': 'tools/table_soak.py', 'clips': N, 'indices_checked_against_fp32': total, 'argmax_mismatches': mism,
                  'eps_first': round(eps0, 5), 'eps_last': round(float(m.eps), 5), 'eps_widened': int(a['widened']), 'audited_frames': int(a['audited_frames']),
                  'crops_per_heatmap': round(cs['crops'] / max(1, cs['heatmaps']), 4), 'noise_weights': os.environ.get('TTUP_TABLE_NOISE_WEIGHTS') == '1',
                  'seconds': round(time.time() - t0, 1)}))
