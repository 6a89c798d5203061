#!/usr/bin/env python3
"""Hub pipeline (TableTennisPipeline) on changing content, clip lengths and frame sizes with exact windows (TTUP_EXACT_WINDOWS=1):
every ball position must equal the one the full-frame fp32 path gives, and predict() must return finite results.  Clips stay below
50 detections: from 50 on the reference's _uplifting_transform builds an all-ones mask and its model raises ValueError
(inference/utils.py:304-307, uplifting/model.py:541-546) -- kept, so predict raises there too."""
import os, sys, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TTUP_SYNTHETIC_WEIGHTS'] = '1'
os.environ['TTUP_EXACT_WINDOWS'] = '1'
import hubconf
from upliftingtabletennis_amd import _lib, refine, synth, wasb
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hub = hubconf.full_pipeline()
bd = hub.ball_detector
sd = bd.model._state_dict
twin = wasb.WASBNet(sd, resolution=(1280, 704), max_batch=1, dtype='f32')
rng = np.random.default_rng(7)
bad = tot = 0
for c, (n, h, w) in enumerate([(48, 720, 1280), (3, 720, 1280), (4, 720, 1280), (25, 720, 1280), (49, 720, 1280), (30, 1080, 1920), (26, 704, 1280), (50, 720, 1280), (27, 540, 960), (51, 720, 1280)]):
    frames, _ = synth.synth_frames(n, h, w, seed=500 + c, sigma=float(rng.uniform(1.2, 4.0)))
    frames = np.clip(np.rint(frames.astype(np.float32) * float(rng.uniform(0.6, 1.6))), 0, 255).astype(np.uint8)
    images = [f for f in frames]
    pos, kp = hub._clip_detections(images, want_table=True)
    fr = torch.from_numpy(frames).cuda()
    x = wasb.preprocess_triples(fr, (1280, 704))
    idx, win = [], []
    for k in range(x.shape[0]):
        _, i1, w1 = twin.forward(x[k:k + 1], want_heatmap=False, want_peaks=True)
        idx.append(i1); win.append(w1)
    ref = refine.refine_windows_device(torch.cat(idx), torch.cat(win), 704, 1280, 1920, 1080, _lib.REFINE_TABLE).cpu().numpy()
    d = np.abs(pos - ref).max() if len(ref) else 0.0
    nb = int((np.abs(pos - ref).max(1) > 0).sum()) if len(ref) else 0
    bad += nb; tot += len(ref)
    spin, p3 = hub.predict(images, 60.0)
    print('clip %d: %d frames %dx%d: %d of %d positions differ from the fp32 path (max %.3g); predict -> spin %s pos3d %s finite %s; eps %.4f' % (
        c, n, w, h, nb, len(ref), d, tuple(spin.shape), tuple(p3.shape), bool(np.isfinite(p3).all()), bd.model.eps), flush=True)
print('total: %d of %d positions differ' % (bad, tot))
