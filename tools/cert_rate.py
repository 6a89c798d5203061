#!/usr/bin/env python3
"""How often the certified argmax needs fp32 crops on the bench clip (planted weights), and the calibrated eps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights
base, _ = synth.synth_frames(34, 720, 1280, seed=0)
clip = torch.from_numpy(np.concatenate([base] * 8)[:258]).cuda()
net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=256, dtype='bf16')
eps = net.calibrate(clip, n=4)
hb, _, _ = net.forward_frames(clip[:10], want_heatmap=True)
net.certify_stats(reset=True)
net.forward_frames(clip)
print('eps_abs %.5f (heat range %.3f) -> %s' % (eps, float(hb.max() - hb.min()), net.certify_stats()))
