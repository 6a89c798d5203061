#!/usr/bin/env python3
"""Print per-op timings of one CNN micro-batch (HIP events inside the library).  TTUP_LIB selects an alternative build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights
import numpy as np
n = 8
frames, _ = synth.synth_frames(n + 2, 720, 1280, seed=0)
net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=n, dtype='bf16')
fr = torch.from_numpy(frames).cuda()
net.forward_frames(fr)
torch.cuda.synchronize()
ops = wasb.time_ops(net, reps=5)
mb = ops[0]['batch']
top = int(os.environ.get('TTUP_TOP', '6'))
print(os.environ.get('TTUP_LIB', 'default'), 'total %.4f ms/frame' % (sum(o['ms'] for o in ops) / mb))
for o in ops[:top]:
    print('  op%-3d %-11s cin%5d cout%3d k%d s%d %4dx%4d  %.4f ms/frame' % (o['index'], o['kind'], o['cin'], o['cout'], o['k'], o['stride'], o['h'], o['w'], o['ms'] / mb))
