#!/usr/bin/env python3
"""Short CNN-only workload for rocprofv3 (kernel trace / PMC passes): one micro-batch of triples, a few repeats.
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ... -d out -- python3 tools/prof_cnn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights  # noqa: E402

n = int(os.environ.get('TTUP_PROF_TRIPLES', '8'))
reps = int(os.environ.get('TTUP_PROF_REPS', '3'))
frames, _ = synth.synth_frames(min(n + 2, 10), 720, 1280, seed=0)
import numpy as np  # noqa: E402
clip = np.concatenate([frames] * ((n + 2 + len(frames) - 1) // len(frames)))[:n + 2]
net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=n, dtype='bf16')
fr = torch.from_numpy(clip).cuda()
for _ in range(reps):
    net.forward_frames(fr, want_heatmap=False)
torch.cuda.synchronize()
print('done')
