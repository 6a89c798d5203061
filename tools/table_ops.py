#!/usr/bin/env python3
"""Per-op timings of the table detector's graph (3 input channels, 13 heads: the head is a separate kernel) beside the ball detector's on
the same box: replay time per micro-batch of 8, the six longest ops, the first and last ops.  (`forward(x)` entry: the stem reads fp32 NCHW.)"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import warnings; warnings.filterwarnings('ignore')
import numpy as np, torch
from upliftingtabletennis_amd import wasb, weights
for name, mk in (('table', lambda: wasb.MyHRNet(weights.random_wasb_state_dict(5, in_ch=3, head_out=13), resolution=(1280, 704), max_batch=8, dtype='bf16')),
                 ('ball', lambda: wasb.WASBNet(weights.random_wasb_state_dict(5), resolution=(1280, 704), max_batch=8, dtype='bf16'))):
    net = mk()
    x = torch.randn((8, net.IN_CH, 704, 1280), device='cuda')
    net.forward(x, want_peaks=True)
    ops = wasb.time_ops(net, reps=5)
    rp = wasb.time_replay(net, reps=10)
    print(name, 'replay %.3f ms per 8, op-event sum %.3f' % (rp, sum(o['ms'] for o in ops)))
    for o in sorted(ops, key=lambda o: -o['ms'])[:6]:
        print('   %-60s %.4f' % (o['kernel'], o['ms']))
    print('   last ops:', [(o['kernel'][:40], round(o['ms'], 4)) for o in ops[-3:]], ' first:', [(o['kernel'][:30], round(o['ms'], 4)) for o in ops[:2]])
