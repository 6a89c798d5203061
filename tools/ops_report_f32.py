#!/usr/bin/env python3
"""Per-op timings of the fp32 path (parity twin / certification crops): one full frame, and a batch of 168x168 crops."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import wasb, weights  # noqa: E402

sd = weights.random_wasb_state_dict(0, planted=True)
for (w, h, b) in ((1280, 704, 1), (168, 168, 64)):
    os.environ['TTUP_MICRO_BATCH'] = str(b)
    net = wasb.WASBNet(sd, resolution=(w, h), max_batch=b, dtype='f32')
    x = torch.randn((b, 9, h, w), device='cuda')
    net.forward(x)
    torch.cuda.synchronize()
    ops = wasb.time_ops(net, batch=b, reps=3, in_graph=True)
    tot = sum(o['ms'] for o in ops)
    fl = sum(o['flops'] for o in ops)
    print('%dx%d batch %d: %.3f ms per micro-batch of %d, %.1f TFLOP/s' % (w, h, b, tot, ops[0]['batch'], fl / (tot * 1e-3) / 1e12))
    groups = {}
    for o in ops:
        key = (o['cin'], o['cout'], o['k'], o['stride'], o['h'], o['w'])
        g = groups.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1; g[1] += o['ms']; g[2] += o['flops']
    for k, (cnt, ms, f) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TTUP_TOP', '16'))]:
        print('   cin%4d cout%4d k%d s%d %4dx%4d x%-2d %.4f ms %5.1f%% %6.1f TFLOP/s' % (*k, cnt, ms, 100 * ms / tot, f / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    del net
