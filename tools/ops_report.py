#!/usr/bin/env python3
"""Per-op and per-kernel timings of one CNN micro-batch of the bench workload (1280x704 network input), measured with HIP
events between consecutive ops of the graph inside the library (ttup_wasb_time_graph).  TTUP_LIB selects another build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights

n = 8
frames, _ = synth.synth_frames(n + 2, 720, 1280, seed=0)
net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=n, dtype='bf16')
fr = torch.from_numpy(frames).cuda()
net.forward_frames(fr)
torch.cuda.synchronize()
ops = wasb.time_ops(net, reps=int(os.environ.get('TTUP_REPS', '10')), in_graph=True)
mb = ops[0]['batch']
tot = sum(o['ms'] for o in ops)
print(os.environ.get('TTUP_LIB', 'default'), 'total %.4f ms per micro-batch of %d (%.4f ms/frame)' % (tot, mb, tot / mb))
if os.environ.get('TTUP_ALL'):
    for o in ops:
        print('  op%-3d %-44s cin%4d cout%3d k%d s%d %4dx%4d  %.4f ms  %6.1f TFLOP/s' % (o['index'], o['kernel'], o['cin'], o['cout'], o['k'], o['stride'],
              o['h'], o['w'], o['ms'], o['flops'] / (o['ms'] * 1e-3) / 1e12 if o['ms'] > 0 else 0))
groups = {}
for o in ops:
    g = groups.setdefault(o['kernel'].split('+')[0], [0, 0.0, 0.0])
    g[0] += 1; g[1] += o['ms']; g[2] += o['flops']
for k, (cnt, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print('  %-40s x%-3d %.4f ms  %5.1f%%  %6.1f TFLOP/s' % (k, cnt, ms, 100 * ms / tot, fl / (ms * 1e-3) / 1e12))
