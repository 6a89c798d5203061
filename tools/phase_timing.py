#!/usr/bin/env python3
"""Phase timing of the 16-channel chain kernel from a TTUP_TIMING build (tools/build_ablate.sh TIMING; TTUP_LIB=...):
per-workgroup s_memtime stamps -> mean cycles per phase."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights, _lib
n = 8
frames, _ = synth.synth_frames(n + 2, 720, 1280, seed=0)
net = wasb.WASBNet(weights.random_wasb_state_dict(0, planted=True), resolution=(1280, 704), max_batch=n, dtype='bf16')
net.forward_frames(torch.from_numpy(frames).cuda())
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8192 * 8, dtype=np.uint64)
rc = lib.ttup_debug_read_timing(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
t = buf.reshape(8192, 8).astype(np.int64)
ok = t[:, 0] > 0
d = np.diff(t[ok][:, :7], axis=1)
names = ['stage+barrier', 'conv1', 'barrier1', 'conv2+barrier', 'conv3+barrier', 'conv4']
print('rc', rc, 'workgroups', int(ok.sum()))
for i, nm in enumerate(names):
    print('%-16s mean %8.0f  median %8.0f  p90 %8.0f cycles (100 MHz s_memtime ticks x?)' % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print('total mean', (t[ok][:, 6] - t[ok][:, 0]).mean())
rt = t[ok][:, 7]
print('in-kernel shader clock: %.0f MHz (s_memtime span / s_memrealtime span x 100 MHz, median over workgroups)' % np.median((t[ok][:, 6] - t[ok][:, 0]) / np.maximum(rt, 1) * 100.0))

buf2 = np.zeros(3 * 32 * 64 * 8, dtype=np.uint64)
lib.ttup_debug_read_timing_it(buf2.ctypes.data_as(ctypes.c_void_p), buf2.size)
t2 = buf2.reshape(3, 32, 64, 8).astype(np.int64)
for kid, name, labels in ((0, 'stem', ['(empty)', 'top barrier', 'conv1', 'barrier + X0 commit (next tile) + issue', 'conv2', 'epilogue+follower+stores (to next top)']),
                          (2, 'bb32 (last launch)', ['wait top barrier', 'commit+barrier+issue', 'conv1', 'barrier', 'conv2+epilogue (to next top)']),
                          (1, 'bneck', ['wait top barrier', 'phase 1', 'barrier', 'phase 2a', 'phase 2b', 'barrier', 'partials+barrier+reduce (to next top)'])):
    x = t2[kid][:, 2:(60 if kid < 2 else 9)]                    # skip warm-up iterations (the 32-channel block has ~11 tiles per workgroup)
    ok2 = (x[..., 0] > 0) & (x[..., 1] > 0)
    n_slots = len(labels)
    print(name, 'tiles sampled', int(ok2.sum()))
    tot = 0.0
    for i in range(n_slots):
        if i + 1 < n_slots:
            d2 = (x[..., i + 1] - x[..., i])[ok2]
        else:                                # last slot: until the next iteration's slot 0
            nxt = t2[kid][:, 3:(61 if kid < 2 else 10), 0]
            d2 = (nxt - x[..., i])[ok2 & (nxt > 0)]
        tot += d2.mean()
        print('  %-42s mean %7.0f  p90 %7.0f' % (labels[i], d2.mean(), np.percentile(d2, 90)))
    print('  total per tile %.0f cycles' % tot)
