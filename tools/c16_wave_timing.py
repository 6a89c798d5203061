#!/usr/bin/env python3
"""Per-wave phase stamps of the 16-channel chain kernel (csrc/chain16.h, build: tools/build_ablate.sh TIMING,TIMING_C16W; run with
TTUP_LIB=.../libttup_TIMING,TIMING_C16W.so).  The buffer holds the LAST launch of the kernel (stage 4: the head form).  Printed: mean
cycles per wave between consecutive stamps -- a wave's own work per phase, and what it waits at each barrier."""
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
lib.ttup_debug_read_timing(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
t = buf.reshape(512, 8, 16).astype(np.int64)
ok = (t[:, :, 0] > 0).all(axis=1)
t = t[ok]
names = ['issue loads + wait + LDS stores', 'barrier 0', 'conv1', 'barrier 1', 'conv2', 'barrier 2', 'conv3', 'barrier 3', 'conv4 + epilogue', 'argmax reduce (+barrier)', 'stores drain']
d = np.diff(t[:, :, :12], axis=2)
print('workgroups', int(ok.sum()))
print('%-34s' % 'phase' + ''.join('  wave%d' % w for w in range(8)) + '     max-wave')
for i, nm in enumerate(names):
    print('%-34s' % nm + ''.join(' %6.0f' % d[:, w, i].mean() for w in range(8)) + '   %8.0f' % d[:, :, i].max(axis=1).mean())
print('%-34s' % 'total' + ''.join(' %6.0f' % (t[:, w, 11] - t[:, w, 0]).mean() for w in range(8)))
span = (t[:, :, 11].max(axis=1) - t[:, :, 0].min(axis=1))
print('workgroup span mean %.0f cycles; start skew between waves %.0f' % (span.mean(), (t[:, :, 0].max(axis=1) - t[:, :, 0].min(axis=1)).mean()))
