#!/usr/bin/env python3
"""Per-WAVE phase stamps of one persistent kernel from a -DTTUP_TIMING -DTTUP_TIMING_WAVES=<id> build (0 stem, 1 Bottleneck tail,
2 32-channel block):   tools/build_ablate.sh TIMING,TIMING_WAVES=1 ;  TTUP_LIB=.../libttup_TIMING,TIMING_WAVES=1.so python3 tools/wave_timing.py
Prints, per wave, the mean cycles between consecutive stamps of a tile and the arrival skew at every stamp relative to wave 0."""
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
buf = np.zeros(3 * 32 * 64 * 8, dtype=np.uint64)
lib.ttup_debug_read_timing_it(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
t = buf.reshape(8, 12, 64, 8).astype(np.int64)          # [wave][workgroup][iteration][slot]
nslot = int(sys.argv[1]) if len(sys.argv) > 1 else 7
x = t[:, :, 3:40, :nslot]
ok = (x > 0).all(axis=(0, 3))                              # (workgroup, iteration) with every wave's stamps
print('tiles sampled', int(ok.sum()))
for w in range(8):
    d = np.diff(x[w], axis=-1)[ok]
    nxt = (t[w, :, 4:41, 0] - x[w][..., nslot - 1])[ok]
    skew = (x[w] - x[0])[ok].mean(axis=0)
    print('wave %d  phases %s  to next top %6.0f | arrival vs wave 0 at each stamp %s' % (w, ' '.join('%6.0f' % v for v in d.mean(axis=0)), nxt.mean(), ' '.join('%6.0f' % v for v in skew)))
