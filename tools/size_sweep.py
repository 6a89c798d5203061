#!/usr/bin/env python3
"""bf16 network against the CPU oracle on a dozen heights / widths around the kernels' tile sizes (24x32, 22x30, 8x32): edge masking,
ragged 16-pixel groups, partial row bands.  Run on a GPU box from the repo root."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import wasb, weights
from oracle import wasb_ref
bad = 0
for (h, w) in [(24, 32), (48, 64), (56, 72), (16, 40), (32, 24), (80, 40), (120, 200), (96, 96), (8, 136), (200, 8), (64, 264), (184, 104)]:
    sd = weights.random_wasb_state_dict(h * 1000 + w)
    x = np.random.default_rng(h + w).standard_normal((2, 9, h, w)).astype(np.float32)
    ref = wasb_ref.wasb_forward(x, sd).numpy()
    scale = ref.max() - ref.min()
    net = wasb.WASBNet(sd, resolution=(w, h), max_batch=2, dtype='bf16')
    heat, idx, win = net.forward(torch.from_numpy(x), want_peaks=True)
    got = heat.cpu().numpy()
    err = np.abs(got - ref).max() / scale
    ok = err <= 4e-2 and np.array_equal(idx.cpu().numpy(), got.reshape(2, -1).argmax(1))
    bad += not ok
    print(h, w, 'err/scale %.4f' % err, 'ok' if ok else 'FAIL')
print('bad', bad)
