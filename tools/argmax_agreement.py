#!/usr/bin/env python3
"""bf16 production path vs fp32 path at 704x1280: argmax agreement rate, heatmap error, and how many pixels sit within
2*eps of the bf16 maximum (what a certified argmax would have to re-evaluate).  Weight sets: noise, planted (eps 0.2 and 1.0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, wasb, weights
N = int(os.environ.get('TTUP_AGREE_FRAMES', '16'))
frames, _ = synth.synth_frames(N + 2, 720, 1280, seed=5)
fr = torch.from_numpy(frames).cuda()
for label, sd in (('noise', weights.random_wasb_state_dict(0, planted=False)), ('planted eps=0.2', weights.random_wasb_state_dict(0, planted=True)),
                  ('planted eps=1.0', weights.random_wasb_state_dict(0, planted=True, eps=1.0))):
    nb = wasb.WASBNet(sd, resolution=(1280, 704), max_batch=8, dtype='bf16')
    nf = wasb.WASBNet(sd, resolution=(1280, 704), max_batch=1, dtype='f32')
    agree, errs, rng_, cands, gaps = 0, [], [], [], []
    for t0 in range(0, N, 8):
        hb, ib, _ = nb.forward_frames(fr[t0:t0 + 10], want_heatmap=True)
        x = wasb.preprocess_triples(fr[t0:t0 + 10], (1280, 704))
        for k in range(8):
            hf, i_f, _ = nf.forward(x[k:k + 1], want_peaks=True)
            a, b = hb[k, 0], hf[0, 0]
            e = (a - b).abs().max().item(); r = (b.max() - b.min()).item()
            errs.append(e); rng_.append(r)
            agree += int(ib[k].item() == i_f[0].item())
            for mult in (2.0, 3.0):
                cands.append(int((a >= a.max() - mult * e).sum().item()))
            # true margin: fp32 max minus the best fp32 value outside the 5x5 neighbourhood of the fp32 argmax
            iy, ix = int(i_f[0]) // 1280, int(i_f[0]) % 1280
            c = b.clone(); c[max(0, iy - 2):iy + 3, max(0, ix - 2):ix + 3] = -1e30
            gaps.append((b.max() - c.max()).item() / r)
    errs, rng_ = np.array(errs), np.array(rng_)
    c2, c3 = np.array(cands[0::2]), np.array(cands[1::2])
    print('%-16s agreement %d/%d  max err/range: max %.4f mean %.4f  candidates within 2*err: median %d max %d; within 3*err: median %d max %d; fp32 peak margin outside 5x5 / range: min %.3f median %.3f'
          % (label, agree, N, (errs / rng_).max(), (errs / rng_).mean(), np.median(c2), c2.max(), np.median(c3), c3.max(), min(gaps), np.median(gaps)), flush=True)
    del nb, nf
