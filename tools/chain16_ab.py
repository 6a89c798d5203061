#!/usr/bin/env python3
"""The round-6 16-channel chain kernel (c16_chain_kernel, csrc/chain16.h) against the run-time-epilogue form of the round-2..5 kernel
(bb_chain2_kernel, TTUP_BB2_GENERIC=1, read once per process: child processes) on the same weights and inputs: fuse-layer outputs of
stages 2-4 and the heatmap, on a ragged all-border size, a size with interior tiles, the bench size and the 13-channel table net.
Same rounding points; the new kernel adds the fuse-layer terms on the matrix pipe (another fp32 summation order), so sums may land on
the neighbouring bf16 value: reported are the share of differing values, the largest difference relative to the tap's range, and the
argmax indices.  (Until the compiled-out forms of the old kernel were deleted in round 6 the comparison ran against those: stage taps
bit-equal, heatmap within 1e-7 of its range.)"""
import os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from upliftingtabletennis_amd import wasb, weights
out = {}
for k, (h, w, b, seed, table) in enumerate(((104, 168, 3, 43, 0), (288, 512, 5, 44, 0), (704, 1280, 2, 45, 0), (288, 512, 2, 46, 1), (96, 160, 3, 47, 0), (64, 32, 2, 48, 0))):
    if table:
        sd = weights.random_wasb_state_dict(seed, in_ch=3, head_out=13)
        net = wasb.MyHRNet(sd, resolution=(w, h), max_batch=b, dtype='bf16')
        x = torch.from_numpy(np.random.default_rng(seed).standard_normal((b, 3, h, w)).astype(np.float32))
        heat = net(x)
        out['heat%%d' %% k] = heat.cpu().numpy()
    else:
        sd = weights.random_wasb_state_dict(seed)
        net = wasb.WASBNet(sd, resolution=(w, h), max_batch=b, dtype='bf16')
        x = torch.from_numpy(np.random.default_rng(seed).standard_normal((b, 9, h, w)).astype(np.float32))
        heat, idx, _ = net.forward(x, want_peaks=True)
        out['heat%%d' %% k] = heat.cpu().numpy(); out['idx%%d' %% k] = idx.cpu().numpy()
    for tap in ('stage2_0', 'stage2_1', 'stage3_0', 'stage3_1', 'stage3_2'):
        out['%%s_%%d' %% (tap, k)] = net.read_tap(tap, b).cpu().numpy()
np.savez(sys.argv[1], **out)
''' % ROOT


def main():
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        # `python tools/chain16_ab.py stream`: the streaming form (csrc/experiments/chain16s.h.inc -- not compiled; when it is wired in
        # again, TTUP_C16_STREAM=1 selects it) against the tile form instead
        pair = (('new', {'TTUP_C16_STREAM': '1'}), ('old', {})) if sys.argv[1:] == ['stream'] else (('new', {}), ('old', {'TTUP_BB2_GENERIC': '1'}))
        for tag, env in pair:
            e = dict(os.environ); e.pop('TTUP_BB2_GENERIC', None); e.pop('TTUP_C16_STREAM', None); e.update(env)
            f = os.path.join(d, tag + '.npz')
            r = subprocess.run([sys.executable, '-c', CHILD, f], env=e, capture_output=True, text=True, timeout=1800)
            if r.returncode != 0:
                print(tag, 'FAILED'); print(r.stderr[-3000:]); sys.exit(1)
            outs[tag] = dict(np.load(f))
    bad = 0
    for k in sorted(outs['new']):
        a, g = outs['new'][k], outs['old'][k]
        if k.startswith('idx'):
            eq = np.array_equal(a, g)
            print('%-12s indices equal: %s' % (k, eq)); bad += not eq
            continue
        rng = float(g.max() - g.min()) or 1.0
        diff = np.abs(a.astype(np.float64) - g)
        print('%-12s shape %-22s differing %.5f  max|d|/range %.3e  mean|d|/range %.3e  nan %d' % (k, a.shape, float((a != g).mean()), diff.max() / rng, diff.mean() / rng, int(np.isnan(a).sum())))
        bad += not (diff.max() / rng <= 2.0 ** -5 and not np.isnan(a).any())
    print('OK' if not bad else 'MISMATCH (%d)' % bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
