#!/usr/bin/env python3
"""The headline pipeline on pure NOISE weights (bench.py's noise_weights_fps leg alone): frames/s, crops per heatmap, share of
heatmaps that overflow the crop budget, full-frame fp32 re-runs per step.  TTUP_CERT_MAXC / TTUP_CERT_LIST: crops per heatmap /
crop-list capacity per heatmap (csrc/certify.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda:0')
pn = bench.Pipeline(dev, seed=0, certify=True, planted=False)
for _ in range(2): pn.step()
torch.cuda.synchronize()
pn.net.certify_stats(reset=True); r0 = pn.worker.fp32_reruns
k = 4; t0 = time.perf_counter(); tk = None
for _ in range(k):
    nx = pn.submit()
    if tk is not None: pn.collect(tk)
    tk = nx
pn.collect(tk); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / k
cs = pn.net.certify_stats()
print('maxc %s list %s: %.1f fps, %.1f ms/step, crops/heatmap %.3f, not certified %.4f (candidate list %d, crops per heatmap %d, crop list %d of %d), reruns/step %.1f' % (os.environ.get('TTUP_CERT_MAXC','8'), os.environ.get('TTUP_CERT_LIST','4'), 256/dt, dt*1e3, cs['crops']/cs['heatmaps'], cs['not_certified']/cs['heatmaps'], cs['over_candidates'], cs['over_crops_per_map'], cs['over_crop_list'], cs['heatmaps'], (pn.worker.fp32_reruns - r0)/k))
