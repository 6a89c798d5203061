#!/usr/bin/env python3
"""Speed of the fp32 parity path (conv_direct_f32_kernel) at crop sizes: what a certified-argmax re-evaluation costs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import wasb, weights
sd = weights.random_wasb_state_dict(0, planted=True)
for (w, h, b) in [(256, 256, 8), (256, 256, 32), (384, 384, 8), (1280, 704, 1)]:
    net = wasb.WASBNet(sd, resolution=(w, h), max_batch=b, dtype='f32')
    x = torch.randn(b, 9, h, w, device='cuda')
    net.forward(x, want_heatmap=False, want_peaks=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        net.forward(x, want_heatmap=False, want_peaks=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('f32 %dx%d batch %d: %.2f ms -> %.3f ms per crop, %.1f TFLOP/s' % (w, h, b, dt * 1e3, dt * 1e3 / b, 344.07e9 * (w * h / (1280 * 704)) * b / dt / 1e12), flush=True)
    del net
