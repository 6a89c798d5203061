#!/usr/bin/env python3
"""What the certified argmax costs on the bench's varied content, on ONE box: the pipelined step (256 triples, four alternating
clips with blob sigma 1.3 / 2 / 3 / 4 px) with the certification off, on, and -- isolated on an idle GPU -- the fp32 crop pass
itself: ms per pass of 64 crops, per crop, per empty pass (launch cost of a provisioned pass without crops).
    python tools/cert_cost.py          (TTUP_F32_EXACT=1 for the fp32-MFMA kernels instead of the split-bf16 ones)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import pipeline, synth, wasb, weights  # noqa: E402

dev = torch.device('cuda:0')
T = 256
sd, usd = weights.random_wasb_state_dict(0, planted=True), weights.random_uplift_state_dict(0, 'large')
table_px = np.concatenate([np.random.default_rng(0).uniform(100, 900, (13, 2)), np.ones((13, 1))], 1)
clips = []
for c, (sigma, gain) in enumerate(((1.3, 0.7), (2.0, 1.0), (3.0, 1.3), (4.0, 1.6))):
    base, _ = synth.hard_clip(34, 720, 1280, seed=100 + c, sigma=sigma, gain=gain)
    clips.append(torch.from_numpy(np.concatenate([base] * 8)[:T + 2]).to(dev))
out = {'f32_kernels': 'fp32-MFMA (TTUP_F32_EXACT)' if os.environ.get('TTUP_F32_EXACT') else 'split-bf16 (conv_x3)'}


def run(certify, k=8):
    w = pipeline.StreamWorker(dev, sd, usd, net_wh=(1280, 704), max_triples=T, traj_len=120, seq_len=121, certify=certify)
    for cl in clips:
        w.collect(w.submit(cl), table_px, 60.0)
    if certify:
        w.net.certify_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tk = None
    for i in range(k):
        nx = w.submit(clips[i % 4])
        if tk is not None:
            w.collect(tk, table_px, 60.0)
        tk = nx
    w.collect(tk, table_px, 60.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    r = {'ms_per_step': round(dt * 1e3, 2), 'fps': round(T / dt, 1)}
    if certify:
        cs = w.net.certify_stats()
        r.update(crops_per_step=cs['crops'] / k, crops_per_heatmap=round(cs['crops'] / max(1, cs['heatmaps']), 3), eps=round(w.certify_eps, 5))
    del w
    torch.cuda.empty_cache()
    return r


out['pipeline_no_certify'] = run(False)
out['pipeline_certified'] = run(True)
# the crop net alone: a 64-crop fp32 pass at 168x168 on an idle GPU
os.environ['TTUP_MICRO_BATCH'] = '64'
net = wasb.WASBNet(sd, resolution=(168, 168), max_batch=64, dtype='f32', lanes=1)
x = torch.randn((64, 9, 168, 168), device=dev)
net.forward(x, want_heatmap=False, want_peaks=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    net.forward(x, want_heatmap=False, want_peaks=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
out['crop_pass_alone'] = {'ms_per_pass_of_64': round(dt * 1e3, 3), 'ms_per_crop': round(dt * 1e3 / 64, 4), 'tflops': round(64 * 344.07e9 * 168 * 168 / (1280 * 704) / dt / 1e12, 1)}
d = out['pipeline_certified']['ms_per_step'] - out['pipeline_no_certify']['ms_per_step']
out['certification_ms_per_step'] = round(d, 2)
out['certification_ms_per_crop_in_pipeline'] = round(d / max(1.0, out['pipeline_certified']['crops_per_step']), 4)
print(json.dumps(out))
