#!/usr/bin/env python3
"""BASELINE config 3: uplift-only throughput (trajectories/s) on synthetic trajectories; prints per-kernel share via torch timers."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import synth, uplift, weights
B = int(os.environ.get('TTUP_UPLIFT_B', '10000')); T = int(os.environ.get('TTUP_UPLIFT_T', '120'))
ball, table, mask, times = [torch.from_numpy(a).cuda() for a in synth.synth_trajectories(min(B, 2000), T, seed=0, pad=1)]
rep = (B + ball.shape[0] - 1) // ball.shape[0]
ball, table, mask, times = [a.repeat((rep,) + (1,) * (a.dim() - 1))[:B].contiguous() for a in (ball, table, mask, times)]
net = uplift.get_model('connectstage', 'large', 'dynamic', 'new', state_dict=weights.random_uplift_state_dict(0, 'large'), max_batch=B, max_len=T + 1)
net(ball[:64], table[:64], mask[:64], times[:64])
torch.cuda.synchronize()
t0 = time.perf_counter()
rot, pos = net(ball, table, mask, times)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gflop = 2.0 if T >= 100 else 0.75
print('uplift B=%d T=%d: %.3f s  -> %.0f trajectories/s  (~%.1f TFLOP/s at %.2f GFLOP/trajectory)' % (B, T, dt, B / dt, B * gflop / dt / 1e3, gflop))
if B <= 64:          # latency of a small call (what the hub surface and the pipeline's per-clip uplift pay): mean of 20 calls on a side stream
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            net(ball, table, mask, times)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            net(ball, table, mask, times)
            st.synchronize()
        print('   small-batch latency (host wall clock, synchronised per call): %.3f ms per forward; %s' % ((time.perf_counter() - t0) / 20 * 1e3, net.graph_info()))
