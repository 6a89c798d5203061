#!/usr/bin/env python3
"""BASELINE config 5: synthetic-dataset generation (drag+Magnus+contact RK4 on the device).
Reports seeds/s through sampler+integrator+selection, accepted trajectories/s, and the end-to-end time of
get_valid_trajectories(125 000) including the device->host copy and the reference-format dictionaries."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upliftingtabletennis_amd import trajgen
mode = os.environ.get('TTUP_TRAJ_MODE', 'final_lose'); direction = 'left_to_right'
n_seeds = int(os.environ.get('TTUP_TRAJ_SEEDS', '262144'))
trajgen.simulate_seeds(list(range(1024)), mode, direction)
torch.cuda.synchronize()
for sub in (4, 1):
    t0 = time.perf_counter()
    res = trajgen.simulate_seeds(np.arange(n_seeds), mode, direction, substeps=sub)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = int((res['n_keep'] > 0).sum().item())
    steps = float(res['n_saved'].double().sum().item()) * 2 * sub
    print('%s substeps=%d: %d seeds in %.3f s -> %.0f seeds/s, %d accepted (%.1f%%) -> %.0f trajectories/s; %.2e RK4 steps/s'
          % (mode, sub, n_seeds, dt, n_seeds / dt, acc, 100.0 * acc / n_seeds, acc / dt, steps / dt))
    del res
want = int(os.environ.get('TTUP_TRAJ_N', '125000'))
t0 = time.perf_counter()
tr = trajgen.get_valid_trajectories(want, 128, mode, direction, batches_per_launch=128)
dt = time.perf_counter() - t0
print('get_valid_trajectories(%d, 128, %s): %.2f s -> %.0f trajectories/s end to end (last seed %d)' % (want, mode, dt, want / dt, tr[-1]['seed']))
