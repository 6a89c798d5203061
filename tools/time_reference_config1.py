#!/usr/bin/env python3
"""wall time of the REFERENCE's own Python on BASELINE config 1 in the build container
(64 synthetic 1280x720 frames -> 62 triples -> WASBNet forward batch 1 per triple as interface.py:102-119 does -> table-variant
refine -> two-detector filter -> _uplifting_transform -> uplift net (B=1, T=50) -> transform_rotationaxes), torch CPU fp32.
The resize + normalise step uses this repo's restatement (cv2 is absent here).  Prints one line for BASELINE.md / DESIGN.md.
    python tools/time_reference_config1.py [n_frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (stubs + reference import helpers)

mg.install_stubs()
from oracle import glue_ref  # noqa: E402
from upliftingtabletennis_amd import synth, weights  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames, _ = synth.synth_frames(n, 720, 1280, seed=0)
model, _ = mg.ref_wasb(weights.random_wasb_state_dict(0, planted=True))
from tabledetection.helper_tabledetection import extract_position_torch_gaussian as extract_position_table  # noqa: E402
from uplifting.model import get_model  # noqa: E402
from uplifting.helper import transform_rotationaxes  # noqa: E402
up = get_model('connectstage', 'large', 'dynamic', 'new')
up.load_state_dict({k: torch.from_numpy(v) for k, v in weights.random_uplift_state_dict(0, 'large').items()}, strict=False)
up.eval()
torch.set_num_threads(os.cpu_count())
t0 = time.time()
pos = []
with torch.no_grad():
    for i in range(1, n - 1):
        x = glue_ref.triple_to_tensor(frames[i - 1], frames[i], frames[i + 1], (1280, 704))
        heat, _ = model(torch.from_numpy(x)[None])
        pos.append(extract_position_table(heat, 1920, 1080).squeeze(0))
t_det = time.time() - t0
pos = np.concatenate(pos, axis=0)
from upliftingtabletennis_amd import glue  # noqa: E402  (bit-equal to the reference's filter / transform: tests/test_cabi.py)
filt, _, times = glue.filter_trajectory_ball(pos, pos, 60.0)
_, table, _, _ = synth.synth_trajectories(1, 4, seed=0)
tk = np.array(table[0], dtype=np.float64) * np.array([1920, 1080, 1.0])
ball, tb, tm, mask = glue._uplifting_transform(filt[:49], tk, times[:49])
with torch.no_grad():
    rot, p3 = up(ball, tb, mask, tm)
    transform_rotationaxes(rot, p3.clone())
dt = time.time() - t0
print('reference (imported from /root/reference, torch %s CPU fp32, %d threads): %d frames -> %d triples in %.1f s (detector %.1f s) = %.3f frames/s'
      % (torch.__version__, torch.get_num_threads(), n, n - 2, dt, t_det, (n - 2) / dt))
