import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import warnings; warnings.filterwarnings('ignore')
import hubconf
from upliftingtabletennis_amd import synth
frames, _ = synth.synth_frames(48, 720, 1280, seed=0)
images = [f for f in frames]
pipe = hubconf.full_pipeline()
for _ in range(3): pipe.predict(images, 60.0)
def t(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return r, min(ts), sum(ts) / len(ts)
cons = lambda k: pipe.table_detector_aux.filter_trajectory(k, k)
_, mn, av = t(lambda: pipe._clip_detections(images, True, cons)); print('clip_detections (ball+table+filter)  min %.1f avg %.1f ms' % (mn, av))
_, mn, av = t(lambda: pipe._clip_detections(images, False, None)); print('clip_detections (ball only)          min %.1f avg %.1f ms' % (mn, av))
_, mn, av = t(lambda: pipe._clip_detections(images, True, lambda k: k)); print('clip_detections (no DBSCAN)          min %.1f avg %.1f ms' % (mn, av))
_, mn, av = t(lambda: pipe.predict(images, 60.0)); print('predict                              min %.1f avg %.1f ms' % (mn, av))
fr = torch.from_numpy(frames).cuda()
_, mn, av = t(lambda: [pipe.table_detector.model.forward_frames(fr[i:i + 8], want_heatmap=False) for i in range(0, 48, 8)]); print('table net 48 frames (device)  min %.1f avg %.1f' % (mn, av))
_, mn, av = t(lambda: [pipe.ball_detector.model.forward_frames(fr[i:i + 10], want_heatmap=False) for i in range(0, 40, 8)]); print('ball net 40 triples chunks of 8 (device)  min %.1f avg %.1f' % (mn, av))
