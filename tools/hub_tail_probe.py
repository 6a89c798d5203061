import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TTUP_SYNTHETIC_WEIGHTS', '1')
import hubconf
from upliftingtabletennis_amd import synth, glue
images = [f for f in synth.synth_frames(48, 720, 1280, seed=0)[0]]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    hub = hubconf.full_pipeline()
for _ in range(3): hub.predict(images, 60.0)
pos, kp = hub._clip_detections(images, want_table=True, table_consumer=lambda k: hub.table_detector_aux.filter_trajectory(k, k))
torch.cuda.synchronize()
T = {}
for _ in range(20):
    t0 = time.perf_counter()
    filtered, _, times_ball = hub.ball_detector.filter_trajectory(pos, pos, 60.0)
    t1 = time.perf_counter()
    bc, tc, tm, mk = glue._uplifting_transform(filtered, np.asarray(kp, dtype=np.float64), times_ball)
    t2 = time.perf_counter()
    spin, p3 = hub.uplifting_model.predict_without_normalization(bc, tc, mk, tm)
    t3 = time.perf_counter()
    for k, v in (('filter_trajectory_ball', t1 - t0), ('_uplifting_transform', t2 - t1), ('uplift predict (upload, forward, rotation axes, download)', t3 - t2)):
        T.setdefault(k, []).append(v * 1e3)
for k, v in T.items(): print('%-60s %.3f ms' % (k, np.median(v)))
# inside predict: forward alone
m = hub.uplifting_model
b_, t_, m_, tm_ = [np.asarray(a, dtype=np.float32) for a in (bc, tc, mk, tm)]
n = m_.shape[-1]
print('n', n, 'valid', int(m_.sum()))
