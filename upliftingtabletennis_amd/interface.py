"""Drop-in boundary: the classes of the reference's ``interface.py`` (:83-312) for the ball-detection ->
refine -> uplift path, same constructor arguments, method names, argument meaning, return types and errors.

Differences forced by the environment (documented in DESIGN.md):
  * no network: weights come from ``TTUP_WEIGHTS`` (a directory laid out like the reference's weight zip:
    inference_balldetection/<name>/model.pt, inference_uplifting/ours/model.pt) or from the folder the reference
    unpacks that zip into under the torch hub directory; when neither exists the constructors raise the reference's
    RuntimeError unless ``TTUP_SYNTHETIC_WEIGHTS=1`` asks for the seeded generators in ``weights.py`` (with a warning);
  * only the in-tree WASB/HRNet detector is built; 'segformerpp_*' needs the un-vendored
    KieDani/SegformerPlusPlus hub repo and raises NotImplementedError;
  * table detection uses the in-tree HRNet ('hrnet'); when a detector's primary SegFormer++ model is unavailable the
    auxiliary in-tree model stands in for both sides of the two-detector agreement filter.
Quirks kept on purpose: BGR frames are fed to the detector as they come (interface.py:96,104-110); the
*table* variant of the refine is used on the hub surface (interface.py:116); visibility is always 1.
"""
import os

import numpy as np
import torch

from . import _lib, calib, glue, refine, uplift, wasb, weights

HEIGHT, WIDTH = 1080, 1920
KEYPOINT_VISIBLE = 1


def _weights_dir():
    """Where reference-format checkpoints are looked for: $TTUP_WEIGHTS, else the folder the reference itself unpacks its
    weight archive into (interface.py:34-73: <torch hub dir>/checkpoints/tt_uplifting_extracted/weights)."""
    d = os.environ.get('TTUP_WEIGHTS', '')
    if d:
        return d
    hub = os.path.join(torch.hub.get_dir(), 'checkpoints', 'tt_uplifting_extracted', 'weights')
    return hub if os.path.isdir(hub) else ''


def _synthetic_or_raise(what, path):
    """No checkpoint: the reference downloads one or raises RuntimeError (interface.py:61,71).  There is no network here,
    so the same RuntimeError is raised unless TTUP_SYNTHETIC_WEIGHTS=1 explicitly asks for seeded random weights
    (benchmarks / smoke tests: the outputs are then meaningless as detections)."""
    if os.environ.get('TTUP_SYNTHETIC_WEIGHTS') == '1':
        import warnings
        warnings.warn('upliftingtabletennis_amd: %s runs on SEEDED RANDOM weights (TTUP_SYNTHETIC_WEIGHTS=1); its outputs '
                      'are not detections' % what, RuntimeWarning, stacklevel=3)
        return
    raise RuntimeError('Failed to download weights: %s not found and there is no network; point TTUP_WEIGHTS at a folder laid '
                       'out like the reference weight archive, or set TTUP_SYNTHETIC_WEIGHTS=1 for seeded random weights' % (path or what))


def _load_ball_checkpoint(model_name):
    """-> (state_dict, resolution (W,H), in_frames).  Reference: inference_balldetection.load_model :40-61."""
    path = os.path.join(_weights_dir(), 'inference_balldetection', model_name, 'model.pt')
    if _weights_dir() and os.path.exists(path):
        sd, info = weights.load_checkpoint_state_dict(path)
        return sd, tuple(info.get('image_resolution', wasb.RESOLUTIONS['wasb'])), int(info.get('in_frames', 3))
    _synthetic_or_raise("BallDetector('%s')" % model_name, path if _weights_dir() else '')
    return weights.random_wasb_state_dict(int(os.environ.get('TTUP_SEED', '0')), planted=True), wasb.RESOLUTIONS['wasb'], 3


def _load_uplift_checkpoint():
    """-> (state_dict, size, transform_mode).  Reference: inference_uplifting.load_model :33-58."""
    path = os.path.join(_weights_dir(), 'inference_uplifting', 'ours', 'model.pt')
    if _weights_dir() and os.path.exists(path):
        sd, info = weights.load_checkpoint_state_dict(path)
        if info.get('name', 'connectstage') != 'connectstage' or info.get('tabletoken_mode', 'dynamic') != 'dynamic':
            raise ValueError('only connectstage/dynamic uplift checkpoints are supported')
        if info.get('time_rotation', 'new') != 'new':       # the reference hands this to get_model (inference_uplifting.py:49-52)
            raise ValueError("only time_rotation='new' uplift checkpoints are supported (got %r)" % info.get('time_rotation'))
        return sd, info.get('size', 'large'), info.get('transform_mode', 'global')
    _synthetic_or_raise('UpliftingModel()', path if _weights_dir() else '')
    return weights.random_uplift_state_dict(int(os.environ.get('TTUP_SEED', '0')), 'large'), 'large', 'global'


class _CertifiedDetector:
    """What the two detectors share: the certified argmax (csrc/certify.hip) of a bf16 handle -- first estimate of the error bound
    eps on the first input, the running audit, widening + re-certification, repair of over-budget heatmaps on the fp32 twin.
    `self.model` is a wasb.WASBNet (one heatmap per triple) or a wasb.MyHRNet (13 keypoint heatmaps per frame)."""
    AUDIT_EVERY = 256          # samples per audited sample (see _audit_picks)
    NO_CERTIFY_ENV = ('TTUP_NO_CERTIFY',)

    def _calibrate(self, frames=None, x=None):
        """A first estimate of the bf16 path's error bound eps is measured on the first input this detector sees, against the fp32
        path; the audits (`_audit_picks`, the crops' candidate errors) keep checking it on later inputs and widen it when one comes
        within the safety factor.  Every returned index is the fp32 argmax as long as eps bounds the frame's error."""
        m = self.model
        if m.certified or m.dtype != 'bf16' or any(os.environ.get(k) == '1' for k in self.NO_CERTIFY_ENV):
            return
        exact = os.environ.get('TTUP_EXACT_WINDOWS') == '1'
        if frames is None:
            n = min(2, x.shape[0])
            hb = m._heat(x[:n])
            twin = m._twin()
            err = max(float(wasb.max_abs_diff(hb[k], twin._heat(x[k:k + 1])[0]).item()) for k in range(n))
            m.set_certify(m.HEADROOM * err, exact_windows=exact)
            m.audit_state = dict(audited_frames=n, max_err_seen=err, widened=0)
        else:
            m.calibrate(frames, n=4, exact_windows=exact)

    def _audit_picks(self, n_samples):
        """One random sample per AUDIT_EVERY samples this detector has processed is re-run on the fp32 twin (eps audit)."""
        if not self.model.certified or n_samples <= 0 or os.environ.get('TTUP_NO_AUDIT') == '1':
            return []
        rng = self.__dict__.setdefault('_audit_rng', np.random.default_rng(0))
        self._since_audit = self.__dict__.get('_since_audit', 0) + n_samples
        picks = []
        while self._since_audit >= self.AUDIT_EVERY:
            self._since_audit -= self.AUDIT_EVERY
            picks.append(int(rng.integers(n_samples)))
        return picks

    def _certified_forward(self, x):
        """(heat, idx, win) of the float input x (the `self.model(x)` seam of `predict`): peaks from the certified argmax -- the fp32
        index the reference's torch.argmax returns -- under an audited eps."""
        m = self.model
        self._calibrate(x=x)
        picks = self._audit_picks(x.shape[0])
        while True:
            heat, idx, win = wasb.WASBNet.forward(m, x, want_heatmap=True, want_peaks=True)
            if not m.certified:
                return heat, idx, win
            status, info = m.certify_status(idx.shape[0]), m.certify_info()          # (masked status: the float entry re-runs the whole chunk on a widening)
            err = m.note_error(m.decode_info(info.cpu().numpy())[1])
            for t in picks:        # eps audit: the bf16 heatmaps of a random sample against the fp32 twin
                err = max(err, m.note_error(float(wasb.max_abs_diff(heat[t], m._twin()._heat(x[t:t + 1])[0]).item()), 1))
            picks = []
            if m.eps_violated(err):
                m.widen_eps(err)
                continue
            m.fix_uncertified(idx, win, x=x, status=status)
            return heat, idx, win

    def _certified_peaks(self, fr):
        """(idx, win) of the samples of the uint8 device clip `fr` (one forward call), certified under an audited eps."""
        m = self.model
        picks = self._audit_picks(fr.shape[0] - (m.NF - 1))
        while True:
            audit = m.audit_async(fr, picks) if picks else None
            eps_used = m.eps if m.certified else None
            _, idx, win = m.forward_frames(fr, want_heatmap=False)
            if not m.certified:
                return idx, win
            status, info = m.certify_status(idx.shape[0], raw=True), m.certify_info()
            err = m.note_error(m.decode_info(info.cpu().numpy())[1])
            if audit is not None:
                err = max(err, m.audit_result(audit))
            picks = []
            st = status.cpu().numpy()
            if m.eps_violated(err):
                m.widen_eps(err)
                todo = m.recertify_subset(idx, win, st, eps_used, fr)       # only the heatmaps whose guard band is not empty
                if todo is None:
                    continue                                              # eps grew past the guard factor: the whole call again
            m.fix_uncertified(idx, win, frames_u8=fr, status=st)
            return idx, win

    def _settle_calls(self, calls, frames, audit=None):
        """Host-side second half of certified forward calls that were only ENQUEUED (the overlapped clip path): `calls` = list of
        dicts {f0, f1 (frame range of the call's input), idx, win, status (flags, host or device), info, eps}.  Folds the calls'
        candidate errors and the audit into eps, re-certifies the heatmaps certified under a stale eps, repairs over-budget heatmaps
        on the fp32 twin.  Returns the set of call positions whose idx / win changed (idx / win are updated in place, or replaced in
        the dict when a whole call was run again)."""
        m = self.model
        changed = set()
        if not m.certified or not calls:
            return changed
        err = m.note_error(max(m.decode_info(np.asarray(c['info'].cpu() if torch.is_tensor(c['info']) else c['info']))[1] for c in calls))
        if audit is not None:
            err = max(err, m.audit_result(audit))
        if m.eps_violated(err):
            m.widen_eps(err)
        for k, c in enumerate(calls):
            st = c['status'].cpu().numpy() if torch.is_tensor(c['status']) else np.array(c['status'])
            fr = frames[c['f0']:c['f1']]
            if c['eps'] < m.eps:
                # an audit found eps too small: the heatmaps of this call whose guard band is not empty are run again under the
                # widened eps; the whole call only when eps grew past the guard factor (audited, blocking; rare)
                todo = m.recertify_subset(c['idx'], c['win'], st, c['eps'], fr)
                if todo is None:
                    c['idx'], c['win'] = self._certified_peaks(fr)
                    changed.add(k)
                    continue
                if todo.size:
                    changed.add(k)          # (st[todo] holds the re-runs' own status now: settled, 0 or 1)
            if ((st & 3) == 2).any():
                # rare: crop budget exceeded -> those samples on the full-frame fp32 path (with the call's own status)
                m.fix_uncertified(c['idx'], c['win'], frames_u8=fr, status=st)
                changed.add(k)
        return changed


class BallDetector(_CertifiedDetector):
    def __init__(self, model_name='segformerpp_b2', max_batch=32, dtype='bf16', lanes=0):
        if 'segformerpp' in model_name or model_name == 'vitpose':
            raise NotImplementedError("detector '%s' depends on code that is not vendored in the reference "
                                      "(KieDani/SegformerPlusPlus / mmcv); only 'wasb' is built" % model_name)
        _lib.require_gpu()
        self.device = torch.device('cuda')
        self.resolution = (WIDTH, HEIGHT)
        sd, res, in_frames = _load_ball_checkpoint(model_name)
        self.model = wasb.get_model(model_name, in_frames=in_frames, resolution=res, pretraining=False, state_dict=sd,
                                    max_batch=max_batch, dtype=dtype, lanes=lanes)
        self.model_resolution = res
        self.max_batch = max_batch

    def predict(self, images):
        """images: list (length B) of [prev, curr, next] BGR uint8 HWC arrays.
        Returns (pred_pos (B,3) float64 [x, y, confidence] in 1920x1080 px, preds (B,1,H,W) float32)."""
        pred_pos, preds = [], []
        w, h = self.model_resolution
        for b0 in range(0, len(images), self.max_batch):
            chunk = images[b0:b0 + self.max_batch]
            xs = []
            for imgs in chunk:
                fr = torch.from_numpy(np.stack([np.asarray(i) for i in imgs])).to(self.device)   # (3,h,w,3) uint8
                xs.append(wasb.preprocess_triples(fr, (w, h)))
            # peaks from the certified argmax (the fp32 index the reference's torch.argmax returns), table-variant fit (interface.py:116)
            heat, idx, win = self._certified_forward(torch.cat(xs))
            pos = refine.refine_windows_device(idx, win, h, w, self.resolution[0], self.resolution[1], _lib.REFINE_TABLE)
            pred_pos.append(pos.cpu().numpy())
            preds.append(heat.cpu().numpy())
        if not pred_pos:
            return np.zeros((0, 3)), np.zeros((0, 1, h, w), np.float32)
        return np.concatenate(pred_pos, axis=0), np.concatenate(preds, axis=0)

    def predict_clip(self, images):
        """Fast path for consecutive frames (what TableTennisPipeline.predict feeds the detector, interface.py:276-279):
        images = list of N BGR uint8 HWC frames -> pred_pos (N-2, 3), the same values `predict` returns for the triples
        (images[i-1], images[i], images[i+1]).  Every frame is uploaded once and pre-processing, CNN, argmax and window
        extraction run fused on the device; no heatmap is written."""
        n = len(images)
        if n < 3:
            return np.zeros((0, 3))
        w, h = self.model_resolution
        out = []
        step = self.max_batch                      # triples per call; consecutive calls overlap by two frames
        for t0 in range(0, n - 2, step):
            fr = torch.from_numpy(np.stack([np.asarray(i) for i in images[t0:t0 + step + 2]])).to(self.device)
            self._calibrate(frames=fr)
            idx, win = self._certified_peaks(fr)
            out.append(refine.refine_windows_device(idx, win, h, w, self.resolution[0], self.resolution[1], _lib.REFINE_TABLE).cpu().numpy())
        return np.concatenate(out, axis=0)

    def filter_trajectory(self, ball_positions, ball_positions_aux, fps):
        return glue.filter_trajectory_ball(ball_positions, ball_positions_aux, fps)


def _load_table_checkpoint(model_name):
    """-> (state_dict, resolution (W,H)).  Reference: inference_tabledetection.load_model :40-57."""
    path = os.path.join(_weights_dir(), 'inference_tabledetection', model_name, 'model.pt')
    if _weights_dir() and os.path.exists(path):
        sd, info = weights.load_checkpoint_state_dict(path)
        return sd, tuple(info.get('image_resolution', (1280, 704)))
    _synthetic_or_raise("TableDetector('%s')" % model_name, path if _weights_dir() else '')
    # seeded stand-in: a planted path to every keypoint head, so the heatmaps are PEAKED like a trained detector's (one dominant
    # maximum per keypoint map; on pure noise weights every map is a field of near-ties and the certified argmax degrades to the
    # full-frame fp32 path -- TTUP_TABLE_NOISE_WEIGHTS=1 selects that regime)
    noise = os.environ.get('TTUP_TABLE_NOISE_WEIGHTS') == '1'
    return weights.random_wasb_state_dict(int(os.environ.get('TTUP_SEED', '0')) + 1, planted=not noise, in_ch=3, head_out=13, plant_all_heads=not noise), (1280, 704)


class TableDetector(_CertifiedDetector):
    NO_CERTIFY_ENV = ('TTUP_NO_CERTIFY', 'TTUP_NO_TABLE_CERTIFY')

    def __init__(self, model_name='segformerpp_b2', max_batch=8, dtype='bf16', lanes=0):
        if 'segformerpp' in model_name or model_name == 'vitpose':
            raise NotImplementedError("detector '%s' depends on code that is not vendored in the reference; only 'hrnet' is built" % model_name)
        _lib.require_gpu()
        self.device = torch.device('cuda')
        self.resolution = (WIDTH, HEIGHT)
        self.KEYPOINT_VISIBLE = KEYPOINT_VISIBLE
        sd, res = _load_table_checkpoint(model_name)
        self.model = wasb.get_table_model(model_name, resolution=res, pretraining=False, state_dict=sd, max_batch=max_batch, dtype=dtype, lanes=lanes)
        self.model_resolution = res
        self.max_batch = max_batch

    def predict(self, images):
        """images: list (length B) of BGR uint8 HWC frames.
        Returns (pred_pos (B,13,3) float64 [x, y, visibility] in 1920x1080 px, preds (B,1,13,H,W) float32 -- the reference
        stacks one (1,13,H,W) tensor per frame with np.array, interface.py:165-167)."""
        pred_pos, preds = [], []
        w, h = self.model_resolution
        for b0 in range(0, len(images), self.max_batch):
            fr = torch.from_numpy(np.stack([np.asarray(i) for i in images[b0:b0 + self.max_batch]])).to(self.device)
            # per-channel peaks from the certified argmax: the reference takes them from fp32 heatmaps (interface.py:148-172 ->
            # tabledetection/helper_tabledetection.py:50-156)
            heat, idx, win = self._certified_forward(wasb.preprocess_frames(fr, (w, h)))
            pos = refine.refine_windows_device(idx, win, h, w, self.resolution[0], self.resolution[1], _lib.REFINE_TABLE)
            pred_pos.append(pos.cpu().numpy().reshape(-1, 13, 3))
            preds.append(heat.cpu().numpy()[:, None])
        if not pred_pos:
            return np.zeros((0, 13, 3)), np.zeros((0, 1, 13, h, w), np.float32)
        return np.concatenate(pred_pos, axis=0), np.concatenate(preds, axis=0)

    def predict_keypoints(self, images):
        """`predict` without the heatmaps: (B,13,3) keypoints only.  Pre-processing, CNN, per-channel argmax and windows run
        fused on the device; nothing but the 39 numbers per frame comes back to the host."""
        w, h = self.model_resolution
        out = []
        for b0 in range(0, len(images), self.max_batch):
            fr = torch.from_numpy(np.stack([np.asarray(i) for i in images[b0:b0 + self.max_batch]])).to(self.device)
            self._calibrate(frames=fr)
            idx, win = self._certified_peaks(fr)
            pos = refine.refine_windows_device(idx.reshape(-1), win.reshape(-1, 9), h, w, self.resolution[0], self.resolution[1], _lib.REFINE_TABLE)
            out.append(pos.cpu().numpy().reshape(-1, 13, 3))
        return np.concatenate(out, axis=0) if out else np.zeros((0, 13, 3))

    def calibrate_camera(self, keypoints):
        """interface.py:174-175: (13,3) keypoints -> (Mint, Mext); host numpy/SciPy like the reference (calib.py)."""
        return calib.calibrate_camera(keypoints)

    def filter_trajectory(self, table_keypoints, table_keypoints_aux):
        return glue.filter_trajectory_table(table_keypoints, table_keypoints_aux)


class UpliftingModel:
    def __init__(self, max_len=128):
        _lib.require_gpu()
        self.device = torch.device('cuda')
        sd, size, self.transform_mode = _load_uplift_checkpoint()
        self.model = uplift.get_model('connectstage', size, 'dynamic', 'new', state_dict=sd, max_batch=64, max_len=max_len)

    def transform(self, data):
        """NormalizeImgCoords (uplifting/transformations.py:252-266): divide by the uplift resolution 2560x1440."""
        r_img, table_img = data['r_img'], data['table_img']
        r_img = r_img / np.array([2560, 1440])
        table_img[..., :2] = table_img[..., :2] / np.array([2560, 1440])
        data['r_img'], data['table_img'] = r_img, table_img
        return data

    def predict(self, ball_coords, table_coords, times):
        data = self.transform({'r_img': ball_coords, 'table_img': table_coords})
        ball_coords, table_coords = data['r_img'], data['table_img']
        mask = np.zeros((ball_coords.shape[0] + 1,), dtype=np.float32)
        mask[:-1] = 1.0
        return self.predict_without_normalization(ball_coords, table_coords, torch.tensor(mask).to(self.device), times)

    def predict_without_normalization(self, ball_coords, table_coords, mask, times):
        host = all(isinstance(a, np.ndarray) or (torch.is_tensor(a) and a.device.type == 'cpu') for a in (ball_coords, table_coords, mask, times))
        if host and np.asarray(ball_coords).ndim == 2:
            # one rally from host arrays (what the pipeline hands over): padded on the host, ONE upload, the number of valid steps from the
            # host mask -- instead of four uploads, four pad kernels and a device-side mask.sum().item() (three host round trips less)
            b_, t_, m_, tm_ = [np.asarray(a, dtype=np.float32) for a in (ball_coords, table_coords, mask, times)]
            n = m_.shape[-1]
            buf = torch.zeros((4 * n + 39,), dtype=torch.float32).pin_memory()
            hb = buf.numpy()
            hb[:2 * b_.shape[0]] = b_.reshape(-1); hb[2 * n:2 * n + 39] = t_.reshape(-1)
            hb[2 * n + 39:3 * n + 39] = m_.reshape(-1); hb[3 * n + 39:3 * n + 39 + tm_.shape[0]] = tm_.reshape(-1)
            # the reference's mask check (uplifting/model.py:541-546) on the host copy: no device round trip for it
            if not (m_.size and float(m_.min()) == 0.0 and float(m_.max()) == 1.0):
                raise ValueError('wrong format for masks. Should be 0, 1 or -1e9, 0.')
            dev = buf.to(self.device, non_blocking=True)
            pred_rotation, pred_position = self.model(dev[:2 * n].view(1, n, 2), dev[2 * n:2 * n + 39].view(1, 13, 3), dev[2 * n + 39:3 * n + 39].view(1, n),
                                                      dev[3 * n + 39:].view(1, n), check_mask=False)
            pred_rotation_local = uplift.transform_rotationaxes(pred_rotation, pred_position.clone()) if self.transform_mode == 'global' else pred_rotation
            t_prime = int(m_.sum())
            return pred_rotation_local.squeeze(0), pred_position[0, :t_prime, :].cpu().numpy()
        ball_coords, table_coords, mask, times = [torch.as_tensor(a).to(self.device, torch.float32) for a in (ball_coords, table_coords, mask, times)]
        if ball_coords.dim() == 2:      # (N,2) -> pad to the mask length like the reference's callers do
            n = mask.shape[-1]
            b = torch.zeros((1, n, 2), device=self.device); b[0, :ball_coords.shape[0]] = ball_coords
            t = torch.zeros((1, n), device=self.device); t[0, :times.shape[0]] = times
            ball_coords, times, mask, table_coords = b, t, mask.reshape(1, n), table_coords.reshape(1, 13, 3)
        pred_rotation, pred_position = self.model(ball_coords, table_coords, mask, times)
        if self.transform_mode == 'global':
            pred_rotation_local = uplift.transform_rotationaxes(pred_rotation, pred_position.clone())
        else:
            pred_rotation_local = pred_rotation
        t_prime = int(mask.sum().item())
        pred_position = pred_position[:, :t_prime, :].cpu().numpy()
        return pred_rotation_local.squeeze(0), pred_position.squeeze(0)


class TableTennisPipeline:
    def __init__(self, max_batch=32):
        _lib.require_gpu()
        self.device = torch.device('cuda')
        self.CHUNK = int(os.environ.get('TTUP_HUB_CHUNK', self.CHUNK))
        self.FIRST = int(os.environ.get('TTUP_HUB_FIRST', self.FIRST))
        self.CHUNK_LONG = max(self.CHUNK, int(os.environ.get('TTUP_HUB_CHUNK_LONG', self.CHUNK_LONG)))
        # one lane per detector: the two handles already run side by side on their own streams; with two lanes each, four CNN streams
        # (plus copy, audit and refine work) collide on the runtime's four hardware queues, and a high-priority table kernel queued
        # behind a ball kernel is no longer ahead of it (measured on a 48-frame clip: 748-757 -> 804-820 frames/s; table lanes alone:
        # 808-811; GPU_MAX_HW_QUEUES=8 with one lane each: 843)
        lanes = int(os.environ.get('TTUP_HUB_LANES', '1'))
        self.ball_detector = BallDetector(model_name='wasb', max_batch=max(max_batch, self.CHUNK_LONG), lanes=lanes)
        self.ball_detector_aux = self.ball_detector       # the primary SegFormer++ detector is not available offline
        self.table_detector = TableDetector(model_name='hrnet', max_batch=max(16, self.CHUNK_LONG), lanes=lanes)
        self.table_detector_aux = self.table_detector
        # the overlapped clip path runs both detectors side by side: the table detector goes first on the GPU, so that its
        # host-side consumer (the DBSCAN keypoint filter) overlaps with the rest of the ball detector
        self.table_detector.model.set_priority(True)
        self.uplifting_model = UpliftingModel()
        self.KEYPOINT_VISIBLE = KEYPOINT_VISIBLE

    def predict(self, images, fps):
        """images: list of BGR frames of one rally; fps: frame rate (interface.py:265-289).
        Returns (pred_spin torch (3,), pred_pos_3d numpy (T',3))."""
        return self._predict(images, fps, None)

    def predict_with_table(self, images, fps, table_keypoints):
        """`predict` with known table keypoints ((13,3) [x,y,vis] in 1920x1080 px), skipping table detection -- an addition
        for fixed-camera streams; the reference surface is `predict`."""
        return self._predict(images, fps, table_keypoints)

    CHUNK = 24          # frames per upload / detector call of the overlapped clip path (measured: 16 -> 68 ms, 24 -> 60 ms, 48 -> 63 ms per 48-frame clip)
    CHUNK_LONG = 64     # ... of clips of at least four chunks, after their first chunk: a detector call drains at its end, so long clips take fewer, larger calls (256 frames: 868 -> see DESIGN.md 11)
    FIRST = 24          # frames of the first chunk (a short first chunk -- 8 frames -- was measured 5 ms SLOWER per clip: its one-micro-batch calls run at half the batched rate)

    def _clip_detections(self, images, want_table, table_consumer=None):
        """Ball positions (N-2,3) and table keypoints ((N,13,3), or what `table_consumer` makes of them) of one clip with everything overlapped: the frames are staged in
        pinned memory and uploaded ONCE in chunks on a copy stream; while chunk k+1 is staged and copied, the table detector
        runs on chunk k on its own stream and the ball detector on the triples whose three frames are already resident on a
        third; the host blocks only at the end.  Same values as `predict_clip` / `predict_keypoints` (same kernels per frame)."""
        n, dev = len(images), self.device
        import time as _time
        tr = self.__dict__.get('_trace')          # tools/hub_trace.py: host time stamps (ms since the call) of the clip path's stages
        t00 = _time.perf_counter()
        mark = (lambda name: tr.append((name, (_time.perf_counter() - t00) * 1e3))) if tr is not None else (lambda name: None)
        C = self.CHUNK if n < 4 * self.CHUNK else self.CHUNK_LONG
        CP = self.CHUNK_LONG                       # rows of the pinned staging buffers
        h0, w0 = np.asarray(images[0]).shape[:2]
        bd, td = self.ball_detector, self.table_detector
        bw, bh = bd.model_resolution
        tw, th = td.model_resolution
        frames = torch.empty((n, h0, w0, 3), dtype=torch.uint8, device=dev)
        st = self.__dict__.setdefault('_streams', None)
        if st is None:
            st = self._streams = {k: torch.cuda.Stream(dev) for k in ('ball', 'table')}
            st['copy'] = st['table']          # uploads ride on the table stream (chunk k+1 behind the table pass of chunk k): one stream fewer
            self._pinned = [torch.empty((CP, h0, w0, 3), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._pin_free = [None, None]
        if self._pinned[0].shape[1:] != (h0, w0, 3):
            self._pinned = [torch.empty((CP, h0, w0, 3), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._pin_free = [None, None]
        cur = torch.cuda.current_stream(dev)
        for s in st.values():
            s.wait_stream(cur)
        # chunk schedule: a short first chunk (FIRST frames) gets the GPU going while the host still stages the bulk of the clip;
        # after it, chunks of C frames.  Each chunk = one upload, one table-detector call and one ball-detector call on the triples
        # whose three frames are resident by then.
        F0 = min(self.FIRST, C, n)
        bounds = [0, F0] + list(range(F0 + C, n, C)) + ([n] if n > F0 else [])
        bounds = sorted(set(bounds))
        ball_out, table_out, ball_calls, table_calls = [], [], [], []
        t_next = 0                        # first triple not yet submitted
        ev = None
        for ci, (c0, c1) in enumerate(zip(bounds[:-1], bounds[1:])):
            pin = self._pinned[ci % 2]
            if self._pin_free[ci % 2] is not None:
                self._pin_free[ci % 2].synchronize()          # the copy that last read this staging buffer is done
            self._stage(images, c0, c1, pin)
            mark('staged chunk %d' % ci)
            with torch.cuda.stream(st['copy']):
                frames[c0:c1].copy_(pin[:c1 - c0], non_blocking=True)
                ev = torch.cuda.Event(); ev.record()
            self._pin_free[ci % 2] = ev
            if ci == 0:
                torch.cuda.current_stream(dev).wait_event(ev)
                bd._calibrate(frames=frames[:c1]) if c1 >= 3 else None        # certified argmax: once per detector
                if want_table:
                    td._calibrate(frames=frames[:c1])
            if want_table:
                with torch.cuda.stream(st['table']):
                    st['table'].wait_event(ev)
                    tm = td.model
                    _, idx, win = tm.forward_frames(frames[c0:c1], want_heatmap=False)
                    # status / info of THIS call, copied right behind it (per-call slot); the keypoints are refined at once from what
                    # the call returned -- settled below, after the stream has drained, and refined again only where a repair changed them
                    table_calls.append({'f0': c0, 'f1': c1, 'idx': idx, 'win': win, 'eps': tm.eps if tm.certified else None,
                                        'status': tm.certify_status(idx.shape[0], raw=True) if tm.certified else None,
                                        'info': tm.certify_info() if tm.certified else None})
                    table_out.append(refine.refine_windows_device(idx.reshape(-1), win.reshape(-1, 9), th, tw, td.resolution[0], td.resolution[1], _lib.REFINE_TABLE))
            # triples t need frames t..t+2: everything up to c1-3 can go now
            while t_next < c1 - 2:
                nt = min(bd.max_batch, c1 - 2 - t_next)
                with torch.cuda.stream(st['ball']):
                    st['ball'].wait_event(ev)
                    fr = frames[t_next:t_next + nt + 2]
                    _, idx, win = bd.model.forward_frames(fr, want_heatmap=False)
                    # status / info of THIS call, copied right behind it (the handle's per-call slot flips with the next call)
                    status = bd.model.certify_status(nt, raw=True) if bd.model.certified else None
                    info = bd.model.certify_info() if bd.model.certified else None
                    ball_calls.append((t_next, nt, idx, win, status, info, bd.model.eps if bd.model.certified else None))
                t_next += nt
        uploaded = [ev]
        mark('all calls enqueued')
        frames.record_stream(st['ball']); frames.record_stream(st['table'])
        # eps audit of the certified argmax: a random triple of the clip on the fp32 twin, on its own stream next to the detectors
        audit = None
        picks = bd._audit_picks(n - 2)
        if picks:
            cur.wait_event(uploaded[-1])
            audit = bd.model.audit_async(frames, picks)
        kp = None
        if want_table:
            # the table detector (high-priority streams) finishes first: its keypoints come back and the host-side DBSCAN filter
            # runs while the ball detector is still busy on the GPU
            t_audit = None
            t_picks = td._audit_picks(n)
            if t_picks:
                cur.wait_event(uploaded[-1])
                t_audit = td.model.audit_async(frames, t_picks)
            cert = td.model.certified and bool(table_calls)
            with torch.cuda.stream(st['table']):
                kp_dev = torch.cat(table_out).reshape(-1, 13, 3)
                kp_host = torch.empty(kp_dev.shape, dtype=kp_dev.dtype, pin_memory=True)
                kp_host.copy_(kp_dev, non_blocking=True)
                if cert:          # the calls' status flags and crop / error info in one copy each
                    st_dev = torch.cat([c['status'] for c in table_calls])
                    in_dev = torch.stack([c['info'] for c in table_calls])
                    st_host = torch.empty(st_dev.shape, dtype=st_dev.dtype, pin_memory=True); st_host.copy_(st_dev, non_blocking=True)
                    in_host = torch.empty(in_dev.shape, dtype=in_dev.dtype, pin_memory=True); in_host.copy_(in_dev, non_blocking=True)
                ev_t = torch.cuda.Event(); ev_t.record()
            ev_t.synchronize()
            mark('table stream drained')
            kp_np = kp_host.numpy()
            if cert:
                o = 0
                for k, c in enumerate(table_calls):
                    nmap = c['idx'].shape[0]
                    c['status'], c['info'] = st_host.numpy()[o:o + nmap], in_host.numpy()[k]
                    o += nmap
                cur.wait_stream(st['table'])
                for k in sorted(td._settle_calls(table_calls, frames, t_audit)):          # rare: re-certified / repaired calls are refined again
                    c = table_calls[k]
                    pos = refine.refine_windows_device(c['idx'].reshape(-1), c['win'].reshape(-1, 9), th, tw, td.resolution[0], td.resolution[1], _lib.REFINE_TABLE)
                    kp_np[c['f0']:c['f1']] = pos.cpu().numpy().reshape(-1, 13, 3)
            mark('table calls settled')
            kp = table_consumer(kp_np) if table_consumer is not None else kp_np.copy()
            mark('keypoint filter done')
        for s in st.values():
            cur.wait_stream(s)
        calls = [{'f0': t0, 'f1': t0 + nt + 2, 'idx': idx, 'win': win, 'status': status, 'info': info, 'eps': eps_used}
                 for (t0, nt, idx, win, status, info, eps_used) in ball_calls]
        bd._settle_calls(calls, frames, audit)
        for c in calls:
            ball_out.append(refine.refine_windows_device(c['idx'], c['win'], bh, bw, bd.resolution[0], bd.resolution[1], _lib.REFINE_TABLE))
        pos = torch.cat(ball_out).cpu().numpy() if ball_out else np.zeros((0, 3))
        mark('ball calls settled, positions on the host')
        return pos, kp

    STAGE_THREADS = 4

    def _stage(self, images, c0, c1, pin):
        """Copy frames c0..c1 of the caller's list into the pinned staging buffer.  The copies are plain memcpys of 2.8 MB each
        (numpy releases the GIL for them), so a few threads bring a 16-frame chunk from 4.5 ms down to about 1.5 ms -- the first
        chunk's staging is the one stretch of a clip during which the GPU has nothing to do."""
        dst = pin.numpy()
        nthr = 1 if os.environ.get('TTUP_HUB_STAGE_THREADS') == '1' else self.STAGE_THREADS
        if nthr <= 1 or c1 - c0 < 4:
            for k in range(c0, c1):
                np.copyto(dst[k - c0], images[k])
            return
        pool = self.__dict__.get('_stage_pool')
        if pool is None:
            import concurrent.futures
            pool = self._stage_pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.STAGE_THREADS)
        list(pool.map(lambda k: np.copyto(dst[k - c0], images[k]), range(c0, c1)))

    def _predict(self, images, fps, table_keypoints):
        overlapped = (self.ball_detector_aux is self.ball_detector and self.table_detector_aux is self.table_detector and len(images) >= 3
                      and self.table_detector.max_batch >= self.CHUNK_LONG and self.ball_detector.max_batch >= self.CHUNK_LONG
                      and os.environ.get('TTUP_HUB_SERIAL') != '1')
        if overlapped:
            ball_positions, kp = self._clip_detections(images, want_table=table_keypoints is None,
                                                       table_consumer=lambda k: self.table_detector_aux.filter_trajectory(k, k))
            ball_positions_aux = ball_positions
            if table_keypoints is None:
                table_keypoints = kp
        else:
            if table_keypoints is None:        # 2. table detection (interface.py:281-283)
                kp = self.table_detector.predict_keypoints(images)
                kp_aux = kp if self.table_detector_aux is self.table_detector else self.table_detector_aux.predict_keypoints(images)
                table_keypoints = self.table_detector_aux.filter_trajectory(kp, kp_aux)
            # the reference builds (prev, curr, next) triples and pushes each through the detector (interface.py:276-279);
            # the triples are consecutive frames, so the clip path computes the same positions with every frame uploaded once
            ball_positions = self.ball_detector.predict_clip(images)
            ball_positions_aux = ball_positions if self.ball_detector_aux is self.ball_detector else self.ball_detector_aux.predict_clip(images)
        filtered, _, times_ball = self.ball_detector.filter_trajectory(ball_positions, ball_positions_aux, fps)
        ball_coords, table_coords, times, mask = glue._uplifting_transform(filtered, np.asarray(table_keypoints, dtype=np.float64), times_ball)
        return self.uplifting_model.predict_without_normalization(ball_coords, table_coords, mask, times)

    def calibrate_camera(self, keypoints):
        """interface.py:291-299: (13,3) table keypoints [x, y, visibility] in pixels -> Mint (3,4), Mext (4,4)."""
        return calib.calibrate_camera(keypoints)

    def reproject(self, positions_3d, Mint, Mext):
        """interface.py:301-312: (N,3) world positions -> (N,2) pixel positions."""
        return calib.reproject(positions_3d, Mint, Mext)
