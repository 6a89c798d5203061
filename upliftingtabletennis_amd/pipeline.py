"""Sharded detect -> refine -> uplift worker (one process per GPU) and the single collective of the path.

The reference is single-process / single-GPU.  The path shards over independent units:
video streams (or frame ranges of one stream with a 1-frame halo, because triple t needs frames t..t+2) and
trajectories.  Weights are replicated; there is no data-path collective.  The only exchange is the final gather
of small fixed-size records -- per frame (x, y, visibility) float64 and per trajectory (spin[3], T', pos[T,3])
float32 -- a few KB per stream, latency-bound on any xGMI topology (``gather_records``).
"""
import numpy as np
import torch

TRAJ_LEN_DEFAULT = 32


def shard_range(n_units, world, rank):
    """Contiguous balanced partition of range(n_units): the first n_units % world ranks get one extra unit."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError('bad world/rank %d/%d' % (world, rank))
    q, r = divmod(n_units, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def frame_range_with_halo(n_frames, world, rank):
    """Split ONE stream of n_frames (n_frames-2 triples) over ranks.  Returns (first_frame, last_frame_exclusive,
    first_triple, n_triples): each rank reads its triples' frames plus the 2-frame look-ahead."""
    t0, t1 = shard_range(max(n_frames - 2, 0), world, rank)
    if t1 <= t0:
        return 0, 0, t0, 0
    return t0, t1 + 2, t0, t1 - t0


def _pack_records(records, spec, device):
    """One byte buffer per rank: int64 row counts of every key (sorted), then each key's rows in a region of fixed capacity."""
    keys = sorted(spec)
    cap = {k: int(spec[k][0]) * int(np.prod(spec[k][1], dtype=np.int64)) * torch.empty((), dtype=spec[k][2]).element_size() for k in keys}
    total = 8 * len(keys) + sum(cap.values())
    buf = torch.zeros((total,), dtype=torch.uint8, device=device)
    rows = []
    off = 8 * len(keys)
    for k in keys:
        v = records[k].contiguous()
        if v.dtype != spec[k][2] or tuple(v.shape[1:]) != tuple(spec[k][1]) or v.shape[0] > spec[k][0]:
            raise ValueError('record %r %s/%s does not fit its spec %s' % (k, tuple(v.shape), v.dtype, spec[k]))
        rows.append(v.shape[0])
        nb = v.numel() * v.element_size()
        if nb:
            buf[off:off + nb] = v.to(device).reshape(-1).view(torch.uint8)
        off += cap[k]
    buf[:8 * len(keys)] = torch.tensor(rows, dtype=torch.int64).view(torch.uint8).to(device)
    return buf, keys, cap


def _unpack_records(flat, world, keys, cap, spec):
    total = flat.numel() // world
    out = {k: [] for k in keys}
    for r in range(world):
        b = flat[r * total:(r + 1) * total]
        rows = b[:8 * len(keys)].clone().view(torch.int64).tolist()
        off = 8 * len(keys)
        for k, n in zip(keys, rows):
            shape, dt = tuple(spec[k][1]), spec[k][2]
            nb = n * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
            out[k].append(b[off:off + nb].clone().view(dt).reshape((n,) + shape))
            off += cap[k]
    return out


def gather_records(records, dist=None, dst=0, spec=None):
    """The one exchange of the path: gather a dict of per-rank records -- tensors whose first dimension may differ per
    rank -- on rank `dst`.  Returns {key: [tensor_of_rank0, tensor_of_rank1, ...]} on dst, None elsewhere; with dist=None (single
    process) it just wraps the local records.

    `spec` = {key: (max_rows, row_shape, dtype)} (StreamWorker.record_spec()) fixes every key's capacity, so a step is exactly ONE
    collective: all ranks contribute one equally sized byte buffer (row counts + rows) to an all_gather.  Without a spec the
    capacities are agreed first (one extra all_reduce of the row counts): two collectives.  nccl (= RCCL) gathers device buffers,
    gloo (CPU tests, single-GPU dry runs) host buffers.  With world > 1 the tensors returned on `dst` are HOST tensors (the
    gathered payload is a few KB of results that the caller reads on the host); with one process the local records are returned as
    they are."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: [v] for k, v in records.items()}
    world, rank = dist.get_world_size(), dist.get_rank()
    on_host = dist.get_backend() == 'gloo'
    any_v = next(iter(records.values()))
    device = torch.device('cpu') if on_host else any_v.device
    if spec is None:
        keys = sorted(records)
        rows = torch.tensor([records[k].shape[0] for k in keys], dtype=torch.int64, device=device)
        dist.all_reduce(rows, op=dist.ReduceOp.MAX)
        spec = {k: (max(int(n), 1), tuple(records[k].shape[1:]), records[k].dtype) for k, n in zip(keys, rows.tolist())}
    buf, keys, cap = _pack_records(records, spec, device)
    flat = torch.empty((world * buf.numel(),), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(flat, buf)
    if rank != dst:
        return None
    return _unpack_records(flat.cpu(), world, keys, cap, spec)


class StreamWorker:
    """Per-GPU worker: owns one CNN handle, one uplift handle, and runs whole clips through the path.
    Raises RuntimeError without a HIP device (no CPU fallback).

    Certified argmax (`certify=True`, bf16): eps -- the bound on |bf16 heatmap - fp32 heatmap| the certification rests on -- is
    estimated on the first clip and then AUDITED while the worker runs: one random triple per `audit_every` triples (per
    `audit_every_fast` until eps has stood for `audit_settle_clips` clips in a row) is re-run on
    the fp32 twin on a side stream, and every fp32 crop the certification computes anyway reports the error at its candidates.
    eps is 1.5 times the largest error seen so far (calibration frames, audits); a new maximum widens it, and of the clips certified
    under the old value only the heatmaps whose guard band (pixels between 2 eps and 2.5 eps below the maximum) is not empty are
    run again -- the whole clip only when eps grows by more than a quarter at once.  `audit` reports the counts; `audit_every=0` switches the side-stream audit off."""

    def __init__(self, device, wasb_state_dict, uplift_state_dict, net_wh=(1280, 704), max_triples=256, uplift_size='large',
                 traj_len=TRAJ_LEN_DEFAULT, seq_len=50, dtype='bf16', certify=True, audit_every=256, audit_seed=0, exact_windows=False,
                 audit_every_fast=64, audit_settle_clips=8, audit_crops_every=16):
        from . import glue, refine, uplift, wasb, _lib
        _lib.require_gpu()
        self._glue, self._refine, self._uplift, self._lib = glue, refine, uplift, _lib
        self.device = torch.device(device)
        self.net_w, self.net_h = net_wh
        self.traj_len, self.seq_len = traj_len, seq_len
        self.max_triples = max_triples
        self.net = wasb.WASBNet(wasb_state_dict, resolution=net_wh, max_batch=max_triples, dtype=dtype, device=self.device)
        self.max_segments = max(64, (max_triples + traj_len - 1) // traj_len)
        self.up = uplift.get_model('connectstage', uplift_size, 'dynamic', 'new', state_dict=uplift_state_dict,
                                   max_batch=self.max_segments, max_len=seq_len, device=self.device)
        # certified argmax (the fp32 path's indices from the bf16 path, csrc/certify.hip): calibrated on the first clip seen
        self.certify = bool(certify) and dtype == 'bf16'
        self.exact_windows = bool(exact_windows)
        self.certify_eps = None
        self.fp32_reruns = 0
        self.recertified_clips = 0
        self.recertified_heatmaps = 0
        # Side-stream audit rate (triples per audited triple), ADAPTIVE: `audit_every_fast` while the bound is still moving -- until
        # `audit_settle_clips` consecutive clips have passed without a widening of eps -- and `audit_every` afterwards; a widening
        # drops back to the fast rate.  New content is where a too-small eps is found (the soaks widen within the first clips of a
        # content change and not again), so the audits are spent there.  `audit['audited_share']` = audited / processed triples.
        self.audit_every = int(audit_every)
        self.audit_every_fast = min(int(audit_every_fast), self.audit_every) if int(audit_every_fast) > 0 else self.audit_every
        self.audit_settle_clips = int(audit_settle_clips)
        self._quiet_clips = 0
        self.frames_seen = 0
        # Audit crops (round 6): one single-candidate heatmap per `audit_crops_every` triples gets an fp32 crop although its index is
        # already certain; the crop reports |bf16 - fp32| at the winner.  The strip audit sees every pixel of a quarter-width strip of
        # one triple per 64-256; this sees ONE pixel -- the one the detection rests on -- of one triple per 16, at 0.1 ms each
        # (1.3 % of a step).  The phase is drawn per clip.  0 = off.
        self.audit_crops_every = int(audit_crops_every)
        self.audit_crop_frames = 0
        self.widen_sources = {'strip': 0, 'candidates': 0}
        self._since_audit = 0
        self._rng = np.random.default_rng(audit_seed)
        self._rng_crops = np.random.default_rng(audit_seed + 1)          # (its own stream: tests replace _rng to steer the strip audit)
        # measurement (bench.py `ambiguous_share`): when set to a list, every collected clip appends the fp32 top-2 margins of its
        # heatmaps (`WASBNet.certify_margins`: +inf for single-candidate heatmaps)
        self.margin_log = None

    def record_spec(self):
        """Capacities of the per-clip records (`gather_records(..., spec=...)`: one collective per step)."""
        return {'xyv': (self.max_triples, (3,), torch.float64), 'spin': (self.max_segments, (3,), torch.float32),
                'pos3d': (self.max_segments, (self.seq_len, 3), torch.float32), 'n_valid': (self.max_segments, (), torch.int64)}

    @property
    def audit(self):
        """{'audited_frames', 'max_err_seen', 'widened', 'eps', 'max_err_over_eps', 'recertified_clips'} of the certified argmax."""
        a = dict(getattr(self.net, 'audit_state', None) or dict(audited_frames=0, max_err_seen=0.0, widened=0))
        a['eps'] = self.certify_eps
        a['max_err_over_eps'] = (a['max_err_seen'] / self.certify_eps) if self.certify_eps else None
        a['recertified_clips'] = self.recertified_clips
        a['recertified_heatmaps'] = self.recertified_heatmaps
        # what the side-stream audit has covered: audited triples (calibration frames included) / triples processed.  A frame whose
        # error exceeds eps while no audited frame's does is missed with probability 1 - (the audit rate at that time) by the strip
        # audit (the candidate-level audit still sees it when it needs a crop): the guarantee is statistical and this is its rate
        a['frames_seen'] = self.frames_seen
        # strip audits (every pixel of a strip of the triple, calibration frames included) + audit crops (the winner's pixel of the triple)
        a['audit_crop_frames'] = self.audit_crop_frames
        a['strip_audited_share'] = (a['audited_frames'] / self.frames_seen) if self.frames_seen else None
        a['audited_share'] = ((a['audited_frames'] + self.audit_crop_frames) / self.frames_seen) if self.frames_seen else None
        a['audit_every_now'] = self.audit_rate()
        a['quiet_clips'] = self._quiet_clips
        a['widen_sources'] = dict(self.widen_sources)
        return a

    def audit_rate(self):
        """Triples per audited triple right now (0 = side-stream audit off)."""
        if not self.certify or self.audit_every <= 0:
            return 0
        return self.audit_every_fast if self._quiet_clips < self.audit_settle_clips else self.audit_every

    def detect(self, frames_u8):
        """(N,h,w,3) uint8 on the device -> (N-2,3) float64 [x, y, visibility] in 1920x1080 px (table-variant refine,
        like interface.py:116)."""
        return self._detect(frames_u8)[0]

    def _calibrated(self, frames_u8):
        if self.certify and self.certify_eps is None:
            self.certify_eps = self.net.calibrate(frames_u8, n=8, exact_windows=self.exact_windows)
        return self.certify_eps

    def _start_audit(self, frames_u8, rerun=False):
        """Enqueue the side-stream audit of this clip (if one is due) next to its detector pass; None otherwise.  rerun=True: the clip
        has been counted (and its audit drawn) already -- a re-run after eps widened past the guard factor is the same triples again."""
        if rerun:
            return None
        picks = self._pick_audits(frames_u8.shape[0] - 2)
        return self.net.audit_async(frames_u8, picks) if picks else None

    def _detect(self, frames_u8):
        self._calibrated(frames_u8)
        if self.certify and self.audit_crops_every > 0:
            every = self.audit_crops_every
            phase = int(self._rng_crops.integers(every))
            self.net.certify_audit_crops(every, phase)
            n = int(frames_u8.shape[0]) - 2
            if not self.__dict__.get('_rerun_pass', False):
                self.audit_crop_frames += len(range((-phase) % every, n, every))          # frames f < n with (f + phase) % every == 0
        _, idx, win = self.net.forward_frames(frames_u8, want_heatmap=False)
        xyv = self._refine.refine_windows_device(idx, win, self.net_h, self.net_w, 1920, 1080, self._lib.REFINE_TABLE)
        # status and info belong to THIS call: both are copied right behind it in stream order, before another call can flip the
        # handle's per-call slot
        status = self.net.certify_status(idx.shape[0], raw=True) if self.certify else None          # bit 2 = guard band not empty
        info = self.net.certify_info() if self.certify else None
        self._last_margin = self.net.certify_margins(idx.shape[0]) if self.certify and self.margin_log is not None else None
        return xyv, idx, win, status, info

    def _pick_audits(self, n_triples):
        """Indices of the triples of a clip that the side-stream audit re-runs on the fp32 twin: one per `audit_every` triples."""
        self.frames_seen += max(0, int(n_triples))
        every = self.audit_rate()
        if every <= 0 or n_triples <= 0:
            return []
        self._since_audit += n_triples
        picks = []
        while self._since_audit >= every:
            self._since_audit -= every
            picks.append(int(self._rng.integers(n_triples)))
        return picks

    def _repair(self, frames_u8, xyv, idx, win, status_host):
        """Rare slow path: heatmaps the certified argmax could not settle inside its crop budget are re-run on the full-frame
        fp32 handle, so that every detection comes from the fp32 argmax.  `status_host` is the status of the call that produced
        idx / win (not of whatever the handle ran last)."""
        bad = np.nonzero((status_host & 3) == 2)[0]
        if bad.size == 0:
            return xyv
        self.fp32_reruns += self.net.fix_uncertified(idx, win, frames_u8=frames_u8, status=status_host)
        sel = torch.as_tensor(bad, device=self.device)
        xyv[sel] = self._refine.refine_windows_device(idx[sel], win[sel], self.net_h, self.net_w, 1920, 1080, self._lib.REFINE_TABLE)
        return xyv

    def _after_clip(self, n_crops, cand_err, audit_ticket):
        """Fold a finished clip's evidence into the eps audit and size the crop budget of the clips to come.  Returns True when eps
        had to be widened (clips certified under a smaller eps must be re-run)."""
        net = self.net
        err = net.note_error(cand_err)
        strip_err = net.audit_result(audit_ticket) if audit_ticket is not None else 0.0
        source = 'strip' if strip_err > err else 'candidates'
        err = max(err, strip_err)
        widened = False
        if net.eps_violated(err):
            self.certify_eps = net.widen_eps(err)
            widened = True
            self.__dict__.setdefault('widen_sources', {'strip': 0, 'candidates': 0})[source] += 1
        self._quiet_clips = 0 if widened else self.__dict__.get('_quiet_clips', 0) + 1
        # crop budget of the next clips (it sizes the number of fp32 passes a call provisions; an unused pass still costs its launches):
        # one and a half times the most any of the last eight clips asked for (+ slack) -- content that alternates between easy and
        # hard clips keeps the hard clips' budget (twice the LAST clip's count sent 40 % of a hard clip that followed an easy one to
        # the full-frame fp32 path); clips that outgrow it are flagged and repaired
        hist = self.__dict__.setdefault('_crop_hist', [])
        hist.append(int(n_crops))
        del hist[:-8]
        net.certify_budget(3 * max(hist) // 2 + 16)
        return widened

    def uplift_segments(self, positions, table_px, fps):
        """Cut the detections into rallies of `traj_len` frames, filter / normalise / pad each like the reference
        (inference/utils.py:70-102, :268-309) and run them as one uplift batch.
        Returns (spin_local (S,3), pos3d (S,seq_len,3), n_valid (S,)) on the device."""
        balls, tables, times, masks = [], [], [], []
        for s in range(0, positions.shape[0], self.traj_len):
            seg = positions[s:s + self.traj_len]
            filt, _, t = self._glue.filter_trajectory_ball(seg, seg, fps)
            b, tb, tm, mk = self._glue._uplifting_transform(filt, table_px, t, self.seq_len)
            balls.append(b); tables.append(tb); times.append(tm); masks.append(mk)
        mask = torch.cat(masks)
        rot, p3 = self.up(torch.cat(balls), torch.cat(tables), mask, torch.cat(times))
        spin = self._uplift.transform_rotationaxes(rot, p3)
        return spin, p3, mask.sum(1).to(torch.int64).to(self.device)

    def _recertify(self, frames_u8, xyv, idx, win, status_host, eps_used):
        """The clip was certified under `eps_used` and eps has grown since.  Heatmaps with an empty guard band keep their result; the
        others are run again under the current eps (`WASBNet.recertify_subset`).  Returns the updated xyv, or None when eps grew
        past the guard factor and the whole clip has to be run again."""
        todo = self.net.recertify_subset(idx, win, status_host, eps_used, frames_u8)
        if todo is None:
            return None
        self.recertified_heatmaps += int(todo.size)
        if todo.size:
            sel = torch.as_tensor(todo, device=self.device)
            xyv[sel] = self._refine.refine_windows_device(idx[sel], win[sel], self.net_h, self.net_w, 1920, 1080, self._lib.REFINE_TABLE)
            # (status_host[todo] now holds the re-runs' own status: 0 = single candidate under the current eps, 1 = fp32 values)
        return xyv

    def _detect_blocking(self, frames_u8, full=False, counted=False):
        """One clip, start to finish, with the audit folded in: (xyv device tensor) whose every index is certified under the
        current eps; full=True: (xyv, idx, win, host status 0/1/2) of the pass that produced it.  Loops only when an audit widens
        eps past the guard factor (eps only grows)."""
        rerun = bool(counted)          # counted=True: the clip went through submit() already (frames_seen, audits)
        while True:
            eps_used = self._calibrated(frames_u8)
            ticket = self._start_audit(frames_u8, rerun)
            self._rerun_pass = rerun
            rerun = True
            xyv, idx, win, status, info = self._detect(frames_u8)
            if status is None:
                self._rerun_pass = False
                return (xyv, idx, win, None) if full else xyv
            n_crops, cand_err = self.net.decode_info(info.cpu().numpy())
            st = status.cpu().numpy()
            if self._after_clip(n_crops, cand_err, ticket) or eps_used < self.certify_eps:
                out = self._recertify(frames_u8, xyv, idx, win, st, eps_used)
                if out is None:
                    self.recertified_clips += 1
                    continue
                xyv = out
            xyv = self._repair(frames_u8, xyv, idx, win, st)
            self._rerun_pass = False
            return (xyv, idx, win, st & 3) if full else xyv

    def process_clip(self, frames_u8, table_px, fps):
        xyv = self._detect_blocking(frames_u8)
        spin, p3, nvalid = self.uplift_segments(xyv.cpu().numpy(), table_px, fps)
        return {'xyv': xyv, 'spin': spin, 'pos3d': p3, 'n_valid': nvalid}

    # Two-phase form of process_clip for back-to-back clips: `submit` only enqueues the detector (and an asynchronous copy
    # of its (N,3) result into pinned host memory) and returns at once; `collect` waits for that copy, runs the host glue
    # and enqueues the uplift.  Submitting clip k+1 before collecting clip k keeps the GPU busy while the host filters
    # and pads the detections of clip k.
    def submit(self, frames_u8):
        # consecutive clips are issued on two alternating streams: the fp32 crop passes of the certified argmax (the handle's own
        # stream, behind clip k's bf16 pass) then overlap with the bf16 micro-batches of clip k+1 instead of delaying them
        subs = self.submit_streams()
        sub = subs['streams'][subs['next']]
        subs['next'] ^= 1
        sub.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(sub):
            return self._submit(frames_u8, sub)

    def submit_streams(self):
        """The two alternating streams `submit` issues clips on (created on first use; callers that are about to create other
        streams -- a process group -- call this first so that the worker's stream-to-queue mapping does not depend on them)."""
        subs = self.__dict__.setdefault('_sub', None)
        if subs is None:
            subs = self._sub = {'streams': [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)], 'next': 0}
        return subs

    def _pinned(self, key, like):
        """A pinned host buffer shaped like `like` from the worker's pool (returned to it by collect): any number of clips may be
        in flight between submit and collect."""
        pool = self.__dict__.setdefault('_pin_pool', {})
        free = pool.setdefault((key, tuple(like.shape), like.dtype), [])
        return free.pop() if free else torch.empty(like.shape, dtype=like.dtype, pin_memory=True)

    def _unpin(self, key, buf):
        if buf is not None:
            self._pin_pool[(key, tuple(buf.shape), buf.dtype)].append(buf)

    def _submit(self, frames_u8, sub):
        eps_used = self._calibrated(frames_u8)
        audit = self._start_audit(frames_u8)          # side stream: shares the GPU with this clip's detector pass
        xyv, idx, win, status, info = self._detect(frames_u8)
        host = self._pinned('xyv', xyv)
        host.copy_(xyv, non_blocking=True)
        st_host = info_host = None
        if status is not None:
            st_host = self._pinned('status', status)
            st_host.copy_(status, non_blocking=True)
            info_host = self._pinned('info', info)
            info_host.copy_(info, non_blocking=True)
        mg_host = None
        if self._last_margin is not None:
            mg_host = self._pinned('margin', self._last_margin)
            mg_host.copy_(self._last_margin, non_blocking=True)
            self._last_margin.record_stream(sub)
        done = torch.cuda.Event()
        done.record()
        for t in (xyv, idx, win, frames_u8) + ((status, info) if status is not None else ()):
            t.record_stream(sub)
        return {'margin': mg_host, 'xyv': xyv, 'host': host, 'done': done, 'frames': frames_u8, 'idx': idx, 'win': win, 'status': st_host, 'info': info_host,
                'stream': sub, 'audit': audit, 'eps': eps_used}

    def collect(self, ticket, table_px, fps):
        ticket['done'].synchronize()
        torch.cuda.current_stream(self.device).wait_stream(ticket['stream'])
        rerun_status = None
        if ticket.get('status') is not None:
            n_crops, cand_err = self.net.decode_info(ticket['info'].numpy())
            widened = self._after_clip(n_crops, cand_err, ticket.get('audit'))
            st = ticket['status'].numpy()
            if widened or ticket['eps'] < self.certify_eps:
                # this clip was certified under an eps that an audit has since found too small: the heatmaps whose guard band is
                # not empty are run again (a few per clip); the whole clip only when eps grew past the guard factor (blocking; rare)
                out = self._recertify(ticket['frames'], ticket['xyv'], ticket['idx'], ticket['win'], st, ticket['eps'])
                if out is None:
                    self.recertified_clips += 1
                    # the ticket then describes the pass that produced its detections (indices, windows, status), not the stale one
                    out, ticket['idx'], ticket['win'], rerun_status = self._detect_blocking(ticket['frames'], full=True, counted=True)
                    st = None
                ticket['xyv'] = out
                ticket['host'].copy_(ticket['xyv'])
            if st is not None and ((st & 3) == 2).any():
                ticket['xyv'] = self._repair(ticket['frames'], ticket['xyv'], ticket['idx'], ticket['win'], st)
                ticket['host'].copy_(ticket['xyv'])
        # the uplift (about a hundred small launches for a handful of trajectories) runs on a side stream, so it shares
        # the GPU with the detector of the clip submitted in the meantime instead of queueing behind it
        side = self.__dict__.get('_side')
        if side is None:
            side = self._side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(side):
            spin, p3, nvalid = self.uplift_segments(ticket['host'].numpy(), table_px, fps)
        side.synchronize()
        # the results were allocated under the side stream and are consumed on the caller's stream (gather_records, RCCL,
        # user code): tell the caching allocator, so their blocks are not handed to the next clip's side-stream uplift
        # while reads queued on the caller's stream are still pending
        cur = torch.cuda.current_stream(self.device)
        for t in (spin, p3, nvalid):
            t.record_stream(cur)
        status_host = None if ticket.get('status') is None else (ticket['status'].numpy() & 3)          # 0 / 1 / 2 (guard bit dropped)
        if status_host is not None and rerun_status is not None:
            status_host = rerun_status
        if ticket.get('margin') is not None and self.margin_log is not None:
            self.margin_log.append(ticket['margin'].numpy().copy())
        for k in ('host', 'status', 'info', 'margin'):
            self._unpin('xyv' if k == 'host' else k, ticket.get(k))
            ticket[k] = None
        ticket['status_host'] = status_host
        return {'xyv': ticket['xyv'], 'spin': spin, 'pos3d': p3, 'n_valid': nvalid, 'status': status_host}

    RECORD_KEYS = ('xyv', 'spin', 'pos3d', 'n_valid')          # what a step hands to gather_records

    def queue_groups(self, cycles=20_000_000):
        """Which of the worker's streams share a hardware queue: HIP maps the streams of a process onto GPU_MAX_HW_QUEUES (4)
        queues in the order in which they are first used, kernels of two streams on one queue do not overlap, and the pipeline's
        throughput moves by up to 6 % with the grouping.  Pairwise probes with two spin kernels (co-resident when both
        take the time of one).  Returns a canonical string, e.g. 'submit0+lane1 | submit1+crops | lane0 | audit+default'; every rank
        of a multi-GPU run should report the same one (bench.py gathers them).  Takes a few hundred milliseconds; idle GPU assumed."""
        dev = self.device
        named = [('submit%d' % k, s) for k, s in enumerate(self.submit_streams()['streams'])]
        ints = self.net.internal_streams()
        n_lanes = len(ints) - (1 if self.net.certified else 0)
        named += [('lane%d' % k, s) for k, s in enumerate(ints[:n_lanes])] + [('crops', s) for s in ints[n_lanes:]]
        if getattr(self.net, '_audit_stream', None) is not None:
            named.append(('audit', self.net._audit_stream))
        if self.__dict__.get('_side') is not None:
            named.append(('uplift', self._side))
        named.append(('default', torch.cuda.default_stream(dev)))

        def spin_ms(streams):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            cur = torch.cuda.current_stream(dev)
            e0.record(cur)
            for st in streams:
                st.wait_event(e0)
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            e1.record(cur)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1)
        one = spin_ms([named[0][1]])
        groups = []
        for name, st in named:
            for g in groups:
                if spin_ms([g[0][1], st]) > 1.6 * one:
                    g.append((name, st))
                    break
            else:
                groups.append([(name, st)])
        return ' | '.join('+'.join(n for n, _ in g) for g in groups)
