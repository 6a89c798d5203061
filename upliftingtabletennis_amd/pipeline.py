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


def gather_records(records, dist=None, dst=0):
    """Gather a dict of same-dtype-per-key tensors (first dim may differ per rank) on rank `dst`.
    Returns {key: [tensor_of_rank0, tensor_of_rank1, ...]} on dst, None elsewhere.  With dist=None (single
    process) it just wraps the local records.  Works with the nccl (=RCCL) and gloo backends."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: [v] for k, v in records.items()}
    world, rank = dist.get_world_size(), dist.get_rank()
    out = {} if rank == dst else None
    on_host = dist.get_backend() == 'gloo'        # gloo (CPU tests, single-GPU dry runs) gathers on the host; nccl = RCCL on device
    for k in sorted(records):
        v = records[k].contiguous()
        if on_host:
            v = v.cpu()
        n = torch.tensor([v.shape[0]], dtype=torch.int64, device=v.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes + [1])
        pad = torch.zeros((mx,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[:v.shape[0]] = v
        # all_gather (every backend implements it on device tensors; the records are a few KB) and keep the result on `dst` only
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        if rank == dst:
            out[k] = [b[:s] for b, s in zip(bufs, sizes)]
    return out


class StreamWorker:
    """Per-GPU worker: owns one CNN handle, one uplift handle, and runs whole clips through the path.
    Raises RuntimeError without a HIP device (no CPU fallback)."""

    def __init__(self, device, wasb_state_dict, uplift_state_dict, net_wh=(1280, 704), max_triples=256, uplift_size='large',
                 traj_len=TRAJ_LEN_DEFAULT, seq_len=50, dtype='bf16', certify=True):
        from . import glue, refine, uplift, wasb, _lib
        _lib.require_gpu()
        self._glue, self._refine, self._uplift, self._lib = glue, refine, uplift, _lib
        self.device = torch.device(device)
        self.net_w, self.net_h = net_wh
        self.traj_len, self.seq_len = traj_len, seq_len
        self.net = wasb.WASBNet(wasb_state_dict, resolution=net_wh, max_batch=max_triples, dtype=dtype, device=self.device)
        self.up = uplift.get_model('connectstage', uplift_size, 'dynamic', 'new', state_dict=uplift_state_dict,
                                   max_batch=max(64, (max_triples + traj_len - 1) // traj_len), max_len=seq_len, device=self.device)
        # certified argmax (bit-exact fp32 indices from the bf16 path, csrc/certify.hip): calibrated on the first clip seen
        self.certify = bool(certify) and dtype == 'bf16'
        self.certify_eps = None
        self.fp32_reruns = 0

    def detect(self, frames_u8):
        """(N,h,w,3) uint8 on the device -> (N-2,3) float64 [x, y, visibility] in 1920x1080 px (table-variant refine,
        like interface.py:116)."""
        return self._detect(frames_u8)[0]

    def _detect(self, frames_u8):
        if self.certify and self.certify_eps is None:
            self.certify_eps = self.net.calibrate(frames_u8, n=4)
        _, idx, win = self.net.forward_frames(frames_u8, want_heatmap=False)
        xyv = self._refine.refine_windows_device(idx, win, self.net_h, self.net_w, 1920, 1080, self._lib.REFINE_TABLE)
        status = self.net.certify_status(idx.shape[0]) if self.certify else None
        return xyv, idx, win, status

    def _repair(self, frames_u8, xyv, idx, win, status_host):
        """Rare slow path: heatmaps the certified argmax could not settle inside its crop budget are re-run on the full-frame
        fp32 handle, so that every detection comes from the fp32 argmax."""
        bad = np.nonzero(status_host == 2)[0]
        if bad.size == 0:
            return xyv
        self.fp32_reruns += self.net.fix_uncertified(idx, win, frames_u8=frames_u8)
        sel = torch.as_tensor(bad, device=self.device)
        xyv[sel] = self._refine.refine_windows_device(idx[sel], win[sel], self.net_h, self.net_w, 1920, 1080, self._lib.REFINE_TABLE)
        return xyv

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

    def process_clip(self, frames_u8, table_px, fps):
        xyv, idx, win, status = self._detect(frames_u8)
        if status is not None:
            xyv = self._repair(frames_u8, xyv, idx, win, status.cpu().numpy())
        spin, p3, nvalid = self.uplift_segments(xyv.cpu().numpy(), table_px, fps)
        return {'xyv': xyv, 'spin': spin, 'pos3d': p3, 'n_valid': nvalid}

    # Two-phase form of process_clip for back-to-back clips: `submit` only enqueues the detector (and an asynchronous copy
    # of its (N,3) result into pinned host memory) and returns at once; `collect` waits for that copy, runs the host glue
    # and enqueues the uplift.  Submitting clip k+1 before collecting clip k keeps the GPU busy while the host filters
    # and pads the detections of clip k.
    def submit(self, frames_u8):
        # consecutive clips are issued on two alternating streams: the fp32 crop passes of the certified argmax (the handle's own
        # stream, behind clip k's bf16 pass) then overlap with the bf16 micro-batches of clip k+1 instead of delaying them
        subs = self.__dict__.setdefault('_sub', None)
        if subs is None:
            subs = self._sub = {'streams': [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)], 'next': 0}
        sub = subs['streams'][subs['next']]
        subs['next'] ^= 1
        sub.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(sub):
            return self._submit(frames_u8, sub)

    def _submit(self, frames_u8, sub):
        xyv, idx, win, status = self._detect(frames_u8)
        ring = self.__dict__.setdefault('_pinned', {})          # two pinned buffers per shape, used alternately
        slot = ring.setdefault(tuple(xyv.shape), {'bufs': [None, None], 'next': 0})
        i = slot['next']; slot['next'] = 1 - i
        if slot['bufs'][i] is None:
            slot['bufs'][i] = torch.empty(xyv.shape, dtype=xyv.dtype, pin_memory=True)
        host = slot['bufs'][i]
        host.copy_(xyv, non_blocking=True)
        st_host = None
        if status is not None:
            st_host = slot.setdefault('status', [None, None])
            if st_host[i] is None:
                st_host[i] = torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
            st_host = st_host[i]
            st_host.copy_(status, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        for t in (xyv, idx, win, frames_u8) + ((status,) if status is not None else ()):
            t.record_stream(sub)
        return {'xyv': xyv, 'host': host, 'done': done, 'frames': frames_u8, 'idx': idx, 'win': win, 'status': st_host, 'stream': sub}

    def collect(self, ticket, table_px, fps):
        ticket['done'].synchronize()
        torch.cuda.current_stream(self.device).wait_stream(ticket['stream'])
        if ticket.get('status') is not None:
            # crop budget of the next clips: twice what this clip needed (+ slack); clips that outgrow it are flagged and repaired below
            need = int((ticket['status'].numpy() == 1).sum())
            self.net.certify_budget(2 * need + 16)
        if ticket.get('status') is not None and (ticket['status'].numpy() == 2).any():
            ticket['xyv'] = self._repair(ticket['frames'], ticket['xyv'], ticket['idx'], ticket['win'], ticket['status'].numpy())
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
        return {'xyv': ticket['xyv'], 'spin': spin, 'pos3d': p3, 'n_valid': nvalid}
