"""Synthetic-trajectory generator on the MI355X (BASELINE config 5): mirror of
syntheticdataset/mujocosimulation.py -- `get_valid_trajectories`, `save_dataset`, `_run_single_simulation` keep the
reference's names, argument meaning and output dictionaries ('positions', 'velocities', 'rotations', 'times', 'Mext',
'Mint', 'bounces', 'seed').  Seeds are sampled, integrated and filtered on the device through the C-ABI
(`ttup_trajgen_simulate`, `ttup_trajgen_select`); the host only orders the survivors the way the reference's process
pool returns them.  No CPU fallback.

MuJoCo is not available, so the flight/contact arithmetic is a restatement of its documented model (parity unpinned,
see DESIGN.md); sampling (`random.Random(seed)`), the sampling loop and all selection rules are pinned by the
reference's own code.
"""
import collections.abc
import ctypes
import os

import numpy as np
import torch

from . import _lib

MODES = ['final_lose', 'final_win', 'intermediate', 'first_good', 'first_short', 'first_long']     # OOB_DEFINITIONS order
DIRECTIONS = ['left_to_right', 'right_to_left']
HEIGHT, WIDTH = 1080, 1920
FPS = 500
SUBSTEPS = 4
_FX, _FY = 2033, 2180
_CAMERA_POS = np.array([0.04381194, 8.92938715, 5.40070126])
_CAMERA_UP = np.array([7.81340900e-04, -4.33644716e-01, 9.01083598e-01])
_CAMERA_RIGHT = np.array([-0.99998599, 0.00437903, 0.0029745])


def camera_matrices():
    """Mext (4,4), Mint (3,3) of the fixed camera, as `_calc_cammatrices` (helper.py:262-280) derives them from the
    MuJoCo camera frame (x normalised, y orthogonalised against x, z = x cross y; rows x, -y, -z)."""
    x = _CAMERA_RIGHT / np.linalg.norm(_CAMERA_RIGHT)
    y = _CAMERA_UP - x * np.dot(x, _CAMERA_UP)
    y = y / np.linalg.norm(y)
    z = np.cross(x, y)
    R = np.stack([x, -y, -z])
    ex = np.eye(4)
    ex[:3, :3] = R
    ex[:3, 3] = -R @ _CAMERA_POS
    fx = (_FX / WIDTH) / 1.0 * WIDTH
    fy = (_FY / HEIGHT) / 1.0 * HEIGHT
    return ex, np.array([[fx, 0, (WIDTH - 1) / 2], [0, fy, (HEIGHT - 1) / 2], [0, 0, 1.0]])


def save_times():
    """Time labels of the sampling loop (floating-point accumulation of 1/FPS, mujocosimulation.py:116,150)."""
    t, out = 0.0, []
    while t < 1.0:
        out.append(t)
        t += 1 / FPS
    return np.array(out)


def seed_order(current_seed, batch_size, num_processes):
    """Order in which one batch comes back from the reference's Pool: process j owns seeds j, j+P, j+2P, ... (:226)."""
    out = []
    for j in range(num_processes):
        out.extend(range(current_seed + j, current_seed + batch_size, num_processes))
    return out


def simulate_seeds(seeds, mode, direction, substeps=SUBSTEPS, device=None, want_init=False):
    """Run the sampling loop and the selection for explicit seeds.  Returns a dict of device tensors:
    samples (S, 9, N) float64, n_saved (N,), n_keep (N,) [0 = rejected], bounces (N, 4), n_bounces (N,), init (9, N)."""
    _lib.require_gpu()
    if mode not in MODES:
        raise AssertionError('Mode %s not supported.' % mode)                       # mujocosimulation.py:268
    if direction not in DIRECTIONS:
        raise AssertionError('Direction %s not supported.' % direction)             # :270
    lib = _lib.load()
    device = torch.device(device if device is not None else 'cuda')
    sd = torch.as_tensor(np.asarray(seeds, dtype=np.int64), device=device)
    n = int(sd.numel())
    if substeps < 1 or substeps > 64:
        raise ValueError('substeps must be in [1, 64]')
    S = lib.ttup_trajgen_max_samples()
    samples = torch.zeros((S, 9, n), dtype=torch.float64, device=device)
    n_saved = torch.zeros((n,), dtype=torch.int32, device=device)
    n_keep = torch.zeros((n,), dtype=torch.int32, device=device)
    bounces = torch.zeros((n, 4), dtype=torch.float64, device=device)
    n_bounces = torch.zeros((n,), dtype=torch.int32, device=device)
    init = torch.zeros((9, n), dtype=torch.float64, device=device) if want_init else None
    ws_bytes = lib.ttup_trajgen_workspace_bytes(n)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=device)
    ex, mint = camera_matrices()
    cam = np.ascontiguousarray(np.concatenate([ex.reshape(-1), mint.reshape(-1)]), dtype=np.float64)
    m, d = MODES.index(mode), DIRECTIONS.index(direction)
    if n == 0:              # no seeds: empty results
        return {'samples': samples, 'n_saved': n_saved, 'n_keep': n_keep, 'bounces': bounces, 'n_bounces': n_bounces, 'init': init, 'seeds': sd}
    with torch.cuda.device(device):
        _lib.check(lib.ttup_trajgen_simulate(_lib.ptr(sd), n, m, d, int(substeps), cam.ctypes.data_as(ctypes.c_void_p), _lib.ptr(samples), _lib.ptr(n_saved),
                                             _lib.ptr(init), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
        _lib.check(lib.ttup_trajgen_select(_lib.ptr(samples), _lib.ptr(n_saved), n, m, d, _lib.ptr(n_keep), _lib.ptr(bounces), _lib.ptr(n_bounces),
                                           _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
    return {'samples': samples, 'n_saved': n_saved, 'n_keep': n_keep, 'bounces': bounces, 'n_bounces': n_bounces, 'init': init, 'seeds': sd}


def select_positions(samples, n_saved, mode, direction):
    """Selection only (count hits, cut, reject) on given samples (S, 9, N) / n_saved (N,) device tensors."""
    _lib.require_gpu()
    lib = _lib.load()
    n = int(n_saved.numel())
    dev = samples.device
    n_keep = torch.zeros((n,), dtype=torch.int32, device=dev)
    bounces = torch.zeros((n, 4), dtype=torch.float64, device=dev)
    n_bounces = torch.zeros((n,), dtype=torch.int32, device=dev)
    ws_bytes = lib.ttup_trajgen_workspace_bytes(n)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ttup_trajgen_select(_lib.ptr(samples.contiguous()), _lib.ptr(n_saved.to(torch.int32).contiguous()), n, MODES.index(mode),
                                           DIRECTIONS.index(direction), _lib.ptr(n_keep), _lib.ptr(bounces), _lib.ptr(n_bounces), _lib.ptr(ws), ws_bytes,
                                           _lib.stream_ptr()))
    return n_keep, bounces, n_bounces


def _seed_order_array(current_seed, batch_size, num_processes):
    """`seed_order` as an int64 array (a stable sort of the batch by seed % num_processes)."""
    k = np.arange(batch_size, dtype=np.int64)
    return current_seed + k[np.argsort(k % num_processes, kind='stable')]


class TrajectoryBatch(collections.abc.Sequence):
    """What `get_valid_trajectories` returns: the survivors as STACKED arrays, read like the reference's list of dictionaries.

    `len(batch)`, `batch[i]`, slices and iteration give reference-format dictionaries ('positions', 'velocities', 'rotations',
    'times', 'Mext', 'Mint', 'bounces', 'seed'; mujocosimulation.py:213-218) that are built ON ACCESS as views of the stacked
    buffers -- 125 000 trajectories are a handful of large arrays, not 125 000 x 8 small ones (building those was 96 % of the
    call's time in round 4).  The views are read-only where several trajectories share the storage ('times', 'Mext', 'Mint').
    `chunks`: one entry per device launch with `rows` (R, 9) float64 -- the kept samples of its trajectories back to back
    (position, velocity, rotation) -- `offsets` (V+1,), `n_keep`, `bounces` (V, 4), `n_bounces`, `seeds`; `stacked()` concatenates them."""

    def __init__(self, chunks, times, ex, mint):
        self.chunks = chunks
        self.times, self.Mext, self.Mint = times, ex, mint
        self.times.setflags(write=False)
        self._starts = np.concatenate([[0], np.cumsum([len(c['n_keep']) for c in chunks])]).astype(np.int64)

    def __len__(self):
        return int(self._starts[-1])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n_all = len(self)
        if i < 0:
            i += n_all
        if not 0 <= i < n_all:
            raise IndexError('trajectory index out of range')
        c = int(np.searchsorted(self._starts, i, side='right')) - 1
        ch, j = self.chunks[c], int(i - self._starts[c])
        o, n = int(ch['offsets'][j]), int(ch['n_keep'][j])
        tr = ch['rows'][o:o + n]
        return {'positions': tr[:, 0:3], 'velocities': tr[:, 3:6], 'rotations': tr[:, 6:9], 'times': self.times[:n],
                'Mext': np.broadcast_to(self.Mext, (n, 4, 4)), 'Mint': np.broadcast_to(self.Mint, (n, 3, 3)),
                'bounces': ch['bounces'][j, :int(ch['n_bounces'][j])], 'seed': int(ch['seeds'][j])}

    def to_list(self):
        """The reference's return type exactly (mujocosimulation.py:213-218, :238): a plain `list` of dictionaries whose arrays are
        INDEPENDENT, writable numpy copies -- for callers that shuffle, append, pickle or edit the trajectories in place.  Costs what
        the reference's own list costs (about 2 s per 125 000 trajectories); the views of `batch[i]` stay the fast path."""
        out = []
        for d in self:
            out.append({k: (np.array(v.cpu() if torch.is_tensor(v) else v) if not isinstance(v, int) else v) for k, v in d.items()})
        return out

    def __add__(self, other):          # `trajectories += more` / `a + b` as with the reference's lists
        return self.to_list() + (other.to_list() if isinstance(other, TrajectoryBatch) else list(other))

    def __radd__(self, other):
        return list(other) + self.to_list()

    def stacked(self):
        """{'rows' (R, 9), 'offsets' (N+1,), 'n_keep' (N,), 'bounces' (N, 4), 'n_bounces' (N,), 'seeds' (N,)} over all launches."""
        cat = torch.cat if any(torch.is_tensor(c['rows']) for c in self.chunks) else np.concatenate
        nk = np.concatenate([c['n_keep'] for c in self.chunks]) if self.chunks else np.zeros(0, np.int64)
        return {'rows': cat([c['rows'] for c in self.chunks]) if self.chunks else np.zeros((0, 9)),
                'offsets': np.concatenate([[0], np.cumsum(nk)]).astype(np.int64), 'n_keep': nk,
                'bounces': np.concatenate([c['bounces'] for c in self.chunks]) if self.chunks else np.zeros((0, 4)),
                'n_bounces': np.concatenate([c['n_bounces'] for c in self.chunks]) if self.chunks else np.zeros(0, np.int32),
                'seeds': np.concatenate([c['seeds'] for c in self.chunks]) if self.chunks else np.zeros(0, np.int64)}


def get_valid_trajectories(num_trajectories, num_processes, mode, direction, substeps=SUBSTEPS, device=None, batches_per_launch=64,
                           as_numpy=True):
    """`get_valid_trajectories` (mujocosimulation.py:222-238): seeds are consumed in batches of min(1024, num_trajectories),
    each batch in the order the reference's `num_processes`-way pool returns it; the first `num_trajectories` survivors
    are returned.  Several batches are integrated per launch (`batches_per_launch`); the result does not depend on it.

    Returns a `TrajectoryBatch`: a sequence of reference-format dictionaries backed by stacked arrays.  Per launch the kept samples
    of the survivors are packed back to back on the device ((rows, 9) float64) and copied into pinned host memory on a side stream
    while the next launch integrates (as_numpy=False: they stay device tensors)."""
    times = save_times()
    ex, mint = camera_matrices()
    batch = min(1024, num_trajectories)
    dev = torch.device(device if device is not None else 'cuda')
    chunks, found, current = [], 0, 0
    copy_stream = None
    while found < num_trajectories:
        seeds = np.concatenate([_seed_order_array(current + k * batch, batch, num_processes) for k in range(batches_per_launch)])
        current += batch * batches_per_launch
        res = simulate_seeds(seeds, mode, direction, substeps, device)
        keep = res['n_keep']
        idx = torch.nonzero(keep > 0).flatten()
        idx = idx[:num_trajectories - found]
        v = int(idx.numel())
        if v == 0:
            continue
        nk_dev = keep[idx].to(torch.int64)
        sel = res['samples'].index_select(2, idx).permute(2, 0, 1)                  # (V, S, 9) view of the (S, 9, V) gather
        mask = torch.arange(sel.shape[1], device=sel.device)[None, :] < nk_dev[:, None]
        packed = sel[mask]                                                           # (rows, 9): every trajectory's kept samples, back to back
        meta = torch.stack([nk_dev, res['n_bounces'][idx].to(torch.int64), res['seeds'][idx].to(torch.int64)]).cpu().numpy()
        bo = res['bounces'][idx].cpu().numpy()
        nk = meta[0]
        if as_numpy:
            if copy_stream is None:
                copy_stream = torch.cuda.Stream(dev)
            host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_stream):
                host.copy_(packed, non_blocking=True)
            packed.record_stream(copy_stream)
            rows = host
        else:
            rows = packed
        chunks.append({'rows': rows, 'offsets': np.concatenate([[0], np.cumsum(nk)]).astype(np.int64), 'n_keep': nk,
                       'bounces': bo, 'n_bounces': meta[1].astype(np.int32), 'seeds': meta[2]})
        found += v
        del res, sel, mask, packed
    if copy_stream is not None:
        copy_stream.synchronize()
        for c in chunks:
            c['_pinned'] = c['rows']                 # keeps the pinned block alive for the views below
            c['rows'] = c['rows'].numpy()
    return TrajectoryBatch(chunks, times, ex, mint)


def save_dataset(path, trajectories_data):
    """`save_dataset` (mujocosimulation.py:241-248): one folder per trajectory, one .npy per key except 'seed'."""
    os.makedirs(path, exist_ok=True)
    for i, traj in enumerate(trajectories_data):
        save_path = os.path.join(path, 'trajectory_%04d' % i)
        os.makedirs(save_path, exist_ok=True)
        for key, value in traj.items():
            if key != 'seed':
                np.save(os.path.join(save_path, '%s.npy' % key), np.asarray(value))


def _run_single_simulation(seed, mode, direction, substeps=SUBSTEPS, device=None):
    """`_run_single_simulation` (:251-259): the trajectory of one seed, or None when it is rejected."""
    res = simulate_seeds([seed], mode, direction, substeps, device)
    n = int(res['n_keep'][0].item())
    if n == 0:
        return None
    tr = res['samples'][:n, :, 0].cpu().numpy()
    ex, mint = camera_matrices()
    nb = int(res['n_bounces'][0].item())
    return {'positions': tr[:, 0:3], 'velocities': tr[:, 3:6], 'rotations': tr[:, 6:9], 'times': save_times()[:n],
            'Mext': np.repeat(ex[None], n, 0), 'Mint': np.repeat(mint[None], n, 0), 'bounces': res['bounces'][0, :nb].cpu().numpy()}
