"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding helpers and the single collective
(gather of per-frame / per-trajectory records).  No compute runs here -- the HIP path has no CPU fallback."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from upliftingtabletennis_amd import pipeline


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 256):
        for world in (1, 2, 3, 8):
            parts = [pipeline.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pipeline.shard_range(4, 2, 2)


def test_frame_ranges_cover_all_triples_with_halo():
    n = 66
    seen = []
    for r in range(4):
        f0, f1, t0, nt = pipeline.frame_range_with_halo(n, 4, r)
        assert f1 - f0 == nt + 2 and f0 == t0
        seen += list(range(t0, t0 + nt))
    assert seen == list(range(n - 2))
    assert pipeline.frame_range_with_halo(2, 2, 1)[3] == 0


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_streams = 5
        s0, s1 = pipeline.shard_range(n_streams, world, rank)
        # fake per-stream records with the real record shapes: ragged number of frames / trajectories per rank
        rng = np.random.default_rng(100 + rank)
        nf = 10 * (s1 - s0) + rank
        rec = {'xyv': torch.from_numpy(rng.uniform(0, 1920, (nf, 3))),
               'spin': torch.from_numpy(rng.standard_normal((s1 - s0, 3)).astype(np.float32)),
               'pos3d': torch.from_numpy(rng.standard_normal((s1 - s0, 50, 3)).astype(np.float32)),
               'n_valid': torch.arange(s0, s1, dtype=torch.int64)}
        # count the collectives a gather issues: with the worker's record spec exactly ONE, without it two
        calls = {'n': 0}
        orig = {k: getattr(dist, k) for k in ('all_gather_into_tensor', 'all_gather', 'all_reduce', 'gather', 'broadcast')}
        for k, fn in orig.items():
            setattr(dist, k, (lambda fn: lambda *a, **kw: (calls.__setitem__('n', calls['n'] + 1), fn(*a, **kw))[1])(fn))
        spec = {'xyv': (64, (3,), torch.float64), 'spin': (8, (3,), torch.float32), 'pos3d': (8, (50, 3), torch.float32), 'n_valid': (8, (), torch.int64)}
        got = pipeline.gather_records(rec, dist, dst=0, spec=spec)
        assert calls['n'] == 1, calls
        got2 = pipeline.gather_records(rec, dist, dst=0)
        assert calls['n'] == 3, calls
        for k, fn in orig.items():
            setattr(dist, k, fn)
        if rank == 0:
            assert all(torch.equal(a, b) for k in got for a, b in zip(got[k], got2[k]))
            try:
                pipeline.gather_records({'xyv': rec['xyv'][:, :2]}, None)            # fine without dist
                pipeline._pack_records({'xyv': rec['xyv'][:, :2]}, {'xyv': spec['xyv']}, torch.device('cpu'))
                raise AssertionError('a record that does not fit its spec must be refused')
            except ValueError:
                pass
        if rank == 0:
            assert got is not None and set(got) == set(rec)
            assert [int(v.shape[0]) for v in got['spin']] == [pipeline.shard_range(n_streams, world, r)[1] - pipeline.shard_range(n_streams, world, r)[0] for r in range(world)]
            assert torch.equal(torch.cat(got['n_valid']), torch.arange(n_streams))
            assert torch.equal(got['xyv'][0], rec['xyv']) and got['xyv'][1].shape == (10 * (n_streams - s1) + 1, 3)
            other = np.random.default_rng(101)
            exp_xyv = other.uniform(0, 1920, (got['xyv'][1].shape[0], 3))
            assert np.array_equal(got['xyv'][1].numpy(), exp_xyv)
            open(os.path.join(out_dir, 'ok'), 'w').write('1')
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_records_world2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def test_gather_records_single_process():
    rec = {'a': torch.arange(6).reshape(3, 2)}
    got = pipeline.gather_records(rec, None)
    assert torch.equal(got['a'][0], rec['a'])


def test_crop_budget_follows_the_largest_demand_of_recent_clips():
    """StreamWorker._after_clip sizes the crop budget of the next calls from the last eight clips' demand (1.5 x the largest + 16):
    a hard clip behind an easy one keeps its budget (the round's varied-content regime lost 40 % of such a clip's heatmaps to the
    full-frame fp32 path under 'twice the last clip').  Host logic only: the handle is a stub."""
    class Net:
        def __init__(self):
            self.budgets = []
        def note_error(self, e, n=0):
            return e
        def eps_violated(self, e):
            return False
        def certify_budget(self, n):
            self.budgets.append(n)
    w = object.__new__(pipeline.StreamWorker)
    w.net = Net()
    w.certify_eps = 0.05
    for n_crops in (15, 68, 256, 256, 15, 68):
        assert w._after_clip(n_crops, 0.01, None) is False
    assert w.net.budgets == [38, 118, 400, 400, 400, 400]
    for n_crops in (10,) * 8:                      # the hard clips age out of the window
        w._after_clip(n_crops, 0.01, None)
    assert w.net.budgets[-1] == 31
