"""f2 (synthetic-trajectory generator) on the MI355X, through the C-ABI, against the reference-generated goldens and the
CPU oracle.  /root/reference is never read here."""
import numpy as np
import pytest
import torch

from conftest import has_gpu
from oracle import trajgen_ref as T

pytestmark = pytest.mark.gpu
if has_gpu():
    from upliftingtabletennis_amd import trajgen


def test_device_sampler_matches_reference_random_module(golden):
    """CPython's MT19937 seeding (init_by_array) and the nine uniform draws of `_init_simulation` on the device:
    positions (no transcendental functions) bit-exact; velocity / spin differ only through sin/cos/atan2 rounding
    (device libm vs glibc): <= 1e-13 relative."""
    g = golden('trajgen.npz')
    seeds = g['init_seeds']
    for mode in T.MODES:
        for direction in T.DIRECTIONS:
            ref = g['init/%s/%s' % (mode, direction)]
            res = trajgen.simulate_seeds(seeds, mode, direction, want_init=True)
            got = res['init'].cpu().numpy().T
            assert np.array_equal(got[:, :3], ref[:, :3]), (mode, direction)
            assert np.abs(got[:, 3:] - ref[:, 3:]).max() <= 1e-13 * np.abs(ref[:, 3:]).max(), (mode, direction)


@pytest.mark.parametrize('mode,direction', [('intermediate', 'left_to_right'), ('first_short', 'right_to_left')])
def test_device_integrator_tracks_oracle(mode, direction):
    """Same RK4 / same force model in fp64 on both sides: the sampled states agree to 1e-9 m while the ball is in free
    flight and to 1e-6 m over the whole second (contacts amplify last-bit differences); sample counts are equal."""
    seeds = list(range(16))
    pos, vel, rot, ns = T.simulate(seeds, mode, direction)
    res = trajgen.simulate_seeds(seeds, mode, direction)
    got = res['samples'].cpu().numpy()                    # (S, 9, N)
    gns = res['n_saved'].cpu().numpy()
    assert np.array_equal(gns, ns)
    for i in range(len(seeds)):
        n = int(ns[i])
        d = np.abs(got[:n, 0:3, i] - pos[i, :n]).max(axis=1) if n else np.zeros(0)
        assert n == 0 or d[:min(n, 20)].max() <= 1e-9, (i, d[:20].max())
        assert n == 0 or d.max() <= 1e-6, (i, d.max())
        if n:
            assert np.abs(got[:n, 3:6, i] - vel[i, :n]).max() <= 1e-4 and np.abs(got[:n, 6:9, i] - rot[i, :n]).max() <= 1e-2


@pytest.mark.parametrize('mode', T.MODES)
def test_device_selection_is_exact_on_oracle_tracks(mode):
    """`ttup_trajgen_select` on the oracle's own samples: kept length and bounce times bit-equal to the restated (and
    reference-pinned) selection."""
    times = T.save_times()
    for direction in T.DIRECTIONS:
        seeds = list(range(12))
        pos, vel, rot, ns = T.simulate(seeds, mode, direction)
        S = len(times)
        samples = np.zeros((S, 9, len(seeds)))
        samples[:, 0:3] = pos.transpose(1, 2, 0)
        nk, bo, nb = trajgen.select_positions(torch.from_numpy(samples).cuda(), torch.from_numpy(ns).cuda(), mode, direction)
        nk, bo, nb = nk.cpu().numpy(), bo.cpu().numpy(), nb.cpu().numpy()
        for i in range(len(seeds)):
            want = T.select(pos[i, :ns[i]], times, mode, direction)
            if want is None:
                assert nk[i] == 0, (mode, direction, i)
            else:
                assert nk[i] == want[0] and nb[i] == len(want[1]) and np.array_equal(bo[i, :nb[i]], want[1]), (mode, direction, i)


def test_device_pipeline_reproduces_reference_worker_decisions(golden):
    """Device sampler + integrator + selection against what the REFERENCE's `find_valid_trajectories_worker` accepted
    (run on the oracle integrator): same seeds, same kept lengths, same bounce times, first trajectory within 1e-6 m."""
    g = golden('trajgen.npz')
    n_sel = int(g['n_sel'][0])
    accepted = 0
    for mode in T.MODES:
        for direction in T.DIRECTIONS:
            key = 'worker/%s/%s' % (mode, direction)
            res = trajgen.simulate_seeds(list(range(n_sel)), mode, direction)
            nk = res['n_keep'].cpu().numpy()
            got = np.nonzero(nk)[0]
            assert np.array_equal(got, g[key + '/seeds']), (mode, direction, got, g[key + '/seeds'])
            assert np.array_equal(nk[got], g[key + '/n'])
            nb = res['n_bounces'].cpu().numpy()[got]
            assert np.array_equal(nb, g[key + '/n_bounces'])
            bo = res['bounces'].cpu().numpy()[got]
            flat = np.concatenate([bo[i, :nb[i]] for i in range(len(got))]) if len(got) else np.zeros(0)
            assert np.array_equal(flat, g[key + '/bounces'])
            if len(got):
                n = int(nk[got[0]])
                tr = res['samples'][:n, :, int(got[0])].cpu().numpy()
                assert np.abs(tr[:, 0:3] - g[key + '/first_positions']).max() <= 1e-6
            accepted += len(got)
    assert accepted >= 50


def test_device_decisions_for_every_mode(golden):
    """Seeds accepted in each of the six modes (found by a device search) and their neighbours: the device's accept / reject
    decisions, kept lengths and bounce times equal the reference worker's (run on the oracle integrator)."""
    g = golden('trajgen.npz')
    for mode in T.MODES:
        for direction in T.DIRECTIONS:
            key = 'rare/%s/%s' % (mode, direction)
            seeds = g[key + '/all_seeds']
            res = trajgen.simulate_seeds(seeds, mode, direction)
            nk = res['n_keep'].cpu().numpy()
            sel = np.nonzero(nk)[0]
            assert np.array_equal(seeds[sel], g[key + '/seeds']), (mode, direction)
            assert np.array_equal(nk[sel], g[key + '/n'])
            nb = res['n_bounces'].cpu().numpy()[sel]
            assert np.array_equal(nb, g[key + '/n_bounces'])
            bo = res['bounces'].cpu().numpy()[sel]
            assert np.array_equal(np.concatenate([bo[i, :nb[i]] for i in range(len(sel))]), g[key + '/bounces'])
            assert len(sel) >= 4


def test_get_valid_trajectories_mirrors_reference_api(tmp_path):
    """Pool ordering, truncation, dictionary keys, save_dataset layout; contents against the oracle's `generate`."""
    got = trajgen.get_valid_trajectories(6, 4, 'final_lose', 'left_to_right', batches_per_launch=3)
    want = T.generate(6, 4, 'final_lose', 'left_to_right')
    assert [t['seed'] for t in got] == [t['seed'] for t in want]
    for a, b in zip(got, want):
        assert set(a) == set(b) == {'positions', 'velocities', 'rotations', 'times', 'Mext', 'Mint', 'bounces', 'seed'}
        assert a['positions'].shape == b['positions'].shape and np.abs(a['positions'] - b['positions']).max() <= 1e-6
        assert np.array_equal(a['times'], b['times']) and np.array_equal(a['bounces'], b['bounces'])
        assert np.array_equal(a['Mext'], b['Mext']) and np.array_equal(a['Mint'], b['Mint'])
    trajgen.save_dataset(str(tmp_path / 'ds'), got)
    p = np.load(str(tmp_path / 'ds' / 'trajectory_0003' / 'positions.npy'))
    assert np.array_equal(p, got[3]['positions']) and not (tmp_path / 'ds' / 'trajectory_0003' / 'seed.npy').exists()
    one = trajgen._run_single_simulation(got[0]['seed'], 'final_lose', 'left_to_right')
    assert one is not None and 'seed' not in one and np.array_equal(one['positions'], got[0]['positions'])
    with pytest.raises(AssertionError):
        trajgen.get_valid_trajectories(1, 1, 'nonsense', 'left_to_right')


def test_trajgen_argument_validation():
    lib = __import__('upliftingtabletennis_amd._lib', fromlist=['x'])
    with pytest.raises(ValueError):
        trajgen.simulate_seeds([0, 1], 'intermediate', 'left_to_right', substeps=0)
    assert lib.load().ttup_trajgen_max_samples() == len(T.save_times())


def test_free_flight_convergence_on_device():
    """Self-consistency of the unpinned physics on the device: substeps 1 -> 2 cuts the free-flight error ~16x."""
    seeds = [3, 7, 11, 19]
    ref = trajgen.simulate_seeds(seeds, 'final_lose', 'left_to_right', substeps=16)['samples'][:20, 0:3].cpu().numpy()
    e1 = np.abs(trajgen.simulate_seeds(seeds, 'final_lose', 'left_to_right', substeps=1)['samples'][:20, 0:3].cpu().numpy() - ref).max()
    e2 = np.abs(trajgen.simulate_seeds(seeds, 'final_lose', 'left_to_right', substeps=2)['samples'][:20, 0:3].cpu().numpy() - ref).max()
    assert e1 < 1e-5 and 8.0 < e1 / max(e2, 1e-300) < 24.0, (e1, e2)
