"""Shared pieces of the end-to-end parity tests (tests/golden/e2e.npz: the reference's own modules chained as interface.py
chains them, tools/make_goldens.py gen_e2e)."""
import numpy as np

from upliftingtabletennis_amd import synth, weights


def e2e_case(g, name):
    """Inputs of an end-to-end fixture, regenerated from its seeds: (frames uint8 (N,h,w,3), fps, ball / table / uplift state_dicts,
    detector resolution (W,H))."""
    n, h, w, s_ball, s_table, s_up, s_clip = [int(v) for v in g[name + '/meta']]
    frames, _ = synth.synth_frames(n, h, w, seed=s_clip)
    return (frames, float(g[name + '/fps']), weights.random_wasb_state_dict(s_ball, planted=True),
            weights.random_wasb_state_dict(s_table, planted=True, in_ch=3, head_out=13, plant_all_heads=True),
            weights.random_uplift_state_dict(s_up, 'large'), (w, h))


def check_spin_pos(spin, pos3d, g, name, tol):
    """(spin_local (3,), pos3d (T',3)) of a pipeline against the fixture.

    pos3d, |spin| and spin_z (= the global rotation's norm and z component: e_z of the local frame is the world z axis,
    uplifting/helper.py:394-420) are compared at `tol` relative to the largest entry, north_star's "3D position/spin within 1e-4 rel".
    The x / y components of the LOCAL spin go through e_x = v0 / |v0| with v0 = pos[1,:2] - pos[0,:2]: a position error dp turns the
    frame by up to 2 dp / |v0|, so they are held to tol + that angle times |spin| -- on the random-weight fixtures |v0| is ~1 % of
    the positions' scale, and the reference's own modules re-run with inputs that differ by 3e-6 px already move spin_x by 2.5e-4
    (tests/test_oracle_golden.py::test_e2e_oracle_matches_reference_chain measures exactly that).
    Returns the measured relative deviations (pos, |spin|, spin_z, spin_xy)."""
    spin, pos3d = np.asarray(spin, dtype=np.float64), np.asarray(pos3d, dtype=np.float64)
    rs, rp = g[name + '/spin'].astype(np.float64), g[name + '/pos3d'].astype(np.float64)
    assert pos3d.shape == rp.shape and spin.shape == (3,)
    pscale, sscale = np.abs(rp).max(), np.linalg.norm(rs)
    dp = np.abs(pos3d - rp).max()
    d_norm = abs(np.linalg.norm(spin) - sscale) / sscale
    d_z = abs(spin[2] - rs[2]) / sscale
    d_xy = np.abs(spin[:2] - rs[:2]).max() / sscale
    v0 = np.linalg.norm(rp[1, :2] - rp[0, :2])
    turn = 2.0 * np.sqrt(2.0) * dp / v0
    assert dp <= tol * pscale, 'pos3d off by %.3e rel (bar %.1e)' % (dp / pscale, tol)
    assert d_norm <= tol and d_z <= tol, '|spin| / spin_z off by %.3e / %.3e rel (bar %.1e)' % (d_norm, d_z, tol)
    assert d_xy <= tol + turn, 'local spin x/y off by %.3e rel (bar %.1e + frame turn %.3e)' % (d_xy, tol, turn)
    return dp / pscale, d_norm, d_z, d_xy


def hard_cases(g):
    """Iterate the near-tie fixture tests/golden/wasb_hard.npz (tools/make_goldens.py gen_hard): yields
    (key, weight seed, weight noise, clip seed, sigma, gain, n_frames, h, w) per clip."""
    for si in range(int(g['n_sets'][0])):
        for ci in range(int(g['n_clips'][si])):
            key = 'set%d/clip%d' % (si, ci)
            wseed, cseed, nf, h, w = [int(v) for v in g[key + '/meta']]
            weps, sigma, gain = [float(v) for v in g[key + '/params']]
            yield key, wseed, weps, cseed, sigma, gain, nf, h, w


def hard_frames(g, key, cseed, sigma, gain, nf, h, w):
    """The clip of a hard case, regenerated from its seed; the fixture's sha256 proves it is the one the reference ran on."""
    import hashlib
    from upliftingtabletennis_amd import synth
    frames, _ = synth.hard_clip(nf, h, w, seed=cseed, sigma=sigma, gain=gain)
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g[key + '/frames_sha256']), 'regenerated frames differ from the fixture\'s (%s)' % key
    return frames


def ragged_trajectories(b, t, pad):
    """synth.synth_trajectories(b, t, pad) with every second trajectory cut to half its length (ragged masks within a batch)."""
    from upliftingtabletennis_amd import synth
    ball, table, mask, times = synth.synth_trajectories(b, t, seed=100 + t, pad=pad)
    mask[1::2, max(2, t // 2):] = 0.0
    return ball, table, mask, times
