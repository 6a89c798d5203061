"""f4: camera calibration on the MI355X (csrc/calib.hip) through the C-ABI, against the reference's own results
(tests/golden/calib.npz, produced by the reference's `calibrate_camera`).  /root/reference is never read here.

The device minimises the reference's objective (sum of re-projection distances) over the reference's subsets with its own
solver; SciPy's BFGS with finite-difference gradients stops on that non-smooth objective where its line search gives up, so
the two agree to about a pixel, not to the last bit.  Measured on the three golden cameras: identical inlier sets, device
objective 0.1-1.9 % BELOW the reference's (5.8127 vs 5.9221, 5.4582 vs 5.4631, 8.0564 vs 8.0749 px), re-projections within
1.01 / 0.61 / 0.20 px, focal lengths within 0.13 %."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu
if has_gpu():
    from upliftingtabletennis_amd import calib

TABLE = None


def _table():
    from upliftingtabletennis_amd import synth
    return synth.TABLE_POINTS


def _objective(kp, Mint, Mext):
    vis = kp[:, 2] == 1
    return np.linalg.norm(calib.reproject(_table()[vis], Mint, Mext) - kp[vis, :2], axis=1)


def test_device_calibration_matches_reference_goldens(golden):
    g = golden('calib.npz')
    n = int(g['n'][0])
    kps = np.stack([g['calib/%d/keypoints' % ci] for ci in range(n)])
    mint, mext, ninl = calib.calibrate_cameras(kps)
    for ci in range(n):
        rMint, rMext = g['calib/%d/Mint' % ci], g['calib/%d/Mext' % ci]
        kp = kps[ci]
        e_dev, e_ref = _objective(kp, mint[ci], mext[ci]), _objective(kp, rMint, rMext)
        # same inlier set as the reference's final model, and an objective on it that is at least as low (the device solver
        # runs to the minimum; SciPy stops a little before it)
        inl_dev, inl_ref = e_dev < 3.5, e_ref < 3.5
        assert np.array_equal(inl_dev, inl_ref) and int(inl_ref.sum()) == int(ninl[ci])
        assert e_dev[inl_ref].sum() <= e_ref[inl_ref].sum() * (1 + 1e-9)
        # the two cameras re-project the table and the golden test points to the same pixels
        pts = np.concatenate([_table(), g['calib/%d/points' % ci]])
        d = np.linalg.norm(calib.reproject(pts, mint[ci], mext[ci]) - calib.reproject(pts, rMint, rMext), axis=1)
        print('camera %d: %d inliers, objective device %.6f reference %.6f px, max re-projection difference %.2e px, fx %.3f vs %.3f'
              % (ci, ninl[ci], e_dev[inl_ref].sum(), e_ref[inl_ref].sum(), d.max(), mint[ci][0, 0], rMint[0, 0]))
        assert d.max() < 1.5          # sanity only: two near-minima of a non-smooth objective; the claim under test is the ordering above (calibration)
        # the device result is a minimum of the reference's objective: SciPy's BFGS (the reference's optimiser, through the pinned
        # oracle) started AT the device matrices does not get below it
        from oracle import calib_ref
        vis_idx = np.nonzero(kp[:, 2] == 1)[0]
        inl = [(int(k) + 1, tuple(kp[k, :2])) for j, k in enumerate(vis_idx) if inl_ref[j]]
        Mi2, Me2 = calib_ref.regress_cameramatrices((1920, 1080), inl, calib_ref.POINTS3D, startmatrices=(mint[ci], mext[ci]))
        e_bfgs = _objective(kp, Mi2, Me2)
        assert e_bfgs[inl_ref].sum() >= e_dev[inl_ref].sum() - 1e-6
        assert abs(mint[ci][0, 0] - rMint[0, 0]) < 5e-3 * rMint[0, 0] and abs(mint[ci][1, 1] - rMint[1, 1]) < 5e-3 * rMint[1, 1]
        assert mint[ci].shape == (3, 4) and mext[ci].shape == (4, 4) and mint[ci][0, 2] == 960 and mint[ci][1, 2] == 540
        assert np.allclose(mext[ci][:3, :3] @ mext[ci][:3, :3].T, np.eye(3), atol=1e-12) and np.array_equal(mext[ci][3], [0, 0, 0, 1])
    # single-camera entry = the hub surface
    Mi, Me = calib.calibrate_camera(kps[1])
    assert np.array_equal(Mi, mint[1]) and np.array_equal(Me, mext[1])


def test_device_calibration_recovers_a_planted_camera_and_rejects_an_outlier():
    from upliftingtabletennis_amd import synth
    rng = np.random.default_rng(4)
    R, c, f = synth._random_camera(rng)
    Mext = np.eye(4); Mext[:3, :3] = R; Mext[:3, 3] = -R @ c
    Mint = np.array([[f, 0, 960.0, 0], [0, f, 540.0, 0], [0, 0, 1, 0]])
    kp = np.concatenate([calib.reproject(synth.TABLE_POINTS, Mint, Mext), np.ones((13, 1))], axis=1)
    mi, me = calib.calibrate_camera(kp)                       # noiseless: exact recovery
    assert abs(mi[0, 0] - f) < 1e-6 * f and abs(mi[1, 1] - f) < 1e-6 * f and np.abs(me - Mext).max() < 1e-7
    bad = kp.copy(); bad[2, :2] += 40.0; bad[5, 2] = 0        # one gross outlier, one invisible keypoint
    mints, mexts, ninl = calib.calibrate_cameras(np.stack([kp, bad]))
    assert ninl.tolist() == [13, 11]
    assert abs(mints[1][0, 0] - f) < 1e-6 * f and np.abs(mexts[1] - Mext).max() < 1e-7
    with pytest.raises(AssertionError):
        calib.calibrate_camera(np.concatenate([kp[:, :2], np.zeros((13, 1))], axis=1))
    with pytest.raises(ValueError):
        calib.calibrate_cameras(np.zeros((2, 12, 3)))


def test_device_calibration_against_the_reference_on_64_cameras(golden):
    """The distribution behind the bar: 64 seeded synthetic cameras -- poses / focal lengths from the ranges the
    reference trains its uplift net on, pixel noise 0.3 / 0.6 / 1.0 px, every fourth with a gross outlier, every fifth with an invisible
    keypoint -- calibrated by the REFERENCE's `calibrate_camera` (tests/golden/calib64.npz, tools/make_goldens.py:gen_calib64; the
    true cameras are in the fixture too) and by the device solver from the same 100 RANSAC subsets per camera.

    Measured (round 6, tools/calib64_probe.py): on 61 cameras both find the same inlier set; there the 13 table points re-project
    within 0.046 px (median) / 0.46 px (p90) / 1.71 px (max) of the reference's, the device objective is at or below the reference's
    on every one of them (ratio 0.9989 median, 1.0000 max), and |fx| differs by 0.03 % (median) / 2.1 % (max: focal length and distance
    trade off; on one camera the reference returns fx < 0 with a mirrored axis -- the same projection).  On the
    other three (7, 31, 63) the REFERENCE's BFGS has diverged -- 3, 4 and 10 inliers, re-projections 323 / 397 / 62 px off the true
    camera, a negative focal length on one -- while the device keeps 12 inliers within 1.8 px of the truth.  Against the TRUE cameras:
    device 0.89 px median / 3.1 px max, reference 0.98 px median.
    Bars: same inlier set wherever the reference kept >= 11 inliers; re-projection difference <= 3.5 px there (2 x the measured
    maximum); where the reference diverged the device must hold >= 11 inliers within 3.5 px of the true camera.  Two minimisers of a
    non-smooth objective (sum of distances; SciPy's finite-difference BFGS stops where its line search gives up, the device's IRLS-LM
    runs on): an ordering and a distribution, not bit parity."""
    g = golden('calib64.npz')
    n = int(g['n'][0])
    kps = np.stack([g['calib64/%d/keypoints' % ci] for ci in range(n)])
    mint, mext, ninl = calib.calibrate_cameras(kps)
    d_same, ratio_same, f_same, truth_dev, truth_ref, diverged = [], [], [], [], [], []
    for ci in range(n):
        rMint, rMext = g['calib64/%d/Mint' % ci], g['calib64/%d/Mext' % ci]
        tMint, tMext = g['calib64/%d/Mint_true' % ci], g['calib64/%d/Mext_true' % ci]
        e_dev, e_ref = _objective(kps[ci], mint[ci], mext[ci]), _objective(kps[ci], rMint, rMext)
        inl_dev, inl_ref = e_dev < 3.5, e_ref < 3.5
        p_dev, p_ref, p_true = (calib.reproject(_table(), mi, me) for mi, me in ((mint[ci], mext[ci]), (rMint, rMext), (tMint, tMext)))
        truth_dev.append(float(np.linalg.norm(p_dev - p_true, axis=1).max()))
        truth_ref.append(float(np.linalg.norm(p_ref - p_true, axis=1).max()))
        if int(inl_ref.sum()) < 11:          # the reference's optimiser left the basin (an outlier / invisible key never costs more than two)
            diverged.append(ci)
            assert int(inl_dev.sum()) >= 11 and truth_dev[-1] <= 3.5, (ci, int(inl_dev.sum()), truth_dev[-1], int(inl_ref.sum()), truth_ref[-1])
            continue
        assert np.array_equal(inl_dev, inl_ref) and int(inl_ref.sum()) == int(ninl[ci]), (ci, inl_dev, inl_ref)
        d_same.append(float(np.linalg.norm(p_dev - p_ref, axis=1).max()))
        ratio_same.append(float(e_dev[inl_ref].sum() / e_ref[inl_ref].sum()))
        f_same.append(float(abs(abs(mint[ci][0, 0]) - abs(rMint[0, 0])) / abs(rMint[0, 0])))          # (the reference returns fx < 0 with a mirrored axis on one camera: the same projection)
    d_same, ratio_same, f_same = np.array(d_same), np.array(ratio_same), np.array(f_same)
    q = lambda a: [round(float(np.percentile(a, p)), 4) for p in (50, 90, 100)]
    print('\n64 cameras, reference diverged on %s; on the other %d [median, p90, max]: re-projection difference %s px, objective device / reference %s, '
          '|fx - fx_ref| / fx_ref %s; against the true cameras: device %s px, reference %s px'
          % (diverged, len(d_same), q(d_same), q(ratio_same), q(f_same), q(np.array(truth_dev)), q(np.array(truth_ref))))
    assert len(diverged) <= 4
    assert d_same.max() <= 3.5 and f_same.max() <= 0.05
    assert ratio_same.max() <= 1 + 1e-6, ratio_same[ratio_same > 1]          # the device never ends above the reference on the reference's own inliers
    assert np.median(truth_dev) <= np.median(truth_ref) + 0.05 and max(truth_dev) <= 3.5
