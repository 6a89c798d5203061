"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's camera calibration and re-projection:
`inference/utils.py:312-329` (`calibrate_camera`), `dataprocessing/regress_cameramatrices.py:38-231` (DLT start, RANSAC over
6-point subsets, BFGS refinement of (fx, fy, t, euler angles)), `dataprocessing/my_dlt.py:5-162`, `interface.py:291-312`
(`reproject`), with the same SciPy calls in the same order (`scipy.optimize.minimize(method='BFGS')`, `scipy.linalg.svd / rq`,
`scipy.spatial.transform.Rotation`).

Pinned: tests/golden/calib.npz was produced by the reference's own `calibrate_camera` (tools/make_goldens.py:gen_calib) and
this restatement reproduces it bit for bit on the build container (tests/test_cabi.py::test_calibration_oracle_matches_reference_and_host_glue).  The product path is the
device solver csrc/calib.hip behind upliftingtabletennis_amd/calib.py; nothing there imports this file.
"""
import numpy as np

WIDTH, HEIGHT = 1920, 1080                       # balldetection/helper_balldetection.py (imported by inference/utils.py:21)
KEYPOINT_VISIBLE = 1
TABLE_HEIGHT, TABLE_WIDTH, TABLE_LENGTH = 0.76, 1.525, 2.74
# the 13 table keypoints in world coordinates (uplifting/helper.py:36-50)
TABLE_POINTS = np.array([
    [-TABLE_LENGTH / 2, TABLE_WIDTH / 2, TABLE_HEIGHT], [-TABLE_LENGTH / 2, -TABLE_WIDTH / 2, TABLE_HEIGHT],
    [0.0, TABLE_WIDTH / 2, TABLE_HEIGHT], [0.0, -TABLE_WIDTH / 2, TABLE_HEIGHT],
    [TABLE_LENGTH / 2, TABLE_WIDTH / 2, TABLE_HEIGHT], [TABLE_LENGTH / 2, -TABLE_WIDTH / 2, TABLE_HEIGHT],
    [0.0, TABLE_WIDTH / 2 + 0.1525, TABLE_HEIGHT], [0.0, -(TABLE_WIDTH / 2 + 0.1525), TABLE_HEIGHT],
    [0.0, 0.0, TABLE_HEIGHT],
    [0.0, TABLE_WIDTH / 2 + 0.1525, TABLE_HEIGHT + 0.1525], [0.0, -(TABLE_WIDTH / 2 + 0.1525), TABLE_HEIGHT + 0.1525],
    [-TABLE_LENGTH / 2, 0, TABLE_HEIGHT], [TABLE_LENGTH / 2, 0, TABLE_HEIGHT],
], dtype=np.float64)
POINTS3D = {i + 1: TABLE_POINTS[i] for i in range(13)}          # regress_cameramatrices.py:21-35 (keys 1..13)


def world2cam(r_world, Mext):
    """uplifting/helper.py:168-204 for one point (3,) or a list of points (T,3) and one 4x4 extrinsic matrix
    (np.einsum with the reference's subscripts, so the summation order is the same)."""
    r_world = np.asarray(r_world, dtype=np.float64)
    Mext = np.asarray(Mext, dtype=np.float64)
    if Mext.ndim != 2 or r_world.ndim not in (1, 2):
        raise ValueError('Shape not supported.')
    hom = np.concatenate([r_world, np.ones(r_world.shape[:-1] + (1,))], axis=-1)
    if r_world.ndim == 1:
        cam = np.einsum('ij,j->i', Mext, hom)
        return cam[:3] / cam[3]
    cam = np.einsum('ij,bj->bi', Mext, hom)
    return cam[:, :3] / cam[:, 3:4]


def cam2img(r_cam, Mint):
    """uplifting/helper.py cam2img: uses Mint[:3,:3] only."""
    r_cam = np.asarray(r_cam, dtype=np.float64)
    Mint = np.asarray(Mint, dtype=np.float64)
    if Mint.ndim != 2 or r_cam.ndim not in (1, 2):
        raise ValueError('Shape not supported.')
    if r_cam.ndim == 1:
        img = np.einsum('ij,j->i', Mint[:3, :3], r_cam)
        return img[:2] / img[2]
    img = np.einsum('ij,bj->bi', Mint[:3, :3], r_cam)
    return img[:, :2] / img[:, 2:3]


# ------------------------------------------------------------------ DLT start (my_dlt.py)
def _normalize_points(points):
    mean, std = np.mean(points, axis=0), np.std(points, axis=0)
    std[std == 0] = 1e-10
    dim = points.shape[1]
    T = np.eye(dim + 1)
    T[:dim, :dim] = np.diag(1.0 / std)
    T[:dim, -1] = -mean / std
    hom = np.hstack((points, np.ones((points.shape[0], 1))))
    return (T @ hom.T).T[:, :dim], T


def dlt_calib(points_3d, points_2d):
    """my_dlt.py:40-162: normalised DLT (SVD null vector), RQ decomposition, sign fixes -> Mint (3,3), Mext (3,4)."""
    from scipy.linalg import rq, svd
    assert points_3d.shape[0] == points_2d.shape[0] and points_3d.shape[1] == 3 and points_2d.shape[1] == 2
    p3, T3 = _normalize_points(points_3d)
    p2, T2 = _normalize_points(points_2d)
    A = np.zeros((len(p3) * 2, 12))
    for i in range(len(p3)):
        X, Y, Z = p3[i]
        x, y = p2[i]
        A[2 * i] = [-X, -Y, -Z, -1, 0, 0, 0, 0, x * X, x * Y, x * Z, x]
        A[2 * i + 1] = [0, 0, 0, 0, -X, -Y, -Z, -1, y * X, y * Y, y * Z, y]
    _, _, Vt = svd(A)
    P = np.linalg.inv(T2) @ Vt[-1, :].reshape(3, 4) @ T3
    if P[2, 3] != 0:
        P /= P[2, 3]
    else:
        P /= np.linalg.norm(P)
    K, Rm = rq(P[:, :3])
    signs = np.diag(np.sign(np.diag(K)))
    K, Rm = K @ signs, signs @ Rm
    if K[2, 2] != 0:
        K /= K[2, 2]
    else:
        raise ValueError("Intrinsic matrix K has K[2,2] close to zero, indicating a degenerate camera.")
    if np.linalg.det(Rm) < 0:
        Rm[:, 2] *= -1
    t = np.linalg.solve(K, P[:, 3])
    return K, np.hstack((Rm, t.reshape(3, 1)))


# ------------------------------------------------------------------ refinement (regress_cameramatrices.py:38-126)
def _matrices(x, resolution):
    from scipy.spatial.transform import Rotation
    w, h = resolution
    fx, fy, tx, ty, tz, a, b, c = x
    Mint = np.array([[fx, 0, w // 2, 0], [0, fy, h // 2, 0], [0, 0, 1, 0]])
    rot = Rotation.from_euler('xyz', [a, b, c], degrees=False).as_matrix()
    Mext = np.array([[rot[0, 0], rot[0, 1], rot[0, 2], tx], [rot[1, 0], rot[1, 1], rot[1, 2], ty],
                     [rot[2, 0], rot[2, 1], rot[2, 2], tz], [0, 0, 0, 1]])
    return Mint, Mext


def regress_cameramatrices(resolution, points2d, points3d=POINTS3D, startmatrices=None, use_lm=False):
    """Minimise the summed re-projection distance over (fx, fy, tx, ty, tz, euler xyz) from `startmatrices`."""
    from scipy.optimize import least_squares, minimize
    from scipy.spatial.transform import Rotation
    p2 = np.array([pt for _, pt in points2d])
    p3 = np.array([points3d[k] for k, _ in points2d])

    def residuals(x):
        Mint, Mext = _matrices(x, resolution)
        return np.sqrt(np.sum(np.square(cam2img(world2cam(p3, Mext), Mint) - p2), axis=1))

    assert startmatrices is not None, 'startmatrices must be provided'
    Mint, Mext = startmatrices
    try:
        angles = Rotation.from_matrix(Mext[:3, :3]).as_euler('xyz', degrees=False)
    except ValueError:
        angles = np.array([0, 0, 0])
    x0 = np.array([Mint[0, 0], Mint[1, 1], Mext[0, 3], Mext[1, 3], Mext[2, 3], angles[0], angles[1], angles[2]])
    x0[5:] = np.mod((x0[5:] + np.pi), (2 * np.pi)) - np.pi
    res = least_squares(residuals, x0, method='lm') if use_lm else minimize(lambda x: np.sum(residuals(x)), x0, method='BFGS')
    return _matrices(res.x, resolution)


def regress_cameramatrices_ransac(resolution, points2d, startmatrices=None, use_lm=False):
    """regress_cameramatrices.py:129-188: 100 subsets of 6 points (keys 10 and 11 always in), inlier threshold 3.5 px,
    first-best subset wins, final refinement on its inliers."""
    max_iterations, num_points, threshold = 100, 6, 3.5
    fixed = [10, 11]
    fixed2d = [(int(k), p) for k, p in points2d if k in fixed]
    best_inliers, best = None, None
    rnd = np.random.default_rng(seed=42)
    for _ in range(max_iterations):
        sampled = rnd.choice([int(k) for k, _ in points2d if k not in fixed], size=num_points - len(fixed), replace=False)
        sampled = [int(s) for s in sampled]
        subset2d = [*fixed2d, *[(int(k), p) for k, p in points2d if k in sampled]]
        subset3d = {k: POINTS3D[k] for k in fixed + sampled}
        Mint, Mext = regress_cameramatrices(resolution, subset2d, subset3d, startmatrices=startmatrices, use_lm=use_lm)
        inliers = [(k, p) for k, p in points2d if np.linalg.norm(cam2img(world2cam(POINTS3D[k], Mext), Mint) - p) < threshold]
        if best_inliers is None or len(inliers) > len(best_inliers):
            best_inliers, best = inliers, (Mint, Mext)
    if best is None:
        raise ValueError("RANSAC failed to find a valid model.")
    subset3d = {k: POINTS3D[k] for k, _ in best_inliers}
    Mint, Mext = regress_cameramatrices(resolution, best_inliers, subset3d, startmatrices=best, use_lm=use_lm)
    return Mint, Mext, len(best_inliers)


def calc_cameramatrices(keypoints_dict, resolution, use_lm=False, use_ransac=False):
    """regress_cameramatrices.py:199-231: DLT on all points, then RANSAC or plain refinement."""
    assert len(keypoints_dict.keys()) >= 6, 'not enough points for DLT'
    points2d = [(k, p) for k, pts in keypoints_dict.items() for p in pts]
    Mint, Mext = dlt_calib(np.array([POINTS3D[k] for k, _ in points2d]), np.array([p for _, p in points2d]))
    if use_ransac:
        return regress_cameramatrices_ransac(resolution, points2d, startmatrices=(Mint, Mext), use_lm=use_lm)
    Mint, Mext = regress_cameramatrices(resolution, points2d, POINTS3D, startmatrices=(Mint, Mext), use_lm=use_lm)
    return Mint, Mext, len(points2d)


def calibrate_camera(table_coords):
    """inference/utils.py:312-329: (13,3) keypoints (x, y, visibility) -> M_int (3,4), M_ext (4,4) (the shapes the
    reference really returns; its docstring says (3,3) / (3,4))."""
    keypoints = {}
    for i, (x, y, v) in enumerate(np.asarray(table_coords)):
        if v == KEYPOINT_VISIBLE:
            keypoints[i + 1] = [(x, y)]
    M_int, M_ext, _ = calc_cameramatrices(keypoints, resolution=(WIDTH, HEIGHT), use_lm=False, use_ransac=True)
    return M_int, M_ext


def reproject(positions_3d, Mint, Mext):
    """interface.py:301-312."""
    return cam2img(world2cam(positions_3d, Mext), Mint)
