"""f4: camera calibration from the 13 table keypoints, and re-projection -- the hub surface's `calibrate_camera` /
`reproject` (reference interface.py:174-175, :291-312; inference/utils.py:312-329).

`calibrate_camera` runs on the MI355X (csrc/calib.hip, one workgroup per camera, one lane per RANSAC subset); the host only
draws the subsets -- with numpy's PCG64 stream seeded 42, exactly the draws of regress_cameramatrices.py:148-152, so that the
device examines the same 100 six-point subsets as the reference.  `calibrate_cameras` is the batched form (many frames /
cameras in one launch).  There is no CPU fallback."""
import numpy as np
import torch

from . import _lib

WIDTH, HEIGHT = 1920, 1080
KEYPOINT_VISIBLE = 1
N_SUBSETS, FIXED_KEYS = 100, (10, 11)


def ransac_subsets(visible_keys, n_subsets=N_SUBSETS):
    """Keys (1..13) of the non-fixed keypoints of every RANSAC subset, (n_subsets, 4) int32: the reference draws 4 of the visible
    keys other than 10 / 11 per iteration from np.random.default_rng(seed=42) (regress_cameramatrices.py:148-152)."""
    pool = [int(k) for k in visible_keys if k not in FIXED_KEYS]
    rnd = np.random.default_rng(seed=42)
    return np.array([rnd.choice(pool, size=4, replace=False) for _ in range(n_subsets)], dtype=np.int32)


def calibrate_cameras(keypoints, resolution=(WIDTH, HEIGHT), max_iter=300, device='cuda'):
    """keypoints (B,13,3) [x, y, visibility] px -> (Mint (B,3,4), Mext (B,4,4), n_inliers (B,)) numpy float64 / int32."""
    _lib.require_gpu()
    lib = _lib.load()
    kp = np.ascontiguousarray(np.asarray(keypoints, dtype=np.float64))
    if kp.ndim != 3 or kp.shape[1:] != (13, 3):
        raise ValueError('keypoints must have shape (B, 13, 3)')
    b = kp.shape[0]
    subsets = np.zeros((b, N_SUBSETS, 4), np.int32)
    for i in range(b):
        vis = [k + 1 for k in range(13) if kp[i, k, 2] == KEYPOINT_VISIBLE]
        assert len(vis) >= 6, 'not enough points for DLT'                                  # regress_cameramatrices.py:209
        subsets[i] = ransac_subsets(vis)          # raises ValueError like the reference when fewer than 4 non-fixed keys are visible
    dev = torch.device(device)
    kpt, sub = torch.from_numpy(kp).to(dev), torch.from_numpy(subsets).to(dev)
    mint = torch.empty((b, 3, 4), dtype=torch.float64, device=dev)
    mext = torch.empty((b, 4, 4), dtype=torch.float64, device=dev)
    ninl = torch.empty((b,), dtype=torch.int32, device=dev)
    status = torch.empty((b,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ttup_calib_forward(_lib.ptr(kpt), _lib.ptr(sub), b, N_SUBSETS, int(resolution[0]), int(resolution[1]), int(max_iter),
                                          _lib.ptr(mint), _lib.ptr(mext), _lib.ptr(ninl), _lib.ptr(status), None, _lib.stream_ptr()))
    st = status.cpu().numpy()
    if (st == -2).any():
        raise ValueError('degenerate camera: the DLT start has no valid intrinsic / rotation split')       # my_dlt.py:127-128
    if (st < 0).any():
        raise ValueError('RANSAC failed to find a valid model.')
    return mint.cpu().numpy(), mext.cpu().numpy(), ninl.cpu().numpy()


def calibrate_camera(table_coords):
    """(13,3) keypoints [x, y, visibility] -> M_int (3,4), M_ext (4,4) (the shapes the reference really returns)."""
    mint, mext, _ = calibrate_cameras(np.asarray(table_coords, dtype=np.float64)[None])
    return mint[0], mext[0]


def world2cam(r_world, Mext):
    """World -> camera coordinates with a 4x4 extrinsic matrix (uplifting/helper.py:168-204); (3,) or (T,3) points."""
    r_world, Mext = np.asarray(r_world, dtype=np.float64), np.asarray(Mext, dtype=np.float64)
    if Mext.ndim != 2 or r_world.ndim not in (1, 2):
        raise ValueError('Shape not supported.')
    hom = np.concatenate([r_world, np.ones(r_world.shape[:-1] + (1,))], axis=-1) @ Mext.T
    return hom[..., :3] / hom[..., 3:4]


def cam2img(r_cam, Mint):
    """Camera -> pixel coordinates with the left 3x3 block of the intrinsic matrix (uplifting/helper.py:137-166)."""
    r_cam, Mint = np.asarray(r_cam, dtype=np.float64), np.asarray(Mint, dtype=np.float64)
    if Mint.ndim != 2 or r_cam.ndim not in (1, 2):
        raise ValueError('Shape not supported.')
    img = r_cam @ Mint[:3, :3].T
    return img[..., :2] / img[..., 2:3]


def reproject(positions_3d, Mint, Mext):
    """interface.py:301-312: (N,3) world positions -> (N,2) pixel positions."""
    return cam2img(world2cam(positions_3d, Mext), Mint)
