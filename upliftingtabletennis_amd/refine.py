"""a3/a4: drop-ins for ``extract_position_torch_gaussian`` (ball: balldetection/helper_balldetection.py:29-110,
called at inference/utils.py:59; table: tabledetection/helper_tabledetection.py:50-156, called at
interface.py:116).  Same arguments, same numpy float64 result; the work runs in csrc/refine.hip."""
import numpy as np
import torch

from . import _lib

HEIGHT, WIDTH = 1080, 1920          # helper_balldetection.py:12
BALL_VISIBLE, BALL_INVISIBLE = 1, 0
KEYPOINT_VISIBLE, KEYPOINT_INVISIBLE = 1, 0


def refine_device(heatmaps, image_width, image_height, variant):
    """(N,H,W) float32 device tensor -> (xyv (N,3) float64, argmax (N,) int64, windows (N,9) float32), all on device."""
    _lib.require_gpu()
    lib = _lib.load()
    heatmaps = heatmaps.contiguous()
    n, h, w = heatmaps.shape
    dev = heatmaps.device
    out = torch.empty((n, 3), dtype=torch.float64, device=dev)
    idx = torch.empty((n,), dtype=torch.int64, device=dev)
    win = torch.empty((n, 9), dtype=torch.float32, device=dev)
    if n == 0:              # nothing to refine (the reference's per-heatmap loop simply does not run)
        return out, idx, win
    ws_bytes = lib.ttup_refine_workspace_bytes(n, h, w)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ttup_refine(_lib.ptr(heatmaps), n, h, w, int(image_width), int(image_height), variant,
                                   _lib.ptr(out), _lib.ptr(idx), _lib.ptr(win), _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
    return out, idx, win


def refine_windows_device(idx, win, h, w, image_width, image_height, variant):
    """Second half only (peaks already found by the CNN's fused epilogue)."""
    lib = _lib.load()
    out = torch.empty((idx.shape[0], 3), dtype=torch.float64, device=idx.device)
    if idx.shape[0] == 0:
        return out
    with torch.cuda.device(idx.device):
        _lib.check(lib.ttup_refine_windows(_lib.ptr(idx), _lib.ptr(win), idx.shape[0], h, w, int(image_width), int(image_height),
                                           variant, _lib.ptr(out), _lib.stream_ptr()))
    return out


def _to_device(heatmaps):
    _lib.require_gpu()
    if not isinstance(heatmaps, torch.Tensor):
        heatmaps = torch.as_tensor(np.asarray(heatmaps))
    return heatmaps.to('cuda' if not heatmaps.is_cuda else heatmaps.device, torch.float32)


def extract_position_ball(heatmaps, image_width, image_height):
    """Ball variant -> np.float64 (B,3) [x, y, visibility]."""
    if len(heatmaps.shape) == 4:
        heatmaps = heatmaps.squeeze(1)
    if len(heatmaps.shape) != 3:
        raise ValueError("Heatmaps must have shape (B, H, W)")
    out, _, _ = refine_device(_to_device(heatmaps), image_width, image_height, _lib.REFINE_BALL)
    return out.cpu().numpy()


def extract_position_table(heatmaps, image_width, image_height, threshold=0.1):
    """Table variant -> np.float64 (B,C,3).  ``threshold`` is accepted and, as in the reference
    (helper_tabledetection.py:106-110 then :142), has no effect on the result."""
    if len(heatmaps.shape) != 4:
        raise ValueError("Heatmaps must have shape (B, C, H, W)")
    b, c, h, w = heatmaps.shape
    out, _, _ = refine_device(_to_device(heatmaps).reshape(b * c, h, w), image_width, image_height, _lib.REFINE_TABLE)
    return out.cpu().numpy().reshape(b, c, 3)
