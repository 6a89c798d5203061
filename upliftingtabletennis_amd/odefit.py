"""g1 (extension, no reference counterpart): batched drag + Magnus ODE fit of a 2-D ball track -- the "RK4 + Jacobian /
Gauss-Newton" uplift that BASELINE.json's north_star names.  The reference's uplift is the transformer (``uplift.py``);
nothing on the drop-in surface calls this module.  Kernel: csrc/odefit.hip; numpy statement of the same model and solver
target: oracle/odefit_ref.py (test infrastructure)."""
import ctypes

import numpy as np
import torch

from . import _lib

H_MAX = 2e-3          # largest RK4 step [s]


def _cam21(mext, mint):
    """Mext (.., 4,4) or (..,3,4), Mint (..,3,3) -> (.., 21) float64: rows 0..2 of Mext, then Mint."""
    mext, mint = np.asarray(mext, np.float64), np.asarray(mint, np.float64)
    return np.concatenate([mext[..., :3, :4].reshape(mext.shape[:-2] + (12,)), mint.reshape(mint.shape[:-2] + (9,))], axis=-1)


def _dev(a, device):
    return torch.as_tensor(np.asarray(a, np.float64) if not isinstance(a, torch.Tensor) else a).to(device, torch.float64).contiguous()


def integrate(params, times, cam=None, h_max=H_MAX, device='cuda'):
    """Forward model: params (B,9), times (B,T) -> (pos3d (B,T,3), pixels (B,T,2) or None); device float64 tensors."""
    _lib.require_gpu()
    lib = _lib.load()
    params, times = _dev(params, device), _dev(times, device)
    b, t = times.shape
    pos = torch.empty((b, t, 3), dtype=torch.float64, device=params.device)
    px = torch.empty((b, t, 2), dtype=torch.float64, device=params.device) if cam is not None else None
    camt = _dev(cam, device) if cam is not None else None
    per = 1 if camt is not None and camt.dim() == 2 and camt.shape[0] == b and b > 1 else 0
    with torch.cuda.device(params.device):
        _lib.check(lib.ttup_odefit_integrate(_lib.ptr(params), _lib.ptr(times), _lib.ptr(camt), per, b, t, float(h_max), _lib.ptr(pos), _lib.ptr(px), _lib.stream_ptr()))
    return pos, px


def fit(ball_xy, times, cam, init, mask=None, h_max=H_MAX, max_iter=80, tol=1e-14, device='cuda'):
    """Levenberg-Marquardt fit of (r0, v0, w0) to pixel tracks.  ball_xy (B,T,2) px, times (B,T) s, cam (21,) or (B,21)
    (see `_cam21`), init (B,9).  Returns dict of device tensors: params (B,9), pos3d (B,T,3), cost (B) mean squared
    reprojection error [px^2], iters (B)."""
    _lib.require_gpu()
    lib = _lib.load()
    obs, times, camt, init = _dev(ball_xy, device), _dev(times, device), _dev(cam, device), _dev(init, device)
    b, t, _ = obs.shape
    if times.shape != (b, t) or init.shape != (b, 9):
        raise ValueError('inconsistent input shapes')
    per = 1 if camt.dim() == 2 and camt.shape[0] == b and b > 1 else 0
    if camt.numel() != (b if per else 1) * 21:
        raise ValueError('cam must hold 21 numbers (Mext rows 0..2, Mint) once or per trajectory')
    maskt = _dev(mask, device) if mask is not None else None
    dev = obs.device
    params = torch.empty((b, 9), dtype=torch.float64, device=dev)
    pos = torch.empty((b, t, 3), dtype=torch.float64, device=dev)
    cost = torch.empty((b,), dtype=torch.float64, device=dev)
    iters = torch.empty((b,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ttup_odefit_forward(_lib.ptr(obs), _lib.ptr(times), _lib.ptr(maskt), _lib.ptr(camt), per, _lib.ptr(init), b, t, float(h_max),
                                           int(max_iter), float(tol), _lib.ptr(params), _lib.ptr(pos), _lib.ptr(cost), _lib.ptr(iters), _lib.stream_ptr()))
    return {'params': params, 'pos3d': pos, 'cost': cost, 'iters': iters}


def synth_arcs(b, t, fps=120.0, seed=0):
    """Planted contact-free arcs over the table seen by the generator's fixed camera: (params (b,9), times (b,t), cam (21,))."""
    from . import trajgen
    rng = np.random.default_rng(seed)
    p = np.zeros((b, 9))
    p[:, 0] = rng.uniform(-1.6, -0.8, b); p[:, 1] = rng.uniform(-0.5, 0.5, b); p[:, 2] = rng.uniform(0.9, 1.3, b)
    p[:, 3] = rng.uniform(2.5, 5.0, b); p[:, 4] = rng.uniform(-0.6, 0.6, b); p[:, 5] = rng.uniform(1.0, 3.0, b)
    p[:, 6:9] = rng.uniform(-150.0, 150.0, (b, 3))
    times = np.tile(np.arange(t) / fps, (b, 1))
    ex, mint = trajgen.camera_matrices()
    return p, times, _cam21(ex, mint)


def bench(device, b=10000, t=120):
    """BASELINE config 3's shape (B trajectories x T steps) through the ODE fit: planted arcs, noiseless pixels, start 5 cm /
    0.5 m/s / 20 rad/s off.  Returns the dict bench.py prints."""
    import time
    p, times, cam = synth_arcs(b, t, seed=1)
    _, px = integrate(p, times, cam, device=device)
    rng = np.random.default_rng(2)
    init = p + np.concatenate([rng.normal(0, 0.05, (b, 3)), rng.normal(0, 0.5, (b, 3)), rng.normal(0, 20.0, (b, 3))], axis=1)
    fit(px[:64], times[:64], cam, init[:64], device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fit(px, times, cam, init, device=device)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = out['params'].cpu().numpy()
    rel = (np.abs(got - p) / np.maximum(np.abs(p), 1.0)).max(1)
    ok = rel < 1e-6              # monocular ambiguity: a few long arcs end in a local minimum (sub-0.3-px cost, different spin)
    err = np.abs(got - p)[ok]
    return {'value': round(b / dt, 1), 'unit': 'trajectories/s', 'seconds': round(dt, 4),
            'config': 'north_star extension (no reference counterpart): RK4 + forward-mode Jacobian + Levenberg-Marquardt fit of drag+Magnus flight, B=%d, T=%d' % (b, t),
            'mean_accepted_steps': round(float(out['iters'].float().mean().item()), 2),
            'recovered_fraction': round(float(ok.mean()), 4), 'recovered_means': 'planted (r0, v0, w0) back to 1e-6 relative from noiseless pixels, start 5 cm / 0.5 m/s / 20 rad/s off',
            'max_abs_error_r0_m': float(err[:, :3].max()), 'max_abs_error_v0_m_s': float(err[:, 3:6].max()), 'max_abs_error_w0_rad_s': float(err[:, 6:].max())}
