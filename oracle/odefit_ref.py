"""TEST INFRASTRUCTURE ONLY -- numpy/scipy statement of the drag + Magnus ODE fit (csrc/odefit.hip).

PARITY UNPINNED: the reference has no ODE fit (its uplift is the transformer, uplifting/model.py), so there is
nothing to pin this against.  The constants restate the MuJoCo scene of syntheticdataset/helper.py:79-117 (40 mm / 2.7 g
sphere, rho 1.225, mu 1.8e-5, fluidcoef "0.235 0.25 0 1 1") the way oracle/trajgen_ref.py does; validation is by
self-consistency: RK4 order, recovery of planted parameters, agreement of the device solver with scipy.optimize.least_squares.
"""
import numpy as np

R_BALL, M_BALL, RHO, MU_AIR, GRAV = 0.02, 0.0027, 1.225, 0.000018, 9.81
C_BLUNT, C_MAGNUS = 0.235, 1.0
VOL, AREA = 4.0 / 3.0 * np.pi * R_BALL ** 3, np.pi * R_BALL ** 2
I_BALL = 0.4 * M_BALL * R_BALL ** 2
K_STOKES, K_QUAD = 6.0 * np.pi * MU_AIR * R_BALL, RHO * C_BLUNT * AREA
K_MAG = C_MAGNUS * RHO * VOL - 0.5 * RHO * VOL
K_SPIN = 8.0 * np.pi * MU_AIR * R_BALL ** 3 / I_BALL


def accel(v, w):
    speed = np.sqrt(np.dot(v, v))
    a = -(K_STOKES + K_QUAD * speed) / M_BALL * v + K_MAG / M_BALL * np.cross(w, v)
    a[2] -= GRAV
    return a


def rk4(s, h):
    r, v, w = s[0:3], s[3:6], s[6:9]
    a1, l1 = accel(v, w), -K_SPIN * w
    v2, w2 = v + 0.5 * h * a1, w + 0.5 * h * l1
    a2, l2 = accel(v2, w2), -K_SPIN * w2
    v3, w3 = v + 0.5 * h * a2, w + 0.5 * h * l2
    a3, l3 = accel(v3, w3), -K_SPIN * w3
    v4, w4 = v + h * a3, w + h * l3
    a4, l4 = accel(v4, w4), -K_SPIN * w4
    return np.concatenate([r + h / 6.0 * (v + 2 * v2 + 2 * v3 + v4), v + h / 6.0 * (a1 + 2 * a2 + 2 * a3 + a4), w + h / 6.0 * (l1 + 2 * l2 + 2 * l3 + l4)])


def substeps(dt, h_max):
    return max(1, int(np.ceil(dt / h_max - 1e-9)))


def integrate(p, times, h_max):
    """p (9,), times (T,) -> positions (T,3) at the time stamps."""
    s = np.array(p, dtype=np.float64)
    out = np.zeros((len(times), 3))
    tprev = times[0]
    for i, t in enumerate(times):
        dt = t - tprev
        if i > 0 and dt > 0:
            n = substeps(dt, h_max)
            for _ in range(n):
                s = rk4(s, dt / n)
            tprev = t
        out[i] = s[:3]
    return out


def project(cam21, pos):
    ex, k = np.asarray(cam21[:12]).reshape(3, 4), np.asarray(cam21[12:]).reshape(3, 3)
    pc = pos @ ex[:, :3].T + ex[:, 3]
    q = pc @ k.T
    return q[:, :2] / q[:, 2:3]


def fit(obs, times, cam21, init, h_max, mask=None):
    """scipy Levenberg-Marquardt on the same residuals (finite-difference Jacobian) -> parameters (9,)."""
    from scipy.optimize import least_squares
    m = np.ones(len(times), bool) if mask is None else np.asarray(mask) != 0

    def res(p):
        return (project(cam21, integrate(p, times, h_max)) - obs)[m].ravel()
    return least_squares(res, np.asarray(init, np.float64), method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale=[1, 1, 1, 10, 10, 10, 100, 100, 100]).x
