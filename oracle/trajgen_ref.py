"""CPU restatement of the synthetic-trajectory generator (BASELINE config 5).  TEST INFRASTRUCTURE ONLY.

Reference: syntheticdataset/mujocosimulation.py (sampling :54-109, stepping/sampling loop :112-151, selection :152-219,
batching :222-238) and syntheticdataset/helper.py (constants :14-40, XML physics :81-114, camera :262-280, hit
counting :282-321).

What is pinned and what is not
  * `init_state`            bit-for-bit against the reference's `_init_simulation` (run here with a stand-in `mujoco`
                            module that only provides the MjModel/MjData containers) -- tests/golden/trajgen.npz
  * `count_hits`, `select`  against the reference's `_count_hits` / `find_valid_trajectories_worker` control flow, driven
                            by THIS file's integrator through the same stand-in -- tests/golden/trajgen.npz
  * `simulate` (the physics)  **parity unpinned**: the arithmetic lives in MuJoCo 3.3.2 (`mujoco.mj_step`,
                            requirements_full.txt:5), which is not importable here and which no reference test exercises.
                            Restated from MuJoCo's published model for this scene:
       fluid (ellipsoid model, sphere r=0.02, fluidcoef "0.235 0.25 0.0 1.0 1.0", density 1.225, viscosity 1.8e-5):
         drag     -(6 pi mu r + rho C_blunt pi r^2 |v|) v          Magnus  C_M rho V (w x v)
         added mass  (rho V/2) (v x w)   (virtual mass of a sphere = V/2; acceleration terms are dropped by MuJoCo)
         torque   -8 pi mu r^3 w         (C_ang = 0; Kutta lift vanishes for a sphere)
       contact (soft constraint, one sphere): a_n = (1-d) a0_n + d (-b v_n - k dist), f_n >= 0; friction drives the
         contact-point slip to zero with aref_t = -b_f v_t, limited by the elliptic cone |f_t| <= mu f_n;
         ball/table and ball/net pairs: solref (-1e6, -17), solreffriction (-0, -200), mu 0.1, solimp (0.98 0.99 0.001 0.5 2);
         ball/ground: MuJoCo defaults solref (0.02, 1), solimp (0.9 0.95 0.001 0.5 2), mu 1.
       integrator: MuJoCo steps 1 ms implicit-in-velocity; here classical RK4 with SUBSTEPS per millisecond
         (self-consistency: 4th-order convergence in free flight, tests/test_trajgen.py).
"""
import math
import random

import numpy as np

# ---------------------------------------------------------------- constants (helper.py:14-40, mujocosimulation.py:28-51)
HEIGHT, WIDTH = 1080, 1920
TABLE_HEIGHT, TABLE_WIDTH, TABLE_LENGTH = 0.76, 1.525, 2.74
NET_POST_OFFSET = 0.1525
NET_HEIGHT_ABOVE_TABLE = 0.1525
NET_TOTAL_HEIGHT = TABLE_HEIGHT + NET_HEIGHT_ABOVE_TABLE
NET_TOTAL_WIDTH = TABLE_WIDTH + 2 * NET_POST_OFFSET
TIMESTEP, MAX_SIMULATION_TIME, FPS = 0.001, 1.0, 500
HIT_Z_TABLE = TABLE_HEIGHT + 0.04
HIT_Z_GROUND = 0.08
HIT_X_MARGIN = 0.01
HIT_W = (0.75, 0.25)
FX, FY = 2033, 2180
CAMERA_POS = np.array([0.04381194, 8.92938715, 5.40070126])
CAMERA_UP = np.array([7.81340900e-04, -4.33644716e-01, 9.01083598e-01])
CAMERA_RIGHT = np.array([-0.99998599, 0.00437903, 0.0029745])

MIN_TRAJ_LEN_FRAMES = int(round(0.2 * FPS))
MIN_TRAJ_CUT_TIME_RATIO = 0.2
NET_CLEARANCE_X_MARGIN = 0.04
OOB = {'final_lose': (6.0, 3.0, -1.0), 'final_win': (TABLE_LENGTH / 2, TABLE_WIDTH, 0.7), 'intermediate': (4.5, 2.5, -1.0),
       'first_good': (2.5, 1.5, -1.0), 'first_short': (2.5, 1.5, 0.5), 'first_long': (2.5, 1.5, -1.0)}
MODES = list(OOB)
DIRECTIONS = ['left_to_right', 'right_to_left']
VALID_COUNTS = {'final_lose': (0, 0, 0), 'final_win': (2, 0, 0), 'intermediate': (1, 0, 0),
                'first_good': (1, 1, 0), 'first_short': (0, 2, 0), 'first_long': (0, 1, 0)}

# physics (helper.py:81-114)
R_BALL, M_BALL = 0.02, 0.0027
RHO, MU_AIR, G = 1.225, 0.000018, 9.81
C_BLUNT, C_MAGNUS = 0.235, 1.0
VOL = 4.0 / 3.0 * math.pi * R_BALL ** 3
AREA = math.pi * R_BALL ** 2
I_BALL = 0.4 * M_BALL * R_BALL ** 2
SUBSTEPS = 4
MAX_SAMPLES = 501


# ---------------------------------------------------------------- sampling (mujocosimulation.py:54-109)
def init_state(seed, mode, direction):
    """(r, v, w) exactly as `_init_simulation` draws them from random.Random(seed)."""
    rng = random.Random(seed)
    sign_x = 1 if direction == 'left_to_right' else -1
    r = np.empty(3)
    if 'first' in mode:
        r[0] = rng.uniform(1.0, 2.5) * sign_x
        r[1] = rng.uniform(-1.5, 1.5)
        r[2] = rng.uniform(0.8, 1.6)
    else:
        r[0] = rng.uniform(0.1, 4.0) * sign_x
        r[1] = rng.uniform(-2.0, 2.0)
        if abs(r[0]) < TABLE_LENGTH / 2 and abs(r[1]) < TABLE_WIDTH / 2:
            r[2] = rng.uniform(0.8, 1.8)
        else:
            r[2] = rng.uniform(0.5, 1.8)
    if 'first' in mode:
        c_y = TABLE_WIDTH / 2 if r[1] > 0 else -TABLE_WIDTH / 2
        c_x = TABLE_LENGTH / 2 if direction == 'left_to_right' else -TABLE_LENGTH / 2
    else:
        c_y = 0
        c_x = -TABLE_LENGTH / 2 if direction == 'left_to_right' else TABLE_LENGTH / 2
    base_phi = 180 + np.rad2deg(math.atan2(r[1] - c_y, r[0] - c_x))
    base_theta = 90 - np.rad2deg(math.atan2(r[2] - TABLE_HEIGHT, abs(r[0] - c_x)))
    if r[2] < TABLE_HEIGHT:
        min_theta, max_theta = max(90.0, base_theta - 25.0), min(170.0, base_theta + 60.0)
    else:
        min_theta, max_theta = max(10.0, base_theta - 25.0), min(150.0, base_theta + 60.0)
    speed = rng.uniform(3.0, 30.0)
    phi = rng.uniform(np.deg2rad(base_phi - 60.0), np.deg2rad(base_phi + 60.0))
    theta = rng.uniform(np.deg2rad(min_theta), np.deg2rad(max_theta))
    v = np.array([speed * math.sin(theta) * math.cos(phi), speed * math.sin(theta) * math.sin(phi), speed * math.cos(theta)])
    speed = rng.uniform(0.0, 500.0)
    phi = rng.uniform(0, 2 * math.pi)
    theta = rng.uniform(0, math.pi)
    w = np.array([speed * math.sin(theta) * math.cos(phi), speed * math.sin(theta) * math.sin(phi), speed * math.cos(theta)])
    return r, v, w


# ---------------------------------------------------------------- camera (helper.py:262-280; MuJoCo fixed camera from xyaxes)
def camera_matrices():
    """Mext (4,4) and Mint (3,3) of the fixed 'main' camera.  MuJoCo normalises x, makes y orthogonal to x, z = x cross y;
    `_calc_cammatrices` flips y and z rows and uses cx=(W-1)/2, cy=(H-1)/2, fx=2033, fy=2180."""
    x = CAMERA_RIGHT / np.linalg.norm(CAMERA_RIGHT)
    y = CAMERA_UP - x * np.dot(x, CAMERA_UP)
    y = y / np.linalg.norm(y)
    z = np.cross(x, y)
    R = np.stack([x, -y, -z])
    ex = np.eye(4)
    ex[:3, :3] = R
    ex[:3, 3] = -R @ CAMERA_POS
    fx = (FX / WIDTH) / 1.0 * WIDTH
    fy = (FY / HEIGHT) / 1.0 * HEIGHT
    return ex, np.array([[fx, 0, (WIDTH - 1) / 2], [0, fy, (HEIGHT - 1) / 2], [0, 0, 1.0]])


def project(r, ex, mint):
    """world2cam + cam2img (helper.py:177-257) for (..., 3) points."""
    rc = r @ ex[:3, :3].T + ex[:3, 3]
    ri = rc @ mint.T
    return ri[..., :2] / ri[..., 2:3]


# ---------------------------------------------------------------- physics (parity unpinned, see header)
def _impedance(dist, d0, dmax, width):
    x = np.clip(np.abs(dist) / width, 0.0, 1.0)
    y = np.where(x < 0.5, 2.0 * x * x, 1.0 - 2.0 * (1.0 - x) * (1.0 - x))        # midpoint 0.5, power 2
    return d0 + y * (dmax - d0)


def _box_contact(c, center, half):
    """Sphere centre c (N,3) against an axis-aligned box: signed distance of the sphere surface and outward normal."""
    q = c - center
    cl = np.clip(q, -half, half)
    diff = q - cl
    dn = np.linalg.norm(diff, axis=1)
    outside = dn > 0
    n_out = diff / np.where(outside, dn, 1.0)[:, None]
    # centre inside the box (deep penetration): push out through the nearest face
    pen = half - np.abs(q)
    ax = np.argmin(pen, axis=1)
    n_in = np.zeros_like(q)
    n_in[np.arange(len(q)), ax] = np.where(q[np.arange(len(q)), ax] >= 0, 1.0, -1.0)
    dist = np.where(outside, dn, -pen[np.arange(len(q)), ax]) - R_BALL
    return dist, np.where(outside[:, None], n_out, n_in)


_CONTACTS = (
    # (kind, geometry, k_raw, b_raw, b_fric_raw, mu, d0, dmax, width, direct)
    ('box', (np.array([0.0, 0.0, TABLE_HEIGHT / 2]), np.array([TABLE_LENGTH / 2, TABLE_WIDTH / 2, TABLE_HEIGHT / 2])), 1.0e6, 17.0, 200.0, 0.1, 0.98, 0.99, 0.001, True),
    ('box', (np.array([0.0, 0.0, TABLE_HEIGHT]), np.array([0.02, TABLE_HEIGHT + NET_POST_OFFSET, NET_HEIGHT_ABOVE_TABLE])), 1.0e6, 17.0, 200.0, 0.1, 0.98, 0.99, 0.001, True),
    ('plane', None, 0.02, 1.0, None, 1.0, 0.9, 0.95, 0.001, False),
)


def accel(r, v, w):
    """Linear and angular acceleration of N balls (N,3 each)."""
    speed = np.linalg.norm(v, axis=1, keepdims=True)
    f = -(6.0 * math.pi * MU_AIR * R_BALL + RHO * C_BLUNT * AREA * speed) * v
    f = f + C_MAGNUS * RHO * VOL * np.cross(w, v) + 0.5 * RHO * VOL * np.cross(v, w)
    f[:, 2] -= M_BALL * G
    tq = -8.0 * math.pi * MU_AIR * R_BALL ** 3 * w
    a0, al0 = f / M_BALL, tq / I_BALL
    fc = np.zeros_like(r)
    tc = np.zeros_like(r)
    for kind, geo, k_raw, b_raw, bf_raw, mu, d0, dmax, width, direct in _CONTACTS:
        if kind == 'box':
            dist, n = _box_contact(r, geo[0], geo[1])
        else:
            dist = r[:, 2] - R_BALL
            n = np.zeros_like(r)
            n[:, 2] = 1.0
        act = dist < 0
        if not act.any():
            continue
        d = _impedance(dist, d0, dmax, width)
        if direct:
            k, b, bf = k_raw * d / (dmax * dmax), b_raw / dmax, bf_raw / dmax
        else:                       # (timeconst, dampratio)
            b = 2.0 / (dmax * k_raw)
            k = d / (dmax * dmax * k_raw * k_raw * b_raw * b_raw)
            bf = b
        arm = -R_BALL * n
        vc = v + np.cross(w, arm)
        vn = np.sum(vc * n, axis=1)
        vt = vc - vn[:, None] * n
        ac = a0 + np.cross(al0, arm)
        an0 = np.sum(ac * n, axis=1)
        at0 = ac - an0[:, None] * n
        fn = np.maximum(M_BALL * d * ((-b * vn - k * dist) - an0), 0.0)
        ft = (M_BALL / 3.5) * d[:, None] * ((-bf * vt) - at0)          # 1/m + r^2/I = 3.5/m at the contact point
        ftn = np.linalg.norm(ft, axis=1)
        lim = mu * fn
        ft = ft * np.where(ftn > lim, lim / np.where(ftn > 0, ftn, 1.0), 1.0)[:, None]
        fn = np.where(act, fn, 0.0)
        ft = np.where(act[:, None], ft, 0.0)
        fc = fc + fn[:, None] * n + ft
        tc = tc + np.cross(arm, ft)
    return a0 + fc / M_BALL, al0 + tc / I_BALL


def rk4_step(r, v, w, h):
    def f(r_, v_, w_):
        a, al = accel(r_, v_, w_)
        return v_, a, al
    k1 = f(r, v, w)
    k2 = f(r + 0.5 * h * k1[0], v + 0.5 * h * k1[1], w + 0.5 * h * k1[2])
    k3 = f(r + 0.5 * h * k2[0], v + 0.5 * h * k2[1], w + 0.5 * h * k2[2])
    k4 = f(r + h * k3[0], v + h * k3[1], w + h * k3[2])
    r = r + (h / 6.0) * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0])
    v = v + (h / 6.0) * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1])
    w = w + (h / 6.0) * (k1[2] + 2 * k2[2] + 2 * k3[2] + k4[2])
    return r, v, w


def step_ms(r, v, w, n_ms, substeps=SUBSTEPS):
    """Advance n_ms MuJoCo timesteps (1 ms each)."""
    h = TIMESTEP / substeps
    for _ in range(n_ms * substeps):
        r, v, w = rk4_step(r, v, w, h)
    return r, v, w


def save_times():
    """The time labels of the sampling loop: next_save_time accumulates 1/FPS in floating point (:116,:150)."""
    t, out = 0.0, []
    while t < MAX_SIMULATION_TIME:
        out.append(t)
        t += 1 / FPS
    return np.array(out)


def is_oob(p, mode, direction):
    """Out-of-bounds rule of the sampling loop (mujocosimulation.py:123-139) for one position."""
    correct_side = p[0] < 0 if direction == 'left_to_right' else p[0] > 0
    ox, oy, oz = OOB[mode]
    if mode == 'final_lose':
        return abs(p[0]) > ox or abs(p[1]) > oy
    if 'final' in mode or 'intermediate' in mode:
        return bool(correct_side and (abs(p[0]) > ox or abs(p[1]) > oy or p[2] < oz))
    if mode == 'first_short':
        return abs(p[0]) > ox or abs(p[1]) > oy or p[2] < oz
    return bool(correct_side and (abs(p[0]) > ox or abs(p[1]) > oy))


def simulate(seeds, mode, direction, substeps=SUBSTEPS):
    """Sampling loop for a batch of seeds.  Returns pos, vel, rot (N, S, 3) and n_saved (N,): sample 0 is the state after
    the first 1 ms step (labelled t=0), sample k the state at 2k ms; a trajectory stops at the first out-of-bounds or
    out-of-image sample (mujocosimulation.py:112-151)."""
    times = save_times()
    n = len(seeds)
    st = [init_state(s, mode, direction) for s in seeds]
    r = np.stack([s[0] for s in st]); v = np.stack([s[1] for s in st]); w = np.stack([s[2] for s in st])
    ex, mint = camera_matrices()
    S = len(times)
    pos = np.zeros((n, S, 3)); vel = np.zeros((n, S, 3)); rot = np.zeros((n, S, 3))
    n_saved = np.zeros(n, dtype=np.int64)
    alive = np.ones(n, dtype=bool)
    r, v, w = step_ms(r, v, w, 1, substeps)
    for k in range(S):
        if k > 0:
            r, v, w = step_ms(r, v, w, 1 if k == 1 else 2, substeps)
        img = project(r, ex, mint)
        for i in np.nonzero(alive)[0]:
            if is_oob(r[i], mode, direction) or not (0 <= img[i, 0] < WIDTH and 0 <= img[i, 1] < HEIGHT):
                alive[i] = False
                continue
            pos[i, k], vel[i, k], rot[i, k] = r[i], v[i], w[i]
            n_saved[i] = k + 1
        if not alive.any():
            break
    return pos, vel, rot, n_saved


# ---------------------------------------------------------------- selection (helper.py:282-321, mujocosimulation.py:152-219)
def count_hits(positions, direction):
    positions = np.asarray(positions)
    x, y, z = positions[:, 0], positions[:, 1], positions[:, 2]
    if direction == 'left_to_right':
        opp = (x < -HIT_X_MARGIN) & (x > -TABLE_LENGTH / 2)
        own = (x < TABLE_LENGTH / 2) & (x > HIT_X_MARGIN)
    else:
        opp = (x < TABLE_LENGTH / 2) & (x > HIT_X_MARGIN)
        own = (x < -HIT_X_MARGIN) & (x > -TABLE_LENGTH / 2)
    base = (z < HIT_Z_TABLE) & (np.abs(y) < TABLE_WIDTH / 2)
    out = []
    for mask in (base & opp, base & own, z <= HIT_Z_GROUND):
        hits, start = [], None
        for i, b in enumerate(mask):
            if i == 0 and b:
                start = i
            elif b and not mask[i - 1]:
                start = i
            if not b and mask[i - 1] and i != 0:          # note: at i == 0 the reference reads mask[-1] but requires i != 0
                end = i - 1
                mid = (end + start) / 2 / FPS
                low = (np.argmin(z[start:end + 1]) + start) / FPS
                hits.append(HIT_W[0] * mid + HIT_W[1] * low)
        out.append(hits)
    return out[0], out[1], out[2]


def select(positions, times, mode, direction):
    """Everything after the sampling loop for one trajectory: returns None (rejected) or (n_keep, bounces)."""
    positions = np.asarray(positions)
    times = np.asarray(times)[:len(positions)]
    if len(positions) < MIN_TRAJ_LEN_FRAMES:
        return None
    ho, hw, hg = count_hits(positions, direction)
    if np.max(positions[:, 2]) > (1.4 if 'first' in mode else 1.8):
        return None
    tmin = MIN_TRAJ_CUT_TIME_RATIO * MAX_SIMULATION_TIME
    cut = -1

    def idx(t):
        return int(np.sum(times < t)) - 1
    if mode in ('final_lose', 'intermediate', 'first_long'):
        if len(hg) > 0 and hg[0] >= tmin:
            cut = idx(hg[0]); hg = []
    elif mode == 'final_win':
        if len(ho) > 2 and ho[2] >= tmin:
            cut = idx(ho[2]); ho = ho[:2]
        elif len(hg) > 0 and hg[0] >= tmin:
            cut = idx(hg[0])
        if cut != -1:
            hg = []
    elif mode == 'first_good':
        if len(ho) > 1 and ho[1] >= tmin:
            cut = idx(ho[1]); ho = ho[:1]
        elif len(hg) > 0 and hg[0] >= tmin:
            cut = idx(hg[0])
        if cut != -1:
            hg = []
    elif mode == 'first_short':
        if len(hw) > 2 and hw[2] >= tmin:
            cut = idx(hw[2]); hw, ho, hg = hw[:2], [], []
        elif len(ho) > 0 and ho[0] >= tmin:
            cut = idx(ho[0]); ho, hg = [], []
        elif len(hg) > 0 and hg[0] >= tmin:
            cut = idx(hg[0]); hg = []
    if cut != -1:
        positions = positions[:cut]
    if len(positions) < MIN_TRAJ_LEN_FRAMES:
        return None
    close = np.abs(positions[:, 0]) < NET_CLEARANCE_X_MARGIN
    if np.any(close):
        if np.max(positions[close, 2]) < NET_TOTAL_HEIGHT and np.min(np.abs(positions[close, 1])) < NET_TOTAL_WIDTH / 2:
            return None
    last_x = positions[-1][0]
    opposite = last_x < 0 if direction == 'left_to_right' else last_x > 0
    if mode in ('final_lose', 'first_long') and not opposite:
        return None
    if (len(ho), len(hw), len(hg)) != VALID_COUNTS[mode]:
        return None
    return len(positions), np.array(sorted(ho + hw))


def seed_order(current_seed, batch_size, num_processes):
    """Order in which `get_valid_trajectories` (:222-238) collects one batch: process j takes seeds j, j+P, j+2P, ..."""
    out = []
    for j in range(num_processes):
        out.extend(range(current_seed + j, current_seed + batch_size, num_processes))
    return out


def generate(num_trajectories, num_processes, mode, direction, substeps=SUBSTEPS):
    """`get_valid_trajectories`: list of dicts with the reference's keys (positions, velocities, rotations, times, Mext,
    Mint, bounces, seed)."""
    times = save_times()
    ex, mint = camera_matrices()
    found, current, batch = [], 0, min(1024, num_trajectories)
    while len(found) < num_trajectories:
        seeds = seed_order(current, batch, num_processes)
        pos, vel, rot, ns = simulate(seeds, mode, direction, substeps)
        for i, s in enumerate(seeds):
            res = select(pos[i, :ns[i]], times, mode, direction)
            if res is None:
                continue
            n, bounces = res
            found.append({'positions': pos[i, :n].copy(), 'velocities': vel[i, :n].copy(), 'rotations': rot[i, :n].copy(),
                          'times': times[:n].copy(), 'Mext': np.repeat(ex[None], n, 0), 'Mint': np.repeat(mint[None], n, 0),
                          'bounces': bounces, 'seed': s})
        current += batch
    return found[:num_trajectories]
