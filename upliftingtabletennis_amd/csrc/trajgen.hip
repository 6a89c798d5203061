// f2: batched synthetic-trajectory generator (reference syntheticdataset/mujocosimulation.py + helper.py), fp64.
//   init_kernel      seed -> MT19937 state exactly as CPython's random.Random(seed) builds it -> the nine uniform draws of
//                    _init_simulation (mujocosimulation.py:54-109) -> initial position / velocity / spin
//   simulate_kernel  one lane per seed: RK4 of gravity + drag + Magnus + added mass + soft contacts (table, net, ground),
//                    1 ms MuJoCo steps split into `substeps`, the reference's sampling loop with its out-of-bounds and
//                    out-of-image stops (:112-151)
//   select_kernel    one lane per seed: _count_hits (helper.py:282-321) and every rejection / cut rule (:152-219)
// The physics restates MuJoCo's published fluid and soft-contact model for this one-sphere scene (see
// oracle/trajgen_ref.py); MuJoCo itself is absent, so that part of the parity is unpinned.  Sampling and selection are
// pinned by the reference's own code (tests/golden/trajgen.npz).
// Layout: samples[(k*9 + c) * n + lane] (k = sample, c = x y z vx vy vz wx wy wz): every store is coalesced over seeds.
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>

using namespace ttup;

namespace {

constexpr int MAX_SAMPLES = 512;          // >= the 500 or 501 labels of the sampling loop
constexpr double PI = 3.141592653589793238462643383279502884;
constexpr double TABLE_HEIGHT = 0.76, TABLE_WIDTH = 1.525, TABLE_LENGTH = 2.74;
constexpr double NET_POST_OFFSET = 0.1525, NET_ABOVE = 0.1525;
constexpr double NET_TOTAL_HEIGHT = TABLE_HEIGHT + NET_ABOVE, NET_TOTAL_WIDTH = TABLE_WIDTH + 2 * NET_POST_OFFSET;
constexpr double R_BALL = 0.02, M_BALL = 0.0027, RHO = 1.225, MU_AIR = 0.000018, GRAV = 9.81;
constexpr double C_BLUNT = 0.235, C_MAGNUS = 1.0;
constexpr double IMG_W = 1920.0, IMG_H = 1080.0;
constexpr int FPS = 500;

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }

// ------------------------------------------------------------------ CPython random.Random(seed)
// init_genrand(19650218) is seed-independent: its 624 words come from the host once (mt0).  init_by_array(key) with
// key = 32-bit little-endian words of abs(seed); the per-lane state lives in global scratch, word-major (coalesced).
__global__ void init_kernel(const long long* seeds, int n, int mode, int direction, const unsigned* mt0, unsigned* mt, double* state) {
#pragma clang fp contract(off)      // a + (b - a) * random() must round like CPython's two operations
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n) return;
    const long long sd = seeds[lane];
    const unsigned long long a = sd < 0 ? (unsigned long long)(-sd) : (unsigned long long)sd;
    const unsigned key[2] = {(unsigned)(a & 0xffffffffu), (unsigned)(a >> 32)};
    const int keylen = key[1] ? 2 : 1;
#define MT(i) mt[(size_t)(i) * n + lane]
    // first pass: 624 steps starting at i = 1
    unsigned prev = mt0[0];
    int j = 0;
    for (int i = 1; i < 624; ++i) {
        const unsigned v = (mt0[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (unsigned)j;
        MT(i) = v; prev = v;
        if (++j >= keylen) j = 0;
    }
    MT(0) = prev;                                   // i wrapped: mt[0] = mt[623]
    {
        const unsigned v = (MT(1) ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (unsigned)j;      // 624th step at i = 1
        MT(1) = v; prev = v;
    }
    // second pass: 623 steps starting at i = 2
    for (int i = 2; i < 624; ++i) {
        const unsigned v = (MT(i) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (unsigned)i;
        MT(i) = v; prev = v;
    }
    {
        const unsigned v = (MT(1) ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;        // wrapped: mt[0] = mt[623], i = 1
        MT(1) = v;
    }
    // mt[0] = 0x80000000; first 18 outputs of the regenerated state
    unsigned out[18];
    unsigned cur = 0x80000000u;
    for (int kk = 0; kk < 18; ++kk) {
        const unsigned nxt = MT(kk + 1);
        const unsigned y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
        unsigned v = MT(kk + 397) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11; v ^= (v << 7) & 0x9d2c5680u; v ^= (v << 15) & 0xefc60000u; v ^= v >> 18;
        out[kk] = v;
        cur = nxt;
    }
#undef MT
    int o = 0;
    auto uniform = [&](double lo, double hi) {
        const unsigned x = out[o] >> 5, y = out[o + 1] >> 6;
        o += 2;
        return lo + (hi - lo) * (((double)x * 67108864.0 + (double)y) * (1.0 / 9007199254740992.0));
    };
    const bool first = mode >= 3;                   // first_good, first_short, first_long
    const bool l2r = direction == 0;
    const double sign_x = l2r ? 1.0 : -1.0;
    double r[3];
    if (first) {
        r[0] = uniform(1.0, 2.5) * sign_x; r[1] = uniform(-1.5, 1.5); r[2] = uniform(0.8, 1.6);
    } else {
        r[0] = uniform(0.1, 4.0) * sign_x; r[1] = uniform(-2.0, 2.0);
        r[2] = (fabs(r[0]) < TABLE_LENGTH / 2 && fabs(r[1]) < TABLE_WIDTH / 2) ? uniform(0.8, 1.8) : uniform(0.5, 1.8);
    }
    double cx, cy;
    if (first) { cy = r[1] > 0 ? TABLE_WIDTH / 2 : -TABLE_WIDTH / 2; cx = l2r ? TABLE_LENGTH / 2 : -TABLE_LENGTH / 2; }
    else { cy = 0.0; cx = l2r ? -TABLE_LENGTH / 2 : TABLE_LENGTH / 2; }
    const double r2d = 180.0 / PI, d2r = PI / 180.0;
    const double base_phi = 180.0 + atan2(r[1] - cy, r[0] - cx) * r2d;
    const double base_theta = 90.0 - atan2(r[2] - TABLE_HEIGHT, fabs(r[0] - cx)) * r2d;
    double min_theta, max_theta;
    if (r[2] < TABLE_HEIGHT) { min_theta = fmax(90.0, base_theta - 25.0); max_theta = fmin(170.0, base_theta + 60.0); }
    else { min_theta = fmax(10.0, base_theta - 25.0); max_theta = fmin(150.0, base_theta + 60.0); }
    double speed = uniform(3.0, 30.0);
    double phi = uniform((base_phi - 60.0) * d2r, (base_phi + 60.0) * d2r);
    double theta = uniform(min_theta * d2r, max_theta * d2r);
    const double v[3] = {speed * sin(theta) * cos(phi), speed * sin(theta) * sin(phi), speed * cos(theta)};
    speed = uniform(0.0, 500.0);
    phi = uniform(0.0, 2.0 * PI);
    theta = uniform(0.0, PI);
    const double w[3] = {speed * sin(theta) * cos(phi), speed * sin(theta) * sin(phi), speed * cos(theta)};
    for (int c = 0; c < 3; ++c) { state[(size_t)c * n + lane] = r[c]; state[(size_t)(3 + c) * n + lane] = v[c]; state[(size_t)(6 + c) * n + lane] = w[c]; }
}

// ------------------------------------------------------------------ physics
struct Contact { double k_raw, b_raw, bf_raw, mu, d0, dmax, width; bool direct; };

__device__ __forceinline__ double impedance(double dist, double d0, double dmax, double width) {
    double x = fabs(dist) / width;
    x = x > 1.0 ? 1.0 : x;
    const double y = x < 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);       // midpoint 0.5, power 2
    return d0 + y * (dmax - d0);
}

// signed distance of the sphere surface to an axis-aligned box and the outward normal
__device__ __forceinline__ double box_contact(V3 c, V3 center, V3 half, V3* n) {
    const V3 q = c - center;
    const V3 cl = {fmin(fmax(q.x, -half.x), half.x), fmin(fmax(q.y, -half.y), half.y), fmin(fmax(q.z, -half.z), half.z)};
    const V3 diff = q - cl;
    const double dn = norm(diff);
    if (dn > 0.0) { *n = (1.0 / dn) * diff; return dn - R_BALL; }
    const double px = half.x - fabs(q.x), py = half.y - fabs(q.y), pz = half.z - fabs(q.z);      // centre inside: nearest face
    if (px <= py && px <= pz) { *n = {q.x >= 0 ? 1.0 : -1.0, 0.0, 0.0}; return -px - R_BALL; }
    if (py <= pz) { *n = {0.0, q.y >= 0 ? 1.0 : -1.0, 0.0}; return -py - R_BALL; }
    *n = {0.0, 0.0, q.z >= 0 ? 1.0 : -1.0};
    return -pz - R_BALL;
}

__device__ __forceinline__ void contact_force(double dist, V3 n, const Contact& ct, V3 v, V3 w, V3 a0, V3 al0, V3* fc, V3* tc) {
    if (!(dist < 0.0)) return;
    const double d = impedance(dist, ct.d0, ct.dmax, ct.width);
    double k, b, bf;
    if (ct.direct) { k = ct.k_raw * d / (ct.dmax * ct.dmax); b = ct.b_raw / ct.dmax; bf = ct.bf_raw / ct.dmax; }
    else { b = 2.0 / (ct.dmax * ct.k_raw); k = d / (ct.dmax * ct.dmax * ct.k_raw * ct.k_raw * ct.b_raw * ct.b_raw); bf = b; }
    const V3 arm = (-R_BALL) * n;
    const V3 vc = v + cross(w, arm);
    const double vn = dot(vc, n);
    const V3 vt = vc - vn * n;
    const V3 ac = a0 + cross(al0, arm);
    const double an0 = dot(ac, n);
    const V3 at0 = ac - an0 * n;
    double fn = M_BALL * d * ((-b * vn - k * dist) - an0);
    fn = fn > 0.0 ? fn : 0.0;
    V3 ft = (M_BALL / 3.5 * d) * ((-bf) * vt - at0);          // 1/m + r^2/I = 3.5/m at the contact point
    const double ftn = norm(ft), lim = ct.mu * fn;
    if (ftn > lim) ft = (lim / ftn) * ft;
    *fc = *fc + fn * n + ft;
    *tc = *tc + cross(arm, ft);
}

__device__ __forceinline__ void accel(V3 r, V3 v, V3 w, V3* a, V3* al) {
    const double VOL = 4.0 / 3.0 * PI * R_BALL * R_BALL * R_BALL, AREA = PI * R_BALL * R_BALL, I_BALL = 0.4 * M_BALL * R_BALL * R_BALL;
    const double speed = norm(v);
    V3 f = (-(6.0 * PI * MU_AIR * R_BALL + RHO * C_BLUNT * AREA * speed)) * v;
    f = f + (C_MAGNUS * RHO * VOL) * cross(w, v) + (0.5 * RHO * VOL) * cross(v, w);
    f.z -= M_BALL * GRAV;
    const V3 tq = (-8.0 * PI * MU_AIR * R_BALL * R_BALL * R_BALL) * w;
    const V3 a0 = (1.0 / M_BALL) * f, al0 = (1.0 / I_BALL) * tq;
    V3 fc = {0, 0, 0}, tc = {0, 0, 0};
    // cheap reject: nothing to touch above the net top or away from table / net / ground
    if (r.z < NET_TOTAL_HEIGHT + R_BALL) {
        const Contact pair = {1.0e6, 17.0, 200.0, 0.1, 0.98, 0.99, 0.001, true};
        const Contact ground = {0.02, 1.0, 0.0, 1.0, 0.9, 0.95, 0.001, false};
        V3 n;
        double dist = box_contact(r, {0.0, 0.0, TABLE_HEIGHT / 2}, {TABLE_LENGTH / 2, TABLE_WIDTH / 2, TABLE_HEIGHT / 2}, &n);
        contact_force(dist, n, pair, v, w, a0, al0, &fc, &tc);
        dist = box_contact(r, {0.0, 0.0, TABLE_HEIGHT}, {0.02, TABLE_HEIGHT + NET_POST_OFFSET, NET_ABOVE}, &n);
        contact_force(dist, n, pair, v, w, a0, al0, &fc, &tc);
        contact_force(r.z - R_BALL, {0.0, 0.0, 1.0}, ground, v, w, a0, al0, &fc, &tc);
    }
    *a = a0 + (1.0 / M_BALL) * fc;
    *al = al0 + (1.0 / I_BALL) * tc;
}

__device__ __forceinline__ void rk4(V3& r, V3& v, V3& w, double h) {
    V3 a1, l1, a2, l2, a3, l3, a4, l4;
    accel(r, v, w, &a1, &l1);
    const V3 v2 = v + (0.5 * h) * a1;
    accel(r + (0.5 * h) * v, v2, w + (0.5 * h) * l1, &a2, &l2);
    const V3 v3 = v + (0.5 * h) * a2;
    accel(r + (0.5 * h) * v2, v3, w + (0.5 * h) * l2, &a3, &l3);
    const V3 v4 = v + h * a3;
    accel(r + h * v3, v4, w + h * l3, &a4, &l4);
    r = r + (h / 6.0) * (v + 2.0 * v2 + 2.0 * v3 + v4);
    v = v + (h / 6.0) * (a1 + 2.0 * a2 + 2.0 * a3 + a4);
    w = w + (h / 6.0) * (l1 + 2.0 * l2 + 2.0 * l3 + l4);
}

struct Cam { double ex[12]; double in[9]; };          // rows 0..2 of Mext, Mint

__device__ __forceinline__ bool out_of_bounds(V3 p, int mode, bool l2r) {
    const bool correct_side = l2r ? p.x < 0 : p.x > 0;
    switch (mode) {
        case 0: return fabs(p.x) > 6.0 || fabs(p.y) > 3.0;                                                         // final_lose
        case 1: return correct_side && (fabs(p.x) > TABLE_LENGTH / 2 || fabs(p.y) > TABLE_WIDTH || p.z < 0.7);    // final_win
        case 2: return correct_side && (fabs(p.x) > 4.5 || fabs(p.y) > 2.5 || p.z < -1.0);                         // intermediate
        case 4: return fabs(p.x) > 2.5 || fabs(p.y) > 1.5 || p.z < 0.5;                                            // first_short
        default: return correct_side && (fabs(p.x) > 2.5 || fabs(p.y) > 1.5);                                      // first_good, first_long
    }
}

__global__ void simulate_kernel(const double* state, int n, int mode, int direction, int substeps, int n_labels, Cam cam,
                                double* samples, int* n_saved) {
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n) return;
    V3 r = {state[lane], state[(size_t)n + lane], state[(size_t)2 * n + lane]};
    V3 v = {state[(size_t)3 * n + lane], state[(size_t)4 * n + lane], state[(size_t)5 * n + lane]};
    V3 w = {state[(size_t)6 * n + lane], state[(size_t)7 * n + lane], state[(size_t)8 * n + lane]};
    const double h = 0.001 / substeps;
    const bool l2r = direction == 0;
    int saved = 0;
    for (int k = 0; k < n_labels; ++k) {
        // sample 0 = state after the first step (the reference steps once before its loop), sample k = state at 2k ms
        const int steps = (k == 0 || k == 1 ? 1 : 2) * substeps;
        for (int s = 0; s < steps; ++s) rk4(r, v, w, h);
        if (out_of_bounds(r, mode, l2r)) break;
        const double xc = cam.ex[0] * r.x + cam.ex[1] * r.y + cam.ex[2] * r.z + cam.ex[3];
        const double yc = cam.ex[4] * r.x + cam.ex[5] * r.y + cam.ex[6] * r.z + cam.ex[7];
        const double zc = cam.ex[8] * r.x + cam.ex[9] * r.y + cam.ex[10] * r.z + cam.ex[11];
        const double iz = cam.in[6] * xc + cam.in[7] * yc + cam.in[8] * zc;
        const double ix = (cam.in[0] * xc + cam.in[1] * yc + cam.in[2] * zc) / iz, iy = (cam.in[3] * xc + cam.in[4] * yc + cam.in[5] * zc) / iz;
        if (!(ix >= 0.0 && ix < IMG_W && iy >= 0.0 && iy < IMG_H)) break;
        double* o = samples + (size_t)k * 9 * n + lane;
        o[0] = r.x; o[(size_t)n] = r.y; o[(size_t)2 * n] = r.z;
        o[(size_t)3 * n] = v.x; o[(size_t)4 * n] = v.y; o[(size_t)5 * n] = v.z;
        o[(size_t)6 * n] = w.x; o[(size_t)7 * n] = w.y; o[(size_t)8 * n] = w.z;
        saved = k + 1;
    }
    n_saved[lane] = saved;
}

// ------------------------------------------------------------------ selection
struct Hits { int count; double t[3]; };          // true count, first three hit times

struct TimesTable { const double* t; };

__global__ void select_kernel(const double* samples, const int* n_saved, int n, int mode, int direction, const double* times,
                              int* n_keep, double* bounces, int* n_bounces) {
#pragma clang fp contract(off)      // hit times are compared bit for bit with the reference's Python arithmetic
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n) return;
    n_keep[lane] = 0; n_bounces[lane] = 0;
    const int len = n_saved[lane];
    const int MIN_LEN = 100;                      // int(round(0.2 * FPS))
    if (len < MIN_LEN) return;
    const bool l2r = direction == 0;
    const bool first = mode >= 3;
#define PX(k) samples[((size_t)(k) * 9 + 0) * n + lane]
#define PY(k) samples[((size_t)(k) * 9 + 1) * n + lane]
#define PZ(k) samples[((size_t)(k) * 9 + 2) * n + lane]
    // ---- _count_hits: three interval state machines in one pass (opponent, own, ground)
    Hits hit[3] = {{0, {0, 0, 0}}, {0, {0, 0, 0}}, {0, {0, 0, 0}}};
    bool prev[3] = {false, false, false};
    int start[3] = {0, 0, 0}, amin[3] = {0, 0, 0};
    double zmin[3] = {0, 0, 0};
    double zmax = -1e300;
    for (int i = 0; i < len; ++i) {
        const double x = PX(i), y = PY(i), z = PZ(i);
        zmax = z > zmax ? z : zmax;
        const bool neg = x < -0.01 && x > -TABLE_LENGTH / 2, pos = x < TABLE_LENGTH / 2 && x > 0.01;
        const bool base = z < TABLE_HEIGHT + 0.04 && fabs(y) < TABLE_WIDTH / 2;
        const bool m[3] = {base && (l2r ? neg : pos), base && (l2r ? pos : neg), z <= 0.08};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (m[q] && (i == 0 || !prev[q])) { start[q] = i; amin[q] = i; zmin[q] = z; }
            else if (m[q] && z < zmin[q]) { amin[q] = i; zmin[q] = z; }           // first index of the minimum (np.argmin)
            if (!m[q] && i != 0 && prev[q]) {
                const int end = i - 1;
                const double mid = (double)(end + start[q]) / 2 / FPS, low = (double)amin[q] / FPS;
                if (hit[q].count < 3) hit[q].t[hit[q].count] = 0.75 * mid + 0.25 * low;
                hit[q].count++;
            }
            prev[q] = m[q];
        }
    }
    if (zmax > (first ? 1.4 : 1.8)) return;
    // ---- cut rules
    int no = hit[0].count, nw = hit[1].count, ng = hit[2].count;
    const double tmin = 0.2 * 1.0;
    auto index_of = [&](double t) { int c = 0; for (int i = 0; i < len; ++i) c += times[i] < t; return c - 1; };
    int cut = -1;
    if (mode == 0 || mode == 2 || mode == 5) {                    // final_lose, intermediate, first_long
        if (ng > 0 && hit[2].t[0] >= tmin) { cut = index_of(hit[2].t[0]); ng = 0; }
    } else if (mode == 1) {                                        // final_win
        if (no > 2 && hit[0].t[2] >= tmin) { cut = index_of(hit[0].t[2]); no = 2; }
        else if (ng > 0 && hit[2].t[0] >= tmin) cut = index_of(hit[2].t[0]);
        if (cut != -1) ng = 0;
    } else if (mode == 3) {                                        // first_good
        if (no > 1 && hit[0].t[1] >= tmin) { cut = index_of(hit[0].t[1]); no = 1; }
        else if (ng > 0 && hit[2].t[0] >= tmin) cut = index_of(hit[2].t[0]);
        if (cut != -1) ng = 0;
    } else {                                                       // first_short
        if (nw > 2 && hit[1].t[2] >= tmin) { cut = index_of(hit[1].t[2]); nw = 2; no = 0; ng = 0; }
        else if (no > 0 && hit[0].t[0] >= tmin) { cut = index_of(hit[0].t[0]); no = 0; ng = 0; }
        else if (ng > 0 && hit[2].t[0] >= tmin) { cut = index_of(hit[2].t[0]); ng = 0; }
    }
    int keep = len;
    if (cut != -1) keep = cut < len ? cut : len;       // positions[:cut_index]
    if (keep < MIN_LEN) return;
    // ---- net clearance, final side, bounce counts
    bool any_close = false;
    double zc = -1e300, yc = 1e300;
    for (int i = 0; i < keep; ++i) {
        if (fabs(PX(i)) < 0.04) { any_close = true; const double z = PZ(i), ay = fabs(PY(i)); zc = z > zc ? z : zc; yc = ay < yc ? ay : yc; }
    }
    if (any_close && zc < NET_TOTAL_HEIGHT && yc < NET_TOTAL_WIDTH / 2) return;
    const double lx = PX(keep - 1);
    if ((mode == 0 || mode == 5) && !(l2r ? lx < 0 : lx > 0)) return;
    const int want_o[6] = {0, 2, 1, 1, 0, 0}, want_w[6] = {0, 0, 0, 1, 2, 1};
    if (no != want_o[mode] || nw != want_w[mode] || ng != 0) return;
    // bounces = sorted(hits_opponent + hits_own): at most 2 + 2 entries here
    double b[4]; int nb = 0;
    for (int i = 0; i < no; ++i) b[nb++] = hit[0].t[i];
    for (int i = 0; i < nw; ++i) b[nb++] = hit[1].t[i];
    for (int i = 1; i < nb; ++i) { const double key = b[i]; int j = i - 1; while (j >= 0 && b[j] > key) { b[j + 1] = b[j]; --j; } b[j + 1] = key; }
    for (int i = 0; i < nb; ++i) bounces[(size_t)lane * 4 + i] = b[i];
    n_bounces[lane] = nb;
    n_keep[lane] = keep;
#undef PX
#undef PY
#undef PZ
}

// labels of the sampling loop: next_save_time accumulates 1/FPS in floating point (mujocosimulation.py:116,150)
int label_table(double* out) {
    int n = 0;
    double t = 0.0;
    while (t < 1.0) { if (out) out[n] = t; ++n; t += 1.0 / FPS; }
    return n;
}

}  // namespace

extern "C" int ttup_trajgen_max_samples(void) { return label_table(nullptr); }

extern "C" size_t ttup_trajgen_workspace_bytes(int n_seeds) {
    return n_seeds <= 0 ? 0 : (size_t)n_seeds * (624 * sizeof(unsigned) + 9 * sizeof(double)) + 624 * sizeof(unsigned) + MAX_SAMPLES * sizeof(double);
}

extern "C" int ttup_trajgen_simulate(const int64_t* seeds_dev, int n_seeds, int mode, int direction, int substeps, const double* cam_host,
                                     double* samples_dev, int* n_saved_dev, double* init_dev, void* workspace, size_t workspace_bytes, void* stream) {
    TTUP_REQUIRE(seeds_dev && cam_host && samples_dev && n_saved_dev && workspace, TTUP_EINVAL, "ttup_trajgen_simulate: null pointer");
    TTUP_REQUIRE(n_seeds >= 0 && mode >= 0 && mode < 6 && (direction == 0 || direction == 1) && substeps >= 1 && substeps <= 64, TTUP_EINVAL,
                 "ttup_trajgen_simulate: bad mode %d / direction %d / substeps %d", mode, direction, substeps);
    TTUP_REQUIRE(workspace_bytes >= ttup_trajgen_workspace_bytes(n_seeds), TTUP_EINVAL, "ttup_trajgen_simulate: workspace too small");
    if (n_seeds == 0) return TTUP_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned* mt0_dev = (unsigned*)workspace;
    double* times_dev = (double*)(mt0_dev + 624);
    double* state = times_dev + MAX_SAMPLES;
    unsigned* mt = (unsigned*)(state + (size_t)9 * n_seeds);
    unsigned mt0[624];
    mt0[0] = 19650218u;
    for (int i = 1; i < 624; ++i) mt0[i] = 1812433253u * (mt0[i - 1] ^ (mt0[i - 1] >> 30)) + (unsigned)i;
    TTUP_HIP_CHECK(hipMemcpyAsync(mt0_dev, mt0, sizeof mt0, hipMemcpyHostToDevice, st));
    TTUP_HIP_CHECK(hipStreamSynchronize(st));          // mt0 lives on this stack frame
    const int threads = 64, blocks = cdiv(n_seeds, threads);
    hipLaunchKernelGGL(init_kernel, dim3(blocks), dim3(threads), 0, st, (const long long*)seeds_dev, n_seeds, mode, direction, mt0_dev, mt, state);
    TTUP_LAUNCH_CHECK();
    if (init_dev) TTUP_HIP_CHECK(hipMemcpyAsync(init_dev, state, (size_t)9 * n_seeds * sizeof(double), hipMemcpyDeviceToDevice, st));
    Cam cam;
    for (int i = 0; i < 12; ++i) cam.ex[i] = cam_host[i];
    for (int i = 0; i < 9; ++i) cam.in[i] = cam_host[16 + i];
    hipLaunchKernelGGL(simulate_kernel, dim3(blocks), dim3(threads), 0, st, state, n_seeds, mode, direction, substeps, label_table(nullptr), cam, samples_dev, n_saved_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

extern "C" int ttup_trajgen_select(const double* samples_dev, const int* n_saved_dev, int n_seeds, int mode, int direction,
                                   int* n_keep_dev, double* bounces_dev, int* n_bounces_dev, void* workspace, size_t workspace_bytes, void* stream) {
    TTUP_REQUIRE(samples_dev && n_saved_dev && n_keep_dev && bounces_dev && n_bounces_dev && workspace, TTUP_EINVAL, "ttup_trajgen_select: null pointer");
    TTUP_REQUIRE(n_seeds >= 0 && mode >= 0 && mode < 6 && (direction == 0 || direction == 1), TTUP_EINVAL, "ttup_trajgen_select: bad mode %d / direction %d", mode, direction);
    TTUP_REQUIRE(workspace_bytes >= ttup_trajgen_workspace_bytes(n_seeds), TTUP_EINVAL, "ttup_trajgen_select: workspace too small");
    if (n_seeds == 0) return TTUP_OK;
    hipStream_t st = (hipStream_t)stream;
    double* times_dev = (double*)((unsigned*)workspace + 624);
    double labels[MAX_SAMPLES];
    const int nl = label_table(labels);
    TTUP_HIP_CHECK(hipMemcpyAsync(times_dev, labels, (size_t)nl * sizeof(double), hipMemcpyHostToDevice, st));
    TTUP_HIP_CHECK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(select_kernel, dim3(cdiv(n_seeds, 64)), dim3(64), 0, st, samples_dev, n_saved_dev, n_seeds, mode, direction, times_dev,
                       n_keep_dev, bounces_dev, n_bounces_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}
