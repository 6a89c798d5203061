// g1 (extension named by BASELINE.json north_star, NO reference counterpart: SURVEY 0.1, 8c): physics-based 2D->3D uplift as a
// batched least-squares fit of drag + Magnus flight dynamics to a detected 2-D track.
//
//   unknowns   p = (r0, v0, w0): position [m], velocity [m/s], spin [rad/s] at the first time stamp (9 numbers)
//   model      free flight of a 40 mm / 2.7 g ball: gravity, quadratic + Stokes drag, Magnus lift, added mass, viscous spin decay
//              -- the smooth part of the generator's force model (csrc/trajgen.hip `accel`, constants restated from the MuJoCo
//              XML of syntheticdataset/helper.py:79-117); contacts are not modelled: one fit = one arc between bounces
//   integrator classical RK4, fp64; every interval between two time stamps is cut into ceil(dt / h_max) equal steps
//   residuals  pinhole projection (Mint . Mext . r) of the state at every valid time stamp minus the observed pixel
//   solver     Levenberg-Marquardt on the Gauss-Newton normal equations (9x9, Marquardt scaling with diag(J'J)); the
//              Jacobian is exact for the discrete flow: the 9 tangent columns d(state)/dp_j are integrated with the SAME RK4
//              through the analytic Jacobian of the acceleration (forward-mode sensitivities, no finite differences)
//
// Mapping: 9 lanes per trajectory (7 trajectories per 64-wide wave, one lane idle): lane j carries the state AND tangent
// column j, so no lane waits for another during the integration; at a time stamp the 2x9 Jacobian block is exchanged with
// wave shuffles and lane j accumulates row j of J'J.  The 9x9 solve is done redundantly by every lane from LDS.
// Validation (tests/test_odefit_gpu.py, oracle/odefit_ref.py): 4th-order convergence under step halving, device == numpy
// oracle to 1e-12, recovery of planted (r0, v0, w0) from their own noiseless projections to 1e-6, agreement with SciPy's
// least_squares.  Parity with the reference is UNPINNED by construction: the reference's uplift is the transformer
// (csrc/uplift.hip), this kernel is never wired in its place.
#include "common.h"
#include <math.h>

using namespace ttup;

namespace {

constexpr double PI = 3.141592653589793238462643383279502884;
constexpr double R_BALL = 0.02, M_BALL = 0.0027, RHO = 1.225, MU_AIR = 0.000018, GRAV = 9.81;
constexpr double C_BLUNT = 0.235, C_MAGNUS = 1.0;
constexpr double VOL = 4.0 / 3.0 * PI * R_BALL * R_BALL * R_BALL, AREA = PI * R_BALL * R_BALL, I_BALL = 0.4 * M_BALL * R_BALL * R_BALL;
constexpr double K_STOKES = 6.0 * PI * MU_AIR * R_BALL, K_QUAD = RHO * C_BLUNT * AREA;
constexpr double K_MAG = C_MAGNUS * RHO * VOL - 0.5 * RHO * VOL;          // cm w x v + ca v x w = (cm - ca) w x v
constexpr double K_SPIN = 8.0 * PI * MU_AIR * R_BALL * R_BALL * R_BALL / I_BALL;
constexpr int NP = 9, GROUPS = 7;           // parameters = lanes per trajectory; trajectories per wave

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// acceleration of the flight model and its directional derivative along (dv, dw)
__device__ __forceinline__ V3 accel(V3 v, V3 w) {
    const double speed = sqrt(dot(v, v));
    V3 a = (-(K_STOKES + K_QUAD * speed) / M_BALL) * v + (K_MAG / M_BALL) * cross(w, v);
    a.z -= GRAV;
    return a;
}
__device__ __forceinline__ V3 daccel(V3 v, V3 w, V3 dv, V3 dw) {
    const double speed = sqrt(dot(v, v));
    const double dspeed = speed > 0.0 ? dot(v, dv) / speed : 0.0;
    return (-(K_STOKES + K_QUAD * speed) / M_BALL) * dv + (-(K_QUAD * dspeed) / M_BALL) * v + (K_MAG / M_BALL) * (cross(dw, v) + cross(w, dv));
}

struct State { V3 r, v, w; };

// one RK4 step of the state and (TAN) of one tangent column
template <bool TAN>
__device__ __forceinline__ void rk4(State& s, State& d, double h) {
    const V3 a1 = accel(s.v, s.w), l1 = (-K_SPIN) * s.w;
    const V3 v2 = s.v + (0.5 * h) * a1, w2 = s.w + (0.5 * h) * l1;
    const V3 a2 = accel(v2, w2), l2 = (-K_SPIN) * w2;
    const V3 v3 = s.v + (0.5 * h) * a2, w3 = s.w + (0.5 * h) * l2;
    const V3 a3 = accel(v3, w3), l3 = (-K_SPIN) * w3;
    const V3 v4 = s.v + h * a3, w4 = s.w + h * l3;
    const V3 a4 = accel(v4, w4), l4 = (-K_SPIN) * w4;
    if (TAN) {
        const V3 da1 = daccel(s.v, s.w, d.v, d.w), dl1 = (-K_SPIN) * d.w;
        const V3 dv2 = d.v + (0.5 * h) * da1, dw2 = d.w + (0.5 * h) * dl1;
        const V3 da2 = daccel(v2, w2, dv2, dw2), dl2 = (-K_SPIN) * dw2;
        const V3 dv3 = d.v + (0.5 * h) * da2, dw3 = d.w + (0.5 * h) * dl2;
        const V3 da3 = daccel(v3, w3, dv3, dw3), dl3 = (-K_SPIN) * dw3;
        const V3 dv4 = d.v + h * da3, dw4 = d.w + h * dl3;
        const V3 da4 = daccel(v4, w4, dv4, dw4), dl4 = (-K_SPIN) * dw4;
        d.r = d.r + (h / 6.0) * (d.v + 2.0 * dv2 + 2.0 * dv3 + dv4);
        d.v = d.v + (h / 6.0) * (da1 + 2.0 * da2 + 2.0 * da3 + da4);
        d.w = d.w + (h / 6.0) * (dl1 + 2.0 * dl2 + 2.0 * dl3 + dl4);
    }
    s.r = s.r + (h / 6.0) * (s.v + 2.0 * v2 + 2.0 * v3 + v4);
    s.v = s.v + (h / 6.0) * (a1 + 2.0 * a2 + 2.0 * a3 + a4);
    s.w = s.w + (h / 6.0) * (l1 + 2.0 * l2 + 2.0 * l3 + l4);
}

__device__ __forceinline__ int substeps(double dt, double hmax) {
    int n = (int)ceil(dt / hmax - 1e-9);
    return n < 1 ? 1 : n;
}

// pixel of a world point and (optionally) the pixel derivative along a world direction; cam = Mext rows 0..2 (12), Mint (9)
__device__ __forceinline__ void project(const double* cam, V3 r, double* u, double* v, const V3* dr, double* du, double* dv) {
    const double xc = cam[0] * r.x + cam[1] * r.y + cam[2] * r.z + cam[3];
    const double yc = cam[4] * r.x + cam[5] * r.y + cam[6] * r.z + cam[7];
    const double zc = cam[8] * r.x + cam[9] * r.y + cam[10] * r.z + cam[11];
    const double* K = cam + 12;
    const double q0 = K[0] * xc + K[1] * yc + K[2] * zc, q1 = K[3] * xc + K[4] * yc + K[5] * zc, q2 = K[6] * xc + K[7] * yc + K[8] * zc;
    *u = q0 / q2; *v = q1 / q2;
    if (dr) {
        const double dx = cam[0] * dr->x + cam[1] * dr->y + cam[2] * dr->z;
        const double dy = cam[4] * dr->x + cam[5] * dr->y + cam[6] * dr->z;
        const double dz = cam[8] * dr->x + cam[9] * dr->y + cam[10] * dr->z;
        const double d0 = K[0] * dx + K[1] * dy + K[2] * dz, d1 = K[3] * dx + K[4] * dy + K[5] * dz, d2 = K[6] * dx + K[7] * dy + K[8] * dz;
        *du = (d0 - *u * d2) / q2; *dv = (d1 - *v * d2) / q2;
    }
}

struct FitArgs {
    const double* obs; const double* times; const double* mask; const double* cam; int cam_per_traj; const double* init;
    int B, T; double hmax; int max_iter; double tol;
    double* params; double* pos3d; double* cost; int* iters;
};

// cost, gradient entry j and row j of J'J at parameters p (lane j = column j); every lane of a group returns the same cost
__device__ __forceinline__ void normal_equations(const FitArgs& a, int traj, int j, int gbase, const double* p, const double* cam, bool active,
                                                 double* cost, double* gj, double* Arow, int* nobs) {
    State s = {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}};
    State d = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // d(state)/dp_j at t0 = unit vector j (written without dynamic indexing so that the state stays in registers)
    d.r = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
    d.v = {j == 3 ? 1.0 : 0.0, j == 4 ? 1.0 : 0.0, j == 5 ? 1.0 : 0.0};
    d.w = {j == 6 ? 1.0 : 0.0, j == 7 ? 1.0 : 0.0, j == 8 ? 1.0 : 0.0};
    double c = 0.0, g = 0.0;
    int n = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) Arow[k] = 0.0;
    const double* tt = a.times + (size_t)traj * a.T;
    double tprev = active ? tt[0] : 0.0;
    for (int i = 0; i < a.T; ++i) {
        bool valid = false;
        if (active) {
            const double t = tt[i];
            if (i > 0) {
                const double dt = t - tprev;
                if (dt > 0.0) {
                    const int ns = substeps(dt, a.hmax);
                    const double h = dt / ns;
                    for (int q = 0; q < ns; ++q) rk4<true>(s, d, h);
                }
                tprev = t;
            }
            valid = !a.mask || a.mask[(size_t)traj * a.T + i] != 0.0;
        }
        double ju = 0.0, jv = 0.0, eu = 0.0, ev = 0.0;
        if (valid) {
            double u, v;
            project(cam, s.r, &u, &v, &d.r, &ju, &jv);
            eu = u - a.obs[((size_t)traj * a.T + i) * 2]; ev = v - a.obs[((size_t)traj * a.T + i) * 2 + 1];
            c += eu * eu + ev * ev; g += ju * eu + jv * ev; ++n;
        }
        // every lane of the wave takes part in the shuffles (inactive / invalid lanes contribute zeros)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const double ku = __shfl(ju, gbase + k, 64), kv = __shfl(jv, gbase + k, 64);
            Arow[k] += ju * ku + jv * kv;
        }
    }
    *cost = c; *gj = g; *nobs = n;
}

// solve (A + lambda * diag(A)) x = -g by Cholesky; A row-major 9x9 in LDS, result in x[9]; false if not positive definite
__device__ __forceinline__ bool solve_lm(const double* A, const double* g, double lambda, double* x, double* pred) {
    double L[NP][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int k = 0; k < NP; ++k) L[i][k] = A[i * NP + k];
#pragma unroll
    for (int i = 0; i < NP; ++i) L[i][i] += lambda * (A[i * NP + i] > 0.0 ? A[i * NP + i] : 1.0);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
#pragma unroll
        for (int k = 0; k <= i; ++k) {
            double sum = L[i][k];
#pragma unroll
            for (int q = 0; q < k; ++q) sum -= L[i][q] * L[k][q];
            if (i == k) { ok = ok && sum > 0.0; L[i][i] = sqrt(sum > 0.0 ? sum : 1.0); }
            else L[i][k] = sum / L[k][k];
        }
    }
    double y[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        double sum = -g[i];
#pragma unroll
        for (int q = 0; q < i; ++q) sum -= L[i][q] * y[q];
        y[i] = sum / L[i][i];
    }
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {
        double sum = y[i];
#pragma unroll
        for (int q = i + 1; q < NP; ++q) sum -= L[q][i] * x[q];
        x[i] = sum / L[i][i];
    }
    // reduction of ||r||^2 that the damped quadratic model predicts for this step: x'(lambda D x - g)
    double pr = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) pr += x[i] * (lambda * (A[i * NP + i] > 0.0 ? A[i * NP + i] : 1.0) * x[i] - g[i]);
    *pred = pr;
    return ok;
}

__global__ __launch_bounds__(64) void odefit_kernel(FitArgs a) {
    __shared__ double sA[GROUPS][NP * NP];
    __shared__ double sg[GROUPS][NP];
    const int lane = threadIdx.x, grp = lane / NP, j = lane % NP;
    const int traj = blockIdx.x * GROUPS + grp;
    const bool active = grp < GROUPS && traj < a.B;
    const int gbase = (grp < GROUPS ? grp : GROUPS - 1) * NP;
    const int g = grp < GROUPS ? grp : 0;
    const double* cam = a.cam + (active && a.cam_per_traj ? (size_t)traj * 21 : 0);
    double p[NP], ptry[NP], Arow[NP], Anew[NP], dp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = active ? a.init[(size_t)traj * NP + k] : 0.0;
    double cost, gj, lambda = 1e-3, nu = 2.0;
    int nobs, it = 0;
    normal_equations(a, traj, j, gbase, p, cam, active, &cost, &gj, Arow, &nobs);
    bool done = !active || nobs == 0;
    // the loop is wave-uniform: finished groups keep taking part in the shuffles and barriers with their last accepted state
    for (int round = 0; round < 4 * a.max_iter; ++round) {
        if (__all(done)) break;
        if (grp < GROUPS) {
#pragma unroll
            for (int k = 0; k < NP; ++k) sA[g][j * NP + k] = Arow[k];
            sg[g][j] = gj;
        }
        __syncthreads();
        double pred;
        const bool pd = solve_lm(sA[g], sg[g], lambda, dp, &pred);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NP; ++k) ptry[k] = p[k] + (pd ? dp[k] : 0.0);
        double cnew, gnew;
        int nn;
        normal_equations(a, traj, j, gbase, ptry, cam, active && !done, &cnew, &gnew, Anew, &nn);
        if (!done) {
            // damping after Nielsen (1999): gain ratio rho = actual / predicted reduction; smooth decrease on good steps,
            // doubling growth on rejected ones -- follows the curved valley of the weakly observable spin without ping-pong
            double step2 = 0.0, scale2 = 0.0;
#pragma unroll
            for (int k = 0; k < NP; ++k) { step2 += dp[k] * dp[k]; scale2 += ptry[k] * ptry[k]; }
            const bool tiny = step2 <= 1e-24 * (scale2 + 1e-12);
            const double rho = (pd && pred > 0.0) ? (cost - cnew) / pred : -1.0;
            if (rho > 0.0) {
#pragma unroll
                for (int k = 0; k < NP; ++k) { p[k] = ptry[k]; Arow[k] = Anew[k]; }
                const double rel = (cost - cnew) / (cost > 0.0 ? cost : 1.0);
                cost = cnew; gj = gnew;
                const double f = 2.0 * rho - 1.0, shrink = 1.0 - f * f * f;
                lambda *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
                lambda = lambda > 1e-15 ? lambda : 1e-15;
                nu = 2.0;
                ++it;
                if (rel < a.tol || tiny || it >= a.max_iter || cost <= 1e-26 * nobs) done = true;
            } else {
                lambda *= nu; nu *= 2.0;
                if (lambda > 1e15 || (pd && tiny)) done = true;
            }
        }
    }
    if (!active) return;
    a.params[(size_t)traj * NP + j] = p[j];
    if (j == 0) {
        if (a.cost) a.cost[traj] = nobs > 0 ? cost / nobs : 0.0;
        if (a.iters) a.iters[traj] = it;
    }
    if (a.pos3d && j == 0) {                          // positions at the time stamps from the accepted parameters
        State s = {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}}, d;
        const double* tt = a.times + (size_t)traj * a.T;
        double tprev = tt[0];
        for (int i = 0; i < a.T; ++i) {
            const double dt = tt[i] - tprev;
            if (i > 0 && dt > 0.0) {
                const int ns = substeps(dt, a.hmax);
                for (int q = 0; q < ns; ++q) rk4<false>(s, d, dt / ns);
                tprev = tt[i];
            }
            double* o = a.pos3d + ((size_t)traj * a.T + i) * 3;
            o[0] = s.r.x; o[1] = s.r.y; o[2] = s.r.z;
        }
    }
}

__global__ void odeint_kernel(const double* params, const double* times, const double* cam, int cam_per_traj, int B, int T, double hmax,
                              double* pos3d, double* px) {
    const int traj = blockIdx.x * blockDim.x + threadIdx.x;
    if (traj >= B) return;
    const double* p = params + (size_t)traj * NP;
    State s = {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}}, d;
    const double* tt = times + (size_t)traj * T;
    const double* cm = cam + (cam_per_traj ? (size_t)traj * 21 : 0);
    double tprev = tt[0];
    for (int i = 0; i < T; ++i) {
        const double dt = tt[i] - tprev;
        if (i > 0 && dt > 0.0) {
            const int ns = substeps(dt, hmax);
            for (int q = 0; q < ns; ++q) rk4<false>(s, d, dt / ns);
            tprev = tt[i];
        }
        if (pos3d) { double* o = pos3d + ((size_t)traj * T + i) * 3; o[0] = s.r.x; o[1] = s.r.y; o[2] = s.r.z; }
        if (px) { double u, v; project(cm, s.r, &u, &v, nullptr, nullptr, nullptr); px[((size_t)traj * T + i) * 2] = u; px[((size_t)traj * T + i) * 2 + 1] = v; }
    }
}

}  // namespace

extern "C" int ttup_odefit_forward(const double* obs_xy_dev, const double* times_dev, const double* mask_dev, const double* cam_dev, int cam_per_traj,
                                   const double* init_dev, int batch, int len, double h_max, int max_iter, double tol,
                                   double* params_dev, double* pos3d_dev, double* cost_dev, int* iters_dev, void* stream) {
    TTUP_REQUIRE(obs_xy_dev && times_dev && cam_dev && init_dev && params_dev, TTUP_EINVAL, "ttup_odefit_forward: null pointer");
    TTUP_REQUIRE(batch >= 0 && len > 0 && h_max > 0.0 && max_iter > 0 && tol >= 0.0, TTUP_EINVAL, "ttup_odefit_forward: bad argument");
    if (batch == 0) return TTUP_OK;
    FitArgs a;
    a.obs = obs_xy_dev; a.times = times_dev; a.mask = mask_dev; a.cam = cam_dev; a.cam_per_traj = cam_per_traj ? 1 : 0; a.init = init_dev;
    a.B = batch; a.T = len; a.hmax = h_max; a.max_iter = max_iter; a.tol = tol;
    a.params = params_dev; a.pos3d = pos3d_dev; a.cost = cost_dev; a.iters = iters_dev;
    hipLaunchKernelGGL(odefit_kernel, dim3(cdiv(batch, GROUPS)), dim3(64), 0, (hipStream_t)stream, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}

extern "C" int ttup_odefit_integrate(const double* params_dev, const double* times_dev, const double* cam_dev, int cam_per_traj, int batch, int len,
                                     double h_max, double* pos3d_dev, double* px_dev, void* stream) {
    TTUP_REQUIRE(params_dev && times_dev && (cam_dev || !px_dev), TTUP_EINVAL, "ttup_odefit_integrate: null pointer");
    TTUP_REQUIRE(batch >= 0 && len > 0 && h_max > 0.0, TTUP_EINVAL, "ttup_odefit_integrate: bad argument");
    if (batch == 0) return TTUP_OK;
    hipLaunchKernelGGL(odeint_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, params_dev, times_dev, cam_dev, cam_per_traj ? 1 : 0,
                       batch, len, h_max, pos3d_dev, px_dev);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}
