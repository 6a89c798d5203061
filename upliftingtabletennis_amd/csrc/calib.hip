// f4: camera calibration from the 13 table keypoints on the device -- one workgroup per camera.
// Replaces, behind `calibrate_camera` (reference inference/utils.py:312-329), the chain
//   calc_cameramatrices(use_ransac=True)      dataprocessing/regress_cameramatrices.py:199-231
//     DLT start on all visible keypoints      dataprocessing/my_dlt.py:40-162 (normalised DLT, RQ split, sign fixes)
//     regress_cameramatrices_ransac           :129-188: 100 six-point subsets (keypoints 10, 11 always in), each refined from
//                                             the DLT start; inliers = re-projection error < 3.5 px over all visible keypoints;
//                                             first subset with the most inliers wins; refinement on its inliers
//     regress_cameramatrices                  :38-126: 8 parameters (fx, fy, t, extrinsic xyz Euler angles), principal point
//                                             fixed at the image centre, objective = SUM of re-projection distances
// The reference runs 101 SciPy BFGS problems one after the other with finite-difference gradients (about 25 s per call).
// Here thread 0 of the workgroup computes the DLT start (12x12 Jacobi eigen-problem in LDS), then every subset gets its own
// lane: the same objective sum_i sqrt(|e_i|^2 + delta^2) (delta = 1e-7 px keeps it differentiable where a point is fitted
// exactly) is minimised with analytic Jacobians by iteratively reweighted Levenberg-Marquardt steps (each step minimises a
// quadratic majoriser of the objective, so the objective never increases); thread 0 then picks the winner and refines it on
// its inliers.  Same objective, same subsets (the host draws them with numpy's PCG64 exactly as the reference does), same
// selection rule -- a different minimiser: results agree with the reference to the accuracy SciPy's BFGS reaches on this
// non-smooth objective (tests/test_calib_gpu.py states the measured bars).
#include "common.h"
#include <math.h>

using namespace ttup;

namespace {

constexpr int NK = 13, NPAR = 8, MAX_SUB = 128;
constexpr double DELTA = 1e-7, INLIER_PX = 3.5;
constexpr double TL = 2.74, TW = 1.525, TH = 0.76, NETX = 0.1525;
// the 13 table keypoints in world coordinates (uplifting/helper.py:36-50)
__constant__ double TABLE_PTS[NK][3] = {
    {-TL / 2, TW / 2, TH}, {-TL / 2, -TW / 2, TH}, {0.0, TW / 2, TH}, {0.0, -TW / 2, TH}, {TL / 2, TW / 2, TH}, {TL / 2, -TW / 2, TH},
    {0.0, TW / 2 + NETX, TH}, {0.0, -(TW / 2 + NETX), TH}, {0.0, 0.0, TH}, {0.0, TW / 2 + NETX, TH + NETX}, {0.0, -(TW / 2 + NETX), TH + NETX},
    {-TL / 2, 0.0, TH}, {TL / 2, 0.0, TH}};

struct Cam8 { double p[NPAR]; };       // fx, fy, tx, ty, tz, a, b, c

// R = Rz(c) Ry(b) Rx(a)  (scipy Rotation.from_euler('xyz', [a, b, c]): extrinsic rotations about x, then y, then z)
__device__ void rot_and_derivs(double a, double b, double c, double R[3][3], double dRa[3][3], double dRb[3][3], double dRc[3][3]) {
    const double sa = sin(a), ca = cos(a), sb = sin(b), cb = cos(b), sc = sin(c), cc = cos(c);
    R[0][0] = cc * cb; R[0][1] = cc * sb * sa - sc * ca; R[0][2] = cc * sb * ca + sc * sa;
    R[1][0] = sc * cb; R[1][1] = sc * sb * sa + cc * ca; R[1][2] = sc * sb * ca - cc * sa;
    R[2][0] = -sb;     R[2][1] = cb * sa;                R[2][2] = cb * ca;
    if (!dRa) return;
    dRa[0][0] = 0; dRa[0][1] = cc * sb * ca + sc * sa;  dRa[0][2] = -cc * sb * sa + sc * ca;
    dRa[1][0] = 0; dRa[1][1] = sc * sb * ca - cc * sa;  dRa[1][2] = -sc * sb * sa - cc * ca;
    dRa[2][0] = 0; dRa[2][1] = cb * ca;                 dRa[2][2] = -cb * sa;
    dRb[0][0] = -cc * sb; dRb[0][1] = cc * cb * sa; dRb[0][2] = cc * cb * ca;
    dRb[1][0] = -sc * sb; dRb[1][1] = sc * cb * sa; dRb[1][2] = sc * cb * ca;
    dRb[2][0] = -cb;      dRb[2][1] = -sb * sa;     dRb[2][2] = -sb * ca;
    dRc[0][0] = -sc * cb; dRc[0][1] = -sc * sb * sa - cc * ca; dRc[0][2] = -sc * sb * ca + cc * sa;
    dRc[1][0] = cc * cb;  dRc[1][1] = cc * sb * sa - sc * ca;  dRc[1][2] = cc * sb * ca + sc * sa;
    dRc[2][0] = 0;        dRc[2][1] = 0;                       dRc[2][2] = 0;
}

// re-projection error vector of keypoint k (and its 2x8 Jacobian) under parameters th; cx, cy = fixed principal point
__device__ void residual(const Cam8& th, const double R[3][3], const double dRa[3][3], const double dRb[3][3], const double dRc[3][3],
                         int k, double ox, double oy, double cx, double cy, double* eu, double* ev, double Ju[NPAR], double Jv[NPAR]) {
    const double X = TABLE_PTS[k][0], Y = TABLE_PTS[k][1], Z = TABLE_PTS[k][2];
    const double x = R[0][0] * X + R[0][1] * Y + R[0][2] * Z + th.p[2];
    const double y = R[1][0] * X + R[1][1] * Y + R[1][2] * Z + th.p[3];
    const double z = R[2][0] * X + R[2][1] * Y + R[2][2] * Z + th.p[4];
    const double iz = 1.0 / z;
    *eu = th.p[0] * x * iz + cx - ox;
    *ev = th.p[1] * y * iz + cy - oy;
    if (!Ju) return;
    Ju[0] = x * iz; Jv[0] = 0.0; Ju[1] = 0.0; Jv[1] = y * iz;
    Ju[2] = th.p[0] * iz; Jv[2] = 0.0; Ju[3] = 0.0; Jv[3] = th.p[1] * iz;
    Ju[4] = -th.p[0] * x * iz * iz; Jv[4] = -th.p[1] * y * iz * iz;
    const double (*dR[3])[3] = {dRa, dRb, dRc};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double dx = dR[q][0][0] * X + dR[q][0][1] * Y + dR[q][0][2] * Z;
        const double dy = dR[q][1][0] * X + dR[q][1][1] * Y + dR[q][1][2] * Z;
        const double dz = dR[q][2][0] * X + dR[q][2][1] * Y + dR[q][2][2] * Z;
        Ju[5 + q] = th.p[0] * (dx * iz - x * dz * iz * iz);
        Jv[5 + q] = th.p[1] * (dy * iz - y * dz * iz * iz);
    }
}

__device__ double objective(const Cam8& th, const double* kp, unsigned mask, double cx, double cy) {
    double R[3][3];
    rot_and_derivs(th.p[5], th.p[6], th.p[7], R, nullptr, nullptr, nullptr);
    double f = 0.0;
    for (int k = 0; k < NK; ++k) {
        if (!(mask >> k & 1)) continue;
        double eu, ev;
        residual(th, R, nullptr, nullptr, nullptr, k, kp[3 * k], kp[3 * k + 1], cx, cy, &eu, &ev, nullptr, nullptr);
        f += sqrt(eu * eu + ev * ev + DELTA * DELTA);
    }
    return f;
}

// 8x8 Cholesky solve of (H + lambda diag(H)) x = -g; false when not positive definite
__device__ bool solve8(const double H[NPAR][NPAR], const double g[NPAR], double lambda, double x[NPAR], double* pred) {
    double L[NPAR][NPAR];
    for (int i = 0; i < NPAR; ++i) for (int k = 0; k < NPAR; ++k) L[i][k] = H[i][k];
    for (int i = 0; i < NPAR; ++i) L[i][i] += lambda * (H[i][i] > 0.0 ? H[i][i] : 1.0);
    bool ok = true;
    for (int i = 0; i < NPAR; ++i)
        for (int k = 0; k <= i; ++k) {
            double s = L[i][k];
            for (int q = 0; q < k; ++q) s -= L[i][q] * L[k][q];
            if (i == k) { ok = ok && s > 0.0 && s == s; L[i][i] = sqrt(s > 0.0 ? s : 1.0); }
            else L[i][k] = s / L[k][k];
        }
    double y[NPAR];
    for (int i = 0; i < NPAR; ++i) { double s = -g[i]; for (int q = 0; q < i; ++q) s -= L[i][q] * y[q]; y[i] = s / L[i][i]; }
    for (int i = NPAR - 1; i >= 0; --i) { double s = y[i]; for (int q = i + 1; q < NPAR; ++q) s -= L[q][i] * x[q]; x[i] = s / L[i][i]; }
    double pr = 0.0;
    for (int i = 0; i < NPAR; ++i) pr += x[i] * (0.5 * lambda * (H[i][i] > 0.0 ? H[i][i] : 1.0) * x[i] - 0.5 * g[i]);
    *pred = pr;                       // decrease of the quadratic majoriser f + g'x + x'Hx/2 along the damped step
    return ok;
}

// minimise sum_k sqrt(|e_k|^2 + delta^2) over the keypoints in `mask`, starting at th (updated in place); returns iterations
__device__ int refine(Cam8& th, const double* kp, unsigned mask, double cx, double cy, int max_iter) {
    double f = objective(th, kp, mask, cx, cy);
    double lambda = 1e-4, nu = 2.0;
    int it = 0;
    for (int round = 0; round < 3 * max_iter && it < max_iter; ++round) {
        double R[3][3], dRa[3][3], dRb[3][3], dRc[3][3];
        rot_and_derivs(th.p[5], th.p[6], th.p[7], R, dRa, dRb, dRc);
        double H[NPAR][NPAR], g[NPAR];
        for (int i = 0; i < NPAR; ++i) { g[i] = 0.0; for (int k = 0; k < NPAR; ++k) H[i][k] = 0.0; }
        for (int k = 0; k < NK; ++k) {
            if (!(mask >> k & 1)) continue;
            double eu, ev, Ju[NPAR], Jv[NPAR];
            residual(th, R, dRa, dRb, dRc, k, kp[3 * k], kp[3 * k + 1], cx, cy, &eu, &ev, Ju, Jv);
            const double w = 1.0 / sqrt(eu * eu + ev * ev + DELTA * DELTA);
            for (int i = 0; i < NPAR; ++i) {
                g[i] += w * (Ju[i] * eu + Jv[i] * ev);
                for (int q = 0; q <= i; ++q) H[i][q] += w * (Ju[i] * Ju[q] + Jv[i] * Jv[q]);
            }
        }
        for (int i = 0; i < NPAR; ++i) for (int q = i + 1; q < NPAR; ++q) H[i][q] = H[q][i];
        double dx[NPAR], pred;
        const bool pd = solve8(H, g, lambda, dx, &pred);
        Cam8 tr = th;
        double step2 = 0.0, scale2 = 0.0;
        for (int i = 0; i < NPAR; ++i) { tr.p[i] += pd ? dx[i] : 0.0; step2 += dx[i] * dx[i]; scale2 += tr.p[i] * tr.p[i]; }
        const double fn = pd ? objective(tr, kp, mask, cx, cy) : f;
        const double rho = (pd && pred > 0.0 && fn == fn) ? (f - fn) / pred : -1.0;
        const bool tiny = step2 <= 1e-26 * (scale2 + 1e-12);
        if (rho > 0.0) {
            const double rel = (f - fn) / (f > 0.0 ? f : 1.0);
            th = tr; f = fn; ++it;
            const double q = 2.0 * rho - 1.0, shrink = 1.0 - q * q * q;
            lambda *= shrink > 1.0 / 3.0 ? shrink : 1.0 / 3.0;
            lambda = lambda > 1e-14 ? lambda : 1e-14;
            nu = 2.0;
            if (rel < 1e-15 || tiny) break;
        } else {
            lambda *= nu; nu *= 2.0;
            if (lambda > 1e14 || (pd && tiny)) break;
        }
    }
    return it;
}

__device__ int count_inliers(const Cam8& th, const double* kp, unsigned vis, double cx, double cy, unsigned* inl) {
    double R[3][3];
    rot_and_derivs(th.p[5], th.p[6], th.p[7], R, nullptr, nullptr, nullptr);
    int n = 0; unsigned m = 0;
    for (int k = 0; k < NK; ++k) {
        if (!(vis >> k & 1)) continue;
        double eu, ev;
        residual(th, R, nullptr, nullptr, nullptr, k, kp[3 * k], kp[3 * k + 1], cx, cy, &eu, &ev, nullptr, nullptr);
        if (sqrt(eu * eu + ev * ev) < INLIER_PX) { ++n; m |= 1u << k; }
    }
    *inl = m;
    return n;
}

// ---- DLT start (my_dlt.py): thread 0, matrices in LDS
__device__ bool dlt_start(const double* kp, unsigned vis, double (*M)[12], double (*V)[12], Cam8* out) {
    double m3[3] = {0, 0, 0}, s3[3] = {0, 0, 0}, m2[2] = {0, 0}, s2[2] = {0, 0};
    int n = 0;
    for (int k = 0; k < NK; ++k) if (vis >> k & 1) { ++n; for (int d = 0; d < 3; ++d) m3[d] += TABLE_PTS[k][d]; m2[0] += kp[3 * k]; m2[1] += kp[3 * k + 1]; }
    for (int d = 0; d < 3; ++d) m3[d] /= n;
    m2[0] /= n; m2[1] /= n;
    for (int k = 0; k < NK; ++k) if (vis >> k & 1) {
        for (int d = 0; d < 3; ++d) s3[d] += (TABLE_PTS[k][d] - m3[d]) * (TABLE_PTS[k][d] - m3[d]);
        s2[0] += (kp[3 * k] - m2[0]) * (kp[3 * k] - m2[0]); s2[1] += (kp[3 * k + 1] - m2[1]) * (kp[3 * k + 1] - m2[1]);
    }
    for (int d = 0; d < 3; ++d) { s3[d] = sqrt(s3[d] / n); if (s3[d] == 0.0) s3[d] = 1e-10; }
    for (int d = 0; d < 2; ++d) { s2[d] = sqrt(s2[d] / n); if (s2[d] == 0.0) s2[d] = 1e-10; }
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) { M[i][j] = 0.0; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int k = 0; k < NK; ++k) if (vis >> k & 1) {
        const double X = (TABLE_PTS[k][0] - m3[0]) / s3[0], Y = (TABLE_PTS[k][1] - m3[1]) / s3[1], Z = (TABLE_PTS[k][2] - m3[2]) / s3[2];
        const double x = (kp[3 * k] - m2[0]) / s2[0], y = (kp[3 * k + 1] - m2[1]) / s2[1];
        const double r0[12] = {-X, -Y, -Z, -1, 0, 0, 0, 0, x * X, x * Y, x * Z, x}, r1[12] = {0, 0, 0, 0, -X, -Y, -Z, -1, y * X, y * Y, y * Z, y};
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) M[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
    }
    // cyclic Jacobi on the 12x12 normal matrix: the eigenvector of its smallest eigenvalue = last right singular vector of A
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < 12; ++i) for (int j = i + 1; j < 12; ++j) off += M[i][j] * M[i][j];
        if (off < 1e-60) break;
        for (int p = 0; p < 11; ++p) for (int q = p + 1; q < 12; ++q) {
            if (M[p][q] == 0.0) continue;
            const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 12; ++k) { const double a = M[k][p], b = M[k][q]; M[k][p] = c * a - s * b; M[k][q] = s * a + c * b; }
            for (int k = 0; k < 12; ++k) { const double a = M[p][k], b = M[q][k]; M[p][k] = c * a - s * b; M[q][k] = s * a + c * b; }
            for (int k = 0; k < 12; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
        }
    }
    int best = 0;
    for (int i = 1; i < 12; ++i) if (M[i][i] < M[best][best]) best = i;
    double Pn[3][4], P[3][4];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) Pn[i][j] = V[4 * i + j][best];
    // P = inv(T2) Pn T3,  T3 = [diag(1/s3) | -m3/s3], inv(T2) = [[s2x, 0, m2x], [0, s2y, m2y], [0, 0, 1]]
    double Q[3][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Q[i][j] = Pn[i][j] / s3[j];
        Q[i][3] = Pn[i][3] - (Pn[i][0] * m3[0] / s3[0] + Pn[i][1] * m3[1] / s3[1] + Pn[i][2] * m3[2] / s3[2]);
    }
    for (int j = 0; j < 4; ++j) { P[0][j] = s2[0] * Q[0][j] + m2[0] * Q[2][j]; P[1][j] = s2[1] * Q[1][j] + m2[1] * Q[2][j]; P[2][j] = Q[2][j]; }
    double nrm = P[2][3];
    if (nrm == 0.0) { nrm = 0.0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) nrm += P[i][j] * P[i][j]; nrm = sqrt(nrm); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) P[i][j] /= nrm;
    // RQ split of the left 3x3 block with a positive diagonal (what scipy's rq + the reference's sign fix give)
    double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, R[3][3];
    auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    K[2][2] = sqrt(dot3(P[2], P[2]));
    if (!(K[2][2] > 0.0)) return false;
    for (int j = 0; j < 3; ++j) R[2][j] = P[2][j] / K[2][2];
    K[1][2] = dot3(P[1], R[2]);
    double v[3];
    for (int j = 0; j < 3; ++j) v[j] = P[1][j] - K[1][2] * R[2][j];
    K[1][1] = sqrt(dot3(v, v));
    if (!(K[1][1] > 0.0)) return false;
    for (int j = 0; j < 3; ++j) R[1][j] = v[j] / K[1][1];
    K[0][2] = dot3(P[0], R[2]); K[0][1] = dot3(P[0], R[1]);
    for (int j = 0; j < 3; ++j) v[j] = P[0][j] - K[0][2] * R[2][j] - K[0][1] * R[1][j];
    K[0][0] = sqrt(dot3(v, v));
    if (!(K[0][0] > 0.0)) return false;
    for (int j = 0; j < 3; ++j) R[0][j] = v[j] / K[0][0];
    const double k22 = K[2][2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K[i][j] /= k22;
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) + R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0) for (int i = 0; i < 3; ++i) R[i][2] = -R[i][2];                // my_dlt.py:131-132 flips the third COLUMN
    // t = K^-1 p4 (K upper triangular, K[2][2] = 1)
    double t[3];
    t[2] = P[2][3] / K[2][2];
    t[1] = (P[1][3] - K[1][2] * t[2]) / K[1][1];
    t[0] = (P[0][3] - K[0][1] * t[1] - K[0][2] * t[2]) / K[0][0];
    // extrinsic xyz Euler angles of R = Rz(c) Ry(b) Rx(a), wrapped to [-pi, pi) (regress_cameramatrices.py:84-91)
    double sb = -R[2][0];
    sb = sb > 1.0 ? 1.0 : (sb < -1.0 ? -1.0 : sb);
    const double PI = 3.141592653589793238462643383279502884;
    double ang[3] = {atan2(R[2][1], R[2][2]), asin(sb), atan2(R[1][0], R[0][0])};
    for (int q = 0; q < 3; ++q) { double w = fmod(ang[q] + PI, 2.0 * PI); if (w < 0) w += 2.0 * PI; ang[q] = w - PI; }
    out->p[0] = K[0][0]; out->p[1] = K[1][1]; out->p[2] = t[0]; out->p[3] = t[1]; out->p[4] = t[2];
    out->p[5] = ang[0]; out->p[6] = ang[1]; out->p[7] = ang[2];
    for (int i = 0; i < NPAR; ++i) if (!(out->p[i] == out->p[i])) return false;
    return true;
}

struct CalibArgs {
    const double* kp; const int* subsets; int B, NS, W, H, max_iter;
    double* mint; double* mext; int* n_inliers; int* status; double* start;
};

__global__ __launch_bounds__(MAX_SUB) void calib_kernel(CalibArgs a) {
    __shared__ double sM[12][12], sV[12][12];
    __shared__ double s_kp[NK * 3];
    __shared__ Cam8 s_start;
    __shared__ Cam8 s_res[MAX_SUB];
    __shared__ int s_cnt[MAX_SUB];
    __shared__ unsigned s_vis;
    __shared__ int s_ok;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < NK * 3) s_kp[tid] = a.kp[(size_t)b * NK * 3 + tid];
    __syncthreads();
    const double cx = (double)(a.W / 2), cy = (double)(a.H / 2);
    if (tid == 0) {
        unsigned vis = 0; int n = 0;
        for (int k = 0; k < NK; ++k) if (s_kp[3 * k + 2] == 1.0) { vis |= 1u << k; ++n; }
        s_vis = vis;
        s_ok = n >= 6 ? (dlt_start(s_kp, vis, sM, sV, &s_start) ? 1 : -2) : -1;
    }
    __syncthreads();
    if (s_ok < 0) {
        if (tid == 0) { a.status[b] = s_ok; a.n_inliers[b] = 0; }
        return;
    }
    const unsigned vis = s_vis;
    if (tid < a.NS) {
        unsigned mask = 0;
        if (vis >> 9 & 1) mask |= 1u << 9;                // keypoints 10 and 11 (the net posts' tops) are in every subset
        if (vis >> 10 & 1) mask |= 1u << 10;
        for (int q = 0; q < 4; ++q) { const int key = a.subsets[((size_t)b * a.NS + tid) * 4 + q]; if (key >= 1 && key <= NK) mask |= 1u << (key - 1); }
        Cam8 th = s_start;
        refine(th, s_kp, mask & vis, cx, cy, a.max_iter);
        unsigned inl;
        s_cnt[tid] = count_inliers(th, s_kp, vis, cx, cy, &inl);
        s_res[tid] = th;
    }
    __syncthreads();
    if (tid != 0) return;
    int best = 0;
    for (int s = 1; s < a.NS; ++s) if (s_cnt[s] > s_cnt[best]) best = s;          // first subset with the most inliers
    Cam8 th = s_res[best];
    unsigned inl;
    const int n_in = count_inliers(th, s_kp, vis, cx, cy, &inl);
    if (n_in > 0) refine(th, s_kp, inl, cx, cy, a.max_iter);
    double R[3][3];
    rot_and_derivs(th.p[5], th.p[6], th.p[7], R, nullptr, nullptr, nullptr);
    double* mi = a.mint + (size_t)b * 12;
    double* me = a.mext + (size_t)b * 16;
    const double MI[12] = {th.p[0], 0, cx, 0, 0, th.p[1], cy, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; ++i) mi[i] = MI[i];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) me[4 * i + j] = R[i][j]; me[4 * i + 3] = th.p[2 + i]; }
    me[12] = 0; me[13] = 0; me[14] = 0; me[15] = 1;
    a.n_inliers[b] = n_in;
    a.status[b] = n_in > 0 ? 0 : -3;
    if (a.start) for (int i = 0; i < NPAR; ++i) a.start[(size_t)b * NPAR + i] = s_start.p[i];
}

}  // namespace

extern "C" int ttup_calib_forward(const double* keypoints_dev, const int* subsets_dev, int batch, int n_subsets, int img_w, int img_h, int max_iter,
                                  double* mint_dev, double* mext_dev, int* n_inliers_dev, int* status_dev, double* start_dev, void* stream) {
    TTUP_REQUIRE(keypoints_dev && subsets_dev && mint_dev && mext_dev && n_inliers_dev && status_dev, TTUP_EINVAL, "ttup_calib_forward: null pointer");
    TTUP_REQUIRE(batch >= 0 && n_subsets >= 1 && n_subsets <= MAX_SUB && img_w > 0 && img_h > 0 && max_iter > 0, TTUP_EINVAL,
                 "ttup_calib_forward: bad argument (1 <= n_subsets <= %d)", MAX_SUB);
    if (batch == 0) return TTUP_OK;
    CalibArgs a;
    a.kp = keypoints_dev; a.subsets = subsets_dev; a.B = batch; a.NS = n_subsets; a.W = img_w; a.H = img_h; a.max_iter = max_iter;
    a.mint = mint_dev; a.mext = mext_dev; a.n_inliers = n_inliers_dev; a.status = status_dev; a.start = start_dev;
    hipLaunchKernelGGL(calib_kernel, dim3(batch), dim3(MAX_SUB), 0, (hipStream_t)stream, a);
    TTUP_LAUNCH_CHECK();
    return TTUP_OK;
}
