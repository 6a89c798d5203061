// Bounded 4-parameter Gaussian fit of a 3x3 heatmap window, solved the way the reference solves it:
// scipy.optimize.minimize(method='L-BFGS-B') with forward-difference gradients
// (reference call sites balldetection/helper_balldetection.py:84-87, tabledetection/helper_tabledetection.py:118-126).
//
// This is a from-scratch fp64 implementation of the L-BFGS-B iteration (Byrd, Lu, Nocedal, Zhu 1995;
// Morales & Nocedal 2011 subspace projection) specialised to n = 4, m = 10:
//   * generalized Cauchy point along the projected steepest-descent path,
//   * subspace minimisation over the free variables followed by the projection / backtracking rule,
//   * More'-Thuente line search (dcsrch/dcstep, ftol 1e-3, gtol 0.9, xtol 0.1, at most 20 trials),
//   * BFGS memory of the last 10 (s,y) pairs, update skipped when s'y <= eps*(-g'd),
//   * stops: projected gradient <= 1e-5, (f_old-f)/max(|f_old|,|f|,1) <= 2.2204460492503131e-09.
// Because n = 4 the limited-memory matrix B = theta*I - W M W' is formed densely by replaying the stored
// pairs through the BFGS recursion started at theta*I (identical in exact arithmetic, Byrd-Nocedal-Schnabel
// 1994, Thm 2.3), so the Cauchy search and the subspace solve are plain 4x4 dense algebra.
// Gradients are scipy's '2-point' scheme: h = 1e-8 absolute, flipped when x+h leaves the box
// (scipy/optimize/_numdiff.py:_adjust_scheme_to_bounds), df/dx with dx recomputed as (x+h)-x.
//
// The same source compiles for gfx950 (hipcc, the product) and for the host (g++, test infrastructure only:
// tests/test_cabi.py builds tests/helpers/host_fit.cpp into a temporary directory and compares it with SciPy).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define TTUP_HD __host__ __device__
#else
#define TTUP_HD
#endif

#pragma clang fp contract(off)

namespace ttup {

struct GaussFit {
    double x[4];     // x0, y0, sigma_x, sigma_y
    double f;
    int nit, nfev;
    int success;     // 1 iff scipy would report success (CONVERGENCE)
};

struct GaussProblem {
    double w[9];     // window values (float32 widened), row-major 3x3, index = y*3+x
    double lo[4], hi[4];
    int clamp_sigma; // table variant: max(0.5, sigma) inside the loss
};

// mean((g - w)^2) with numpy's summation order for 9 elements (pairwise block of 8, then the tail)
TTUP_HD inline double gauss_loss(const GaussProblem& P, const double* p) {
    const double x0 = p[0], y0 = p[1];
    double sx = p[2], sy = p[3];
    if (P.clamp_sigma) { sx = sx < 0.5 ? 0.5 : sx; sy = sy < 0.5 ? 0.5 : sy; }
    const double dx2 = 2.0 * (sx * sx), dy2 = 2.0 * (sy * sy);
    double e[9];
    for (int i = 0; i < 9; ++i) {
        const double xx = (double)(i % 3) - x0, yy = (double)(i / 3) - y0;
        const double a = (xx * xx) / dx2, b = (yy * yy) / dy2;
        const double gss = exp(-(a + b));
        const double r = gss - P.w[i];
        e[i] = r * r;
    }
    double s = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    s += e[8];
    return s / 9.0;
}

struct LbState {
    double S[10][4], Y[10][4];
    int col;
    double theta;
};

TTUP_HD inline void lb_eval(const GaussProblem& P, const double* x, double* f, double* g, int* nfev) {
    const double f0 = gauss_loss(P, x);
    double x1[4] = {x[0], x[1], x[2], x[3]};
    for (int i = 0; i < 4; ++i) {
        double h = 1e-8;
        const double lower = x[i] - P.lo[i], upper = P.hi[i] - x[i];
        const double xp = x[i] + h;
        const bool violated = xp < P.lo[i] || xp > P.hi[i];
        const double mx = lower > upper ? lower : upper;
        const bool fitting = fabs(h) <= mx;
        if (violated && fitting) h = -h;
        else if (!fitting) h = (upper >= lower) ? upper : -lower;
        x1[i] = x[i] + h;
        const double dx = x1[i] - x[i];
        g[i] = (gauss_loss(P, x1) - f0) / dx;
        x1[i] = x[i];
    }
    *f = f0;
    *nfev += 5;
}

TTUP_HD inline double lb_projgr(const GaussProblem& P, const double* x, const double* g) {
    double nrm = 0.0;
    for (int i = 0; i < 4; ++i) {
        double gi = g[i];
        if (gi < 0.0) { const double t = x[i] - P.hi[i]; gi = t > gi ? t : gi; }
        else { const double t = x[i] - P.lo[i]; gi = t < gi ? t : gi; }
        const double a = fabs(gi);
        nrm = a > nrm ? a : nrm;
    }
    return nrm;
}

// B = theta*I replayed through the stored pairs (oldest first)
TTUP_HD inline void lb_dense_b(const LbState& st, double B[4][4]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] = i == j ? st.theta : 0.0;
    for (int k = 0; k < st.col; ++k) {
        const double* s = st.S[k]; const double* y = st.Y[k];
        double Bs[4], sBs = 0.0, ys = 0.0;
        for (int i = 0; i < 4; ++i) { double a = 0.0; for (int j = 0; j < 4; ++j) a += B[i][j] * s[j]; Bs[i] = a; }
        for (int i = 0; i < 4; ++i) { sBs += s[i] * Bs[i]; ys += y[i] * s[i]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] += y[i] * y[j] / ys - Bs[i] * Bs[j] / sBs;
    }
}

// generalized Cauchy point; iwhere: 1/2 = fixed at lower/upper bound, 0 = free, -3 = free with zero gradient
TTUP_HD inline void lb_cauchy(const GaussProblem& P, const double* x, const double* g, const double B[4][4], double theta,
                              double sbgnrm, double* xcp, int* iwhere) {
    const double epsmch = 2.220446049250313e-16;
    for (int i = 0; i < 4; ++i) xcp[i] = x[i];
    if (sbgnrm <= 0.0) return;
    double d[4], tbrk[4];
    bool hasbrk[4];
    int nbreak = 0, nfree_nobrk = 0;
    bool bnded = true;
    double f1 = 0.0;
    for (int i = 0; i < 4; ++i) {
        const double neggi = -g[i];
        const double tl = x[i] - P.lo[i], tu = P.hi[i] - x[i];
        const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
        iwhere[i] = 0;
        if (xlower) { if (neggi <= 0.0) iwhere[i] = 1; }
        else if (xupper) { if (neggi >= 0.0) iwhere[i] = 2; }
        else if (fabs(neggi) <= 0.0) iwhere[i] = -3;
        hasbrk[i] = false; tbrk[i] = 0.0;
        if (iwhere[i] != 0) { d[i] = 0.0; continue; }
        d[i] = neggi;
        f1 -= neggi * neggi;
        if (neggi < 0.0) { hasbrk[i] = true; tbrk[i] = tl / (-neggi); ++nbreak; }
        else if (neggi > 0.0) { hasbrk[i] = true; tbrk[i] = tu / neggi; ++nbreak; }
        else { ++nfree_nobrk; if (fabs(neggi) > 0.0) bnded = false; }
    }
    bool any_d = false;
    for (int i = 0; i < 4; ++i) any_d |= (iwhere[i] == 0);
    if (nbreak == 0 && !any_d) return;
    // f2 = d'Bd
    double f2 = 0.0;
    for (int i = 0; i < 4; ++i) { double a = 0.0; for (int j = 0; j < 4; ++j) a += B[i][j] * d[j]; f2 += d[i] * a; }
    const double f2_org = -theta * f1;
    double dtm = -f1 / f2, tsum = 0.0, tj = 0.0;
    double z[4] = {0.0, 0.0, 0.0, 0.0};     // xcp - x accumulated so far
    int nleft = nbreak;
    bool done_all = false;
    while (nleft > 0) {
        int ibp = -1;
        for (int i = 0; i < 4; ++i) if (hasbrk[i] && (ibp < 0 || tbrk[i] < tbrk[ibp])) ibp = i;
        const double tj0 = tj;
        tj = tbrk[ibp];
        const double dt = tj - tj0;
        if (dtm < dt) break;
        tsum += dt; --nleft; hasbrk[ibp] = false;
        for (int i = 0; i < 4; ++i) z[i] += dt * d[i];
        const double dibp = d[ibp];
        d[ibp] = 0.0;
        if (dibp > 0.0) { z[ibp] = P.hi[ibp] - x[ibp]; xcp[ibp] = P.hi[ibp]; iwhere[ibp] = 2; }
        else { z[ibp] = P.lo[ibp] - x[ibp]; xcp[ibp] = P.lo[ibp]; iwhere[ibp] = 1; }
        if (nleft == 0 && nbreak == 4) { done_all = true; break; }
        // derivative information of the next segment: f1 = g'd + z'Bd, f2 = d'Bd
        double Bd[4];
        for (int i = 0; i < 4; ++i) { double a = 0.0; for (int j = 0; j < 4; ++j) a += B[i][j] * d[j]; Bd[i] = a; }
        f1 = 0.0; f2 = 0.0;
        for (int i = 0; i < 4; ++i) { f1 += g[i] * d[i] + z[i] * Bd[i]; f2 += d[i] * Bd[i]; }
        const double floor2 = epsmch * f2_org;
        f2 = f2 > floor2 ? f2 : floor2;
        if (nleft > 0) dtm = -f1 / f2;
        else if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
        else dtm = -f1 / f2;
    }
    if (done_all) return;
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    for (int i = 0; i < 4; ++i) if (d[i] != 0.0) xcp[i] = x[i] + tsum * d[i];
    (void)nfree_nobrk;
}

// solve A d = r for the nf x nf leading system (symmetric positive definite in exact arithmetic); returns false on breakdown
TTUP_HD inline bool lb_solve(double A[4][4], double* r, int nf) {
    for (int k = 0; k < nf; ++k) {
        int piv = k; double best = fabs(A[k][k]);
        for (int i = k + 1; i < nf; ++i) if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
        if (!(best > 0.0)) return false;
        if (piv != k) { for (int j = 0; j < nf; ++j) { const double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; } const double t = r[k]; r[k] = r[piv]; r[piv] = t; }
        for (int i = k + 1; i < nf; ++i) {
            const double m = A[i][k] / A[k][k];
            for (int j = k; j < nf; ++j) A[i][j] -= m * A[k][j];
            r[i] -= m * r[k];
        }
    }
    for (int k = nf - 1; k >= 0; --k) { double a = r[k]; for (int j = k + 1; j < nf; ++j) a -= A[k][j] * r[j]; r[k] = a / A[k][k]; }
    return true;
}

// subspace minimisation (direct primal method) + Morales-Nocedal projection; z enters as xcp, leaves as the new target
TTUP_HD inline bool lb_subsm(const GaussProblem& P, const double* x, const double* g, const double B[4][4], const int* iwhere, double* z) {
    int ind[4], nf = 0;
    for (int i = 0; i < 4; ++i) if (iwhere[i] <= 0) ind[nf++] = i;
    if (nf == 0) return true;
    double A[4][4], d[4];
    for (int a = 0; a < nf; ++a) {
        const int k = ind[a];
        double bz = 0.0;
        for (int j = 0; j < 4; ++j) bz += B[k][j] * (z[j] - x[j]);
        d[a] = -(g[k] + bz);
        for (int b = 0; b < nf; ++b) A[a][b] = B[k][ind[b]];
    }
    if (!lb_solve(A, d, nf)) return false;
    double xp[4] = {z[0], z[1], z[2], z[3]};
    bool hit = false;
    for (int a = 0; a < nf; ++a) {
        const int k = ind[a];
        double v = z[k] + d[a];
        v = v < P.lo[k] ? P.lo[k] : v;
        v = v > P.hi[k] ? P.hi[k] : v;
        z[k] = v;
        if (v == P.lo[k] || v == P.hi[k]) hit = true;
    }
    if (!hit) return true;
    double ddp = 0.0;
    for (int i = 0; i < 4; ++i) ddp += (z[i] - x[i]) * g[i];
    if (ddp <= 0.0) return true;
    // projected point is not a descent direction: fall back to truncating the Newton step at the first bound
    for (int i = 0; i < 4; ++i) z[i] = xp[i];
    double alpha = 1.0, temp1 = 1.0;
    int ibd = -1;
    for (int a = 0; a < nf; ++a) {
        const int k = ind[a];
        const double dk = d[a];
        if (dk < 0.0) { const double t2 = P.lo[k] - z[k]; if (t2 >= 0.0) temp1 = 0.0; else if (dk * alpha < t2) temp1 = t2 / dk; }
        else if (dk > 0.0) { const double t2 = P.hi[k] - z[k]; if (t2 <= 0.0) temp1 = 0.0; else if (dk * alpha > t2) temp1 = t2 / dk; }
        if (temp1 < alpha) { alpha = temp1; ibd = a; }
    }
    if (alpha < 1.0 && ibd >= 0) {
        const int k = ind[ibd];
        if (d[ibd] > 0.0) { z[k] = P.hi[k]; d[ibd] = 0.0; }
        else if (d[ibd] < 0.0) { z[k] = P.lo[k]; d[ibd] = 0.0; }
    }
    for (int a = 0; a < nf; ++a) z[ind[a]] += alpha * d[a];
    return true;
}

// ---- More'-Thuente line search state (MINPACK-2 dcsrch / dcstep)
struct Dcsrch {
    bool brackt; int stage;
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};
enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2, LS_ERROR = 3 };

TTUP_HD inline double lb_max3(double a, double b, double c) { a = a > b ? a : b; return a > c ? a : c; }

TTUP_HD inline void lb_dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                              double fp, double dp, bool& brackt, double stpmin, double stpmax) {
    const double sgnd = dp * (dx / fabs(dx));
    double stpf;
    if (fp > fx) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = lb_max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
        const double stpc = stx + r * (stp - stx);
        const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        stpf = fabs(stpc - stx) <= fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = lb_max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
        const double stpc = stp + r * (stx - stp);
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
        brackt = true;
    } else if (fabs(dp) < fabs(dx)) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = lb_max3(fabs(theta), fabs(dx), fabs(dp));
        double arg = (theta / s) * (theta / s) - (dx / s) * (dp / s);
        arg = arg > 0.0 ? arg : 0.0;
        double gamma = s * sqrt(arg);
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            stpf = fabs(stpc - stp) < fabs(stpq - stp) ? stpc : stpq;
            const double lim = stp + 0.66 * (sty - stp);
            if (stp > stx) stpf = lim < stpf ? lim : stpf;
            else stpf = lim > stpf ? lim : stpf;
        } else {
            stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
            stpf = stpf > stpmax ? stpmax : stpf;
            stpf = stpf < stpmin ? stpmin : stpf;
        }
    } else {
        if (brackt) {
            const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            const double s = lb_max3(fabs(theta), fabs(dy), fabs(dp));
            double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
            stpf = stp + r * (sty - stp);
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > fx) { sty = stp; fy = fp; dy = dp; }
    else {
        if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}

// one dcsrch call; start=true on the first call of a line search
TTUP_HD inline int lb_dcsrch(Dcsrch& s, double& stp, double f, double g, bool start, double stpmin, double stpmax) {
    const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    if (start) {
        if (stp < stpmin || stp > stpmax || g >= 0.0 || stpmax < stpmin) return LS_ERROR;
        s.brackt = false; s.stage = 1; s.finit = f; s.ginit = g; s.gtest = ftol * s.ginit;
        s.width = stpmax - stpmin; s.width1 = s.width / p5;
        s.stx = 0.0; s.fx = s.finit; s.gx = s.ginit; s.sty = 0.0; s.fy = s.finit; s.gy = s.ginit;
        s.stmin = 0.0; s.stmax = stp + xtrapu * stp;
        return LS_FG;
    }
    const double ftest = s.finit + stp * s.gtest;
    if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
    int task = LS_FG;
    if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) task = LS_WARN;
    if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = LS_WARN;
    if (stp == stpmax && f <= ftest && g <= s.gtest) task = LS_WARN;
    if (stp == stpmin && (f > ftest || g >= s.gtest)) task = LS_WARN;
    if (f <= ftest && fabs(g) <= gtol * (-s.ginit)) task = LS_CONV;
    if (task != LS_FG) return task;
    if (s.stage == 1 && f <= s.fx && f > ftest) {
        const double fm = f - stp * s.gtest;
        double fxm = s.fx - s.stx * s.gtest, fym = s.fy - s.sty * s.gtest;
        const double gm = g - s.gtest;
        double gxm = s.gx - s.gtest, gym = s.gy - s.gtest;
        lb_dcstep(s.stx, fxm, gxm, s.sty, fym, gym, stp, fm, gm, s.brackt, s.stmin, s.stmax);
        s.fx = fxm + s.stx * s.gtest; s.fy = fym + s.sty * s.gtest; s.gx = gxm + s.gtest; s.gy = gym + s.gtest;
    } else {
        lb_dcstep(s.stx, s.fx, s.gx, s.sty, s.fy, s.gy, stp, f, g, s.brackt, s.stmin, s.stmax);
    }
    if (s.brackt) {
        if (fabs(s.sty - s.stx) >= p66 * s.width1) stp = s.stx + p5 * (s.sty - s.stx);
        s.width1 = s.width; s.width = fabs(s.sty - s.stx);
    }
    if (s.brackt) { s.stmin = s.stx < s.sty ? s.stx : s.sty; s.stmax = s.stx > s.sty ? s.stx : s.sty; }
    else { s.stmin = stp + xtrapl * (stp - s.stx); s.stmax = stp + xtrapu * (stp - s.stx); }
    stp = stp > stpmax ? stpmax : stp;
    stp = stp < stpmin ? stpmin : stp;
    if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) || (s.brackt && s.stmax - s.stmin <= xtol * s.stmax)) stp = s.stx;
    return LS_FG;
}

TTUP_HD inline void fit_gaussian_lbfgsb(const GaussProblem& P, GaussFit* out) {
    const double epsmch = 2.220446049250313e-16, pgtol = 1e-5, ftol_rel = 2.2204460492503131e-09;
    const int maxls = 20, maxiter = 15000, maxfun = 15000, mmax = 10;
    LbState st; st.col = 0; st.theta = 1.0;
    double x[4], g[4], f;
    for (int i = 0; i < 4; ++i) { double v = 1.0; v = v < P.lo[i] ? P.lo[i] : v; v = v > P.hi[i] ? P.hi[i] : v; x[i] = v; }
    int nfev = 0, iter = 0, success = 0;
    lb_eval(P, x, &f, g, &nfev);
    double sbgnrm = lb_projgr(P, x, g);
    bool running = sbgnrm > pgtol;
    if (!running) success = 1;
    while (running) {
        double B[4][4];
        lb_dense_b(st, B);
        double z[4]; int iwhere[4];
        lb_cauchy(P, x, g, B, st.theta, sbgnrm, z, iwhere);
        bool ok = true;
        if (st.col > 0) ok = lb_subsm(P, x, g, B, iwhere, z);
        if (!ok) { st.col = 0; st.theta = 1.0; continue; }     // singular system: refresh memory, restart iteration
        double d[4], xold[4], gold[4];
        for (int i = 0; i < 4; ++i) { d[i] = z[i] - x[i]; xold[i] = x[i]; gold[i] = g[i]; }
        const double fold = f;
        // ---- line search (lnsrlb)
        double stpmx = 1e10;
        if (iter == 0) stpmx = 1.0;
        else {
            for (int i = 0; i < 4; ++i) {
                const double a1 = d[i];
                if (a1 < 0.0) { const double a2 = P.lo[i] - x[i]; if (a2 >= 0.0) stpmx = 0.0; else if (a1 * stpmx < a2) stpmx = a2 / a1; }
                else if (a1 > 0.0) { const double a2 = P.hi[i] - x[i]; if (a2 <= 0.0) stpmx = 0.0; else if (a1 * stpmx > a2) stpmx = a2 / a1; }
            }
        }
        double stp = 1.0;                      // all four variables are boxed
        double gd = 0.0;
        for (int i = 0; i < 4; ++i) gd += g[i] * d[i];
        const double gdold = gd;
        int info = 0, iback = 0, ifun = 0;
        Dcsrch ls;
        if (gd >= 0.0) info = -4;
        else {
            int task = lb_dcsrch(ls, stp, f, gd, true, 0.0, stpmx);
            if (task == LS_ERROR) info = -3;     // dcsrch input error (e.g. stp > stpmax): treated like a failed search
            while (info == 0 && task == LS_FG) {
                ++ifun; iback = ifun - 1;
                if (iback >= maxls) break;
                if (stp == 1.0) for (int i = 0; i < 4; ++i) x[i] = z[i];
                else for (int i = 0; i < 4; ++i) x[i] = stp * d[i] + xold[i];
                lb_eval(P, x, &f, g, &nfev);
                gd = 0.0;
                for (int i = 0; i < 4; ++i) gd += g[i] * d[i];
                task = lb_dcsrch(ls, stp, f, gd, false, 0.0, stpmx);
            }
        }
        if (info != 0 || iback >= maxls) {
            for (int i = 0; i < 4; ++i) { x[i] = xold[i]; g[i] = gold[i]; }
            f = fold;
            if (st.col == 0) { ++iter; success = 0; break; }          // ABNORMAL_TERMINATION_IN_LNSRCH
            st.col = 0; st.theta = 1.0;                                // refresh memory and restart
            continue;
        }
        ++iter;
        sbgnrm = lb_projgr(P, x, g);
        // driver side (scipy loop): iteration / evaluation limits
        if (iter >= maxiter || nfev > maxfun) { success = 0; break; }
        if (sbgnrm <= pgtol) { success = 1; break; }
        double ddum = lb_max3(fabs(fold), fabs(f), 1.0);
        if ((fold - f) <= ftol_rel * ddum) { success = 1; break; }
        double r[4], rr = 0.0, dr;
        for (int i = 0; i < 4; ++i) { r[i] = g[i] - gold[i]; rr += r[i] * r[i]; }
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; for (int i = 0; i < 4; ++i) d[i] *= stp; ddum = -gdold * stp; }
        if (dr <= epsmch * ddum) continue;                              // skip the update
        if (st.col == mmax) {
            for (int k = 1; k < mmax; ++k) for (int i = 0; i < 4; ++i) { st.S[k - 1][i] = st.S[k][i]; st.Y[k - 1][i] = st.Y[k][i]; }
            st.col = mmax - 1;
        }
        for (int i = 0; i < 4; ++i) { st.S[st.col][i] = d[i]; st.Y[st.col][i] = r[i]; }
        ++st.col;
        st.theta = rr / dr;
    }
    for (int i = 0; i < 4; ++i) out->x[i] = x[i];
    out->f = f; out->nit = iter; out->nfev = nfev; out->success = success;
}

// window (float32[9], row-major 3x3 around the peak) -> sub-pixel offset inside the window, reference semantics
TTUP_HD inline void refine_window(const float* win, int variant, double* x_off, double* y_off, GaussFit* fit_out) {
    GaussProblem P;
    for (int i = 0; i < 9; ++i) P.w[i] = (double)win[i];
    const double smax = variant == 0 ? 50.0 : 3.0;
    P.lo[0] = 0.0; P.lo[1] = 0.0; P.lo[2] = 0.5; P.lo[3] = 0.5;
    P.hi[0] = 3.0; P.hi[1] = 3.0; P.hi[2] = smax; P.hi[3] = smax;
    P.clamp_sigma = variant != 0;
    GaussFit fit;
    fit_gaussian_lbfgsb(P, &fit);
    if (fit.success) { *x_off = fit.x[0]; *y_off = fit.x[1]; }
    else {
        // fallback of the reference (helper_tabledetection.py:130-134): mean position of the window maximum
        float mx = win[0];
        for (int i = 1; i < 9; ++i) mx = win[i] > mx ? win[i] : mx;
        double sx = 0.0, sy = 0.0; int cnt = 0;
        for (int i = 0; i < 9; ++i) if (win[i] == mx) { sx += i % 3; sy += i / 3; ++cnt; }
        *x_off = cnt ? sx / cnt : 1.0; *y_off = cnt ? sy / cnt : 1.0;
    }
    if (fit_out) *fit_out = fit;
}

}  // namespace ttup
