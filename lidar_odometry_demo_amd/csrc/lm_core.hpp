// Levenberg-Marquardt policy of one ceres::Solve call of CloudMatcher::align (reference
// src/cloud_matcher.cpp:109-112, 157-158: DENSE_QR, max_num_iterations 4, function_tolerance
// 1e-5, everything else Ceres' defaults), restated on the reduced 6x6 / 6x1 normal equations.
//
// Written as a resumable state machine so that ONE source serves both drivers:
//   * the host loop in align_driver.cpp (lom_align_with_hooks: multi-GPU exchange, CPU tests);
//   * the device-resident loop in match.hip (k_lm), where one lane runs these functions and the
//     whole grid performs the evaluation a step asks for.
// lm_begin / lm_feed return LM_EVAL when they need the sums at S.cand (correspondences fixed),
// LM_DONE when the solve is over; S.x then holds the solution.
//
// What differs from the reference by construction: the reference solves the stacked
// least-squares system by QR over all residual rows, here the same step comes from
// (S A S + D^2 / radius) y = S g on the reduced system.  Both are the minimiser of the same damped
// quadratic; poses agree far inside the 1e-4 m / 1e-4 rad bar.
#pragma once
#include <cfloat>
#include <cmath>

#include "pose_math.hpp"

#if defined(__clang__)
#define LOM_UNROLL _Pragma("unroll")
#else
#define LOM_UNROLL
#endif

namespace lom {

constexpr double kPriorW = 10.0;  // cloud_matcher.cpp:153  diag(0.1,0.1,0.1).inverse()

enum { LM_DONE = 0, LM_EVAL = 1, LM_PROPOSE = 2 };

struct Normal {
    double A[6][6];
    double g[6];
    double cost;
};

struct LmState {
    double x[7];  // current parameters [qw qx qy qz tx ty tz]
    double x_norm;
    double prior_b[3];
    Normal N;  // normal equations at x
    double scale[6], diag[6];
    double radius, decrease_factor;
    int reuse_diag, invalid_run, iter;
    double cand[7];  // the point the pending evaluation is for
    double model_change;
    // results
    int recorded;     // iterations Ceres would list in its summary (iteration 0 included)
    int evaluations;  // residual evaluations spent, the initial one included
    double last_step_norm;
    double cost;
};

// sums (device layout, prior excluded) -> full normal equations incl. the
// NormalPrior on translation (residual 10 (t - t_guess), Jacobian 10 I).
LOM_HD void lm_assemble(const double *s, const double x[7], const double prior_b[3], Normal &n)
{
    int k = 0;
LOM_UNROLL
    for (int a = 0; a < 6; a++) {
LOM_UNROLL
        for (int b = a; b < 6; b++) {
            n.A[a][b] = s[k];
            n.A[b][a] = s[k];
            k++;
        }
    }
LOM_UNROLL
    for (int a = 0; a < 6; a++) n.g[a] = s[21 + a];
    n.cost = s[27];
LOM_UNROLL
    for (int a = 0; a < 3; a++) {
        const double r = kPriorW * (x[4 + a] - prior_b[a]);
        n.A[3 + a][3 + a] += kPriorW * kPriorW;
        n.g[3 + a] += kPriorW * r;
        n.cost += 0.5 * r * r;
    }
}

LOM_HD bool lm_finite(double v) { return fabs(v) <= DBL_MAX; }  // false for NaN too

// M y = b for a symmetric positive definite 6x6 M (lower triangle read).  One reciprocal per
// column; every other "division" is a multiplication by it (divisions are the bulk of this
// function's code and time on the GPU).
LOM_HD bool cholesky_solve6(const double M[6][6], const double b[6], double y[6])
{
    double L[6][6], inv[6];
    bool ok = true;
LOM_UNROLL
    for (int i = 0; i < 6; i++) {
LOM_UNROLL
        for (int j = 0; j <= i; j++) {
            double s = M[i][j];
LOM_UNROLL
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) ok = false;
                L[i][i] = sqrt(s);
                inv[i] = 1.0 / L[i][i];
            } else {
                L[i][j] = s * inv[j];
            }
        }
    }
    if (!ok) return false;
    double z[6];
LOM_UNROLL
    for (int i = 0; i < 6; i++) {
        double s = b[i];
LOM_UNROLL
        for (int k = 0; k < i; k++) s -= L[i][k] * z[k];
        z[i] = s * inv[i];
    }
LOM_UNROLL
    for (int i = 5; i >= 0; i--) {
        double s = z[i];
LOM_UNROLL
        for (int k = i + 1; k < 6; k++) s -= L[k][i] * y[k];
        y[i] = s * inv[i];
    }
LOM_UNROLL
    for (int i = 0; i < 6; i++)
        if (!lm_finite(y[i])) ok = false;
    return ok;
}

LOM_HD double lm_gmax(const Normal &n)
{
    double m = 0.0;
LOM_UNROLL
    for (int c = 0; c < 6; c++) m = fmax(m, fabs(n.g[c]));
    return m;
}

LOM_HD double lm_norm7(const double *v)
{
    double s = 0.0;
LOM_UNROLL
    for (int i = 0; i < 7; i++) s += v[i] * v[i];
    return sqrt(s);
}

// Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy constants
constexpr int kLmMaxIter = 4;
constexpr double kLmFtol = 1e-5, kLmGtol = 1e-10, kLmPtol = 1e-8;
constexpr double kLmMinRelDec = 1e-3, kLmMinDiag = 1e-6, kLmMaxDiag = 1e32, kLmMaxRadius = 1e16;

// from the current iterate: solve for steps until one is worth evaluating (LM_EVAL, S.cand set)
// or the iteration budget / invalid-step budget is used up (LM_DONE)
LOM_HD int lm_propose(LmState &S)
{
    while (S.iter <= kLmMaxIter) {
        double As[6][6], gs[6], M[6][6];
LOM_UNROLL
        for (int a = 0; a < 6; a++) {
            gs[a] = S.N.g[a] * S.scale[a];
LOM_UNROLL
            for (int b = 0; b < 6; b++) As[a][b] = S.N.A[a][b] * S.scale[a] * S.scale[b];
        }
        if (!S.reuse_diag) {
LOM_UNROLL
            for (int c = 0; c < 6; c++) S.diag[c] = fmin(fmax(As[c][c], kLmMinDiag), kLmMaxDiag);
        }
        const double inv_radius = 1.0 / S.radius;
LOM_UNROLL
        for (int a = 0; a < 6; a++) {
LOM_UNROLL
            for (int b = 0; b < 6; b++) M[a][b] = As[a][b];
            M[a][a] += S.diag[a] * inv_radius;
        }
        double y[6], step[6];
        const bool ok = cholesky_solve6(M, gs, y);
        S.reuse_diag = 1;
        double model_change = 0.0;
        if (ok) {
            // -(J s).(r + J s / 2) = -g.s - s^T A s / 2   (scaled space)
            double gsdot = 0.0, quad = 0.0;
LOM_UNROLL
            for (int c = 0; c < 6; c++) step[c] = -y[c];
LOM_UNROLL
            for (int a = 0; a < 6; a++) {
                gsdot += gs[a] * step[a];
                double row = 0.0;
LOM_UNROLL
                for (int b = 0; b < 6; b++) row += As[a][b] * step[b];
                quad += step[a] * row;
            }
            model_change = -gsdot - 0.5 * quad;
        }
        if (!ok || !(model_change > 0.0)) {
            if (++S.invalid_run >= 5) break;
            S.radius /= S.decrease_factor;
            S.decrease_factor *= 2.0;
            S.recorded++;
            S.last_step_norm = 0.0;
            S.iter++;
            continue;
        }
        S.invalid_run = 0;
        double delta[6];
LOM_UNROLL
        for (int c = 0; c < 6; c++) delta[c] = step[c] * S.scale[c];
        manifold_plus(S.x, delta, S.cand);
        S.model_change = model_change;
        return LM_EVAL;
    }
    S.cost = S.N.cost;
    return LM_DONE;
}

// `first`: sums of the evaluation at x (iteration 0).  The *_head functions return LM_DONE or
// LM_PROPOSE (= call lm_propose next); the device loop calls them that way so that the solve code
// exists once in the kernel.
LOM_HD int lm_begin_head(LmState &S, const double *first, const double x[7], const double prior_b[3])
{
LOM_UNROLL
    for (int i = 0; i < 7; i++) S.x[i] = x[i];
LOM_UNROLL
    for (int i = 0; i < 3; i++) S.prior_b[i] = prior_b[i];
    S.radius = 1e4;
    S.decrease_factor = 2.0;
    S.reuse_diag = 0;
    S.invalid_run = 0;
    S.iter = 1;
    S.recorded = 1;
    S.evaluations = 1;
    S.last_step_norm = 0.0;
    S.model_change = 0.0;
LOM_UNROLL
    for (int i = 0; i < 7; i++) S.cand[i] = x[i];
LOM_UNROLL
    for (int c = 0; c < 6; c++) S.diag[c] = 0.0;
    lm_assemble(first, S.x, S.prior_b, S.N);
    // Jacobi scaling, computed once at iteration 0: 1 / (1 + ||column||)
LOM_UNROLL
    for (int c = 0; c < 6; c++) S.scale[c] = 1.0 / (1.0 + sqrt(S.N.A[c][c]));
    S.x_norm = lm_norm7(S.x);
    S.cost = S.N.cost;
    if (lm_gmax(S.N) <= kLmGtol) return LM_DONE;
    return LM_PROPOSE;
}

LOM_HD int lm_begin(LmState &S, const double *first, const double x[7], const double prior_b[3])
{
    const int a = lm_begin_head(S, first, x, prior_b);
    return a == LM_PROPOSE ? lm_propose(S) : a;
}

// `sums`: the evaluation at S.cand.  One evaluation at the candidate serves the accept test (cost)
// and, if accepted, the next iteration (Jacobian) -- the reference evaluates the cost first and the
// Jacobian after acceptance; same numbers, one pass.
LOM_HD int lm_feed_head(LmState &S, const double *sums)
{
    S.evaluations++;
    Normal C;
    lm_assemble(sums, S.cand, S.prior_b, C);
    double d7[7];
LOM_UNROLL
    for (int i = 0; i < 7; i++) d7[i] = S.x[i] - S.cand[i];
    const double sn = lm_norm7(d7);
    S.cost = S.N.cost;
    if (sn <= kLmPtol * (S.x_norm + kLmPtol)) return LM_DONE;         // parameter tolerance: not recorded
    const double cost_change = S.N.cost - C.cost;
    if (fabs(cost_change) <= kLmFtol * S.N.cost) return LM_DONE;      // function tolerance: not recorded
    const double rel_dec = cost_change / S.model_change;
    if (rel_dec > kLmMinRelDec) {
LOM_UNROLL
        for (int i = 0; i < 7; i++) S.x[i] = S.cand[i];
        S.x_norm = lm_norm7(S.x);
        S.N = C;
        const double d3 = 2.0 * rel_dec - 1.0;
        S.radius = S.radius / fmax(1.0 / 3.0, 1.0 - d3 * d3 * d3);
        S.radius = fmin(kLmMaxRadius, S.radius);
        S.decrease_factor = 2.0;
        S.reuse_diag = 0;
    } else {
        S.radius /= S.decrease_factor;
        S.decrease_factor *= 2.0;
        S.reuse_diag = 1;
    }
    S.recorded++;
    S.last_step_norm = sn;
    S.cost = S.N.cost;
    if (lm_gmax(S.N) <= kLmGtol) return LM_DONE;
    S.iter++;
    return LM_PROPOSE;
}

LOM_HD int lm_feed(LmState &S, const double *sums)
{
    const int a = lm_feed_head(S, sums);
    return a == LM_PROPOSE ? lm_propose(S) : a;
}

}  // namespace lom
