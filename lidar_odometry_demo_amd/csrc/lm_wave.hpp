// lm_core.hpp's Levenberg-Marquardt policy executed by ONE WAVE of k_lm (match.hip): the device-resident solve of
// CloudMatcher::align (reference src/cloud_matcher.cpp:157-158).  Device code only; shared with
// tools/microbench/policy.hip, which times it in isolation and checks it against lm_core.hpp's serial results.
#pragma once
#include <hip/hip_runtime.h>

#include "lm_core.hpp"
#include "pose_math.hpp"

namespace lom {

// The step runs on ONE wave while the rest of the grid waits for it, at one instruction per 5 (independent) to 9
// (dependent) cycles (tools/microbench/exec_skip.hip; the SIMD does not skip the 16-lane passes of inactive lanes):
// its time is its instruction count.  Inside this header a * b + c contracts to one FMA (the library is built with
// -ffp-contract=off for the f32 search arithmetic, which must round like the reference's x86 build; nothing here
// is compared bit for bit with the host's lm_core.hpp -- the contract is "same decisions, rounding-level
// differences", checked by tools/microbench/policy.hip and the parity tests).
#pragma clang fp contract(fast)

__device__ __forceinline__ double lmw_norm7(const double *v)
{
    double s = v[0] * v[0];
#pragma unroll
    for (int i = 1; i < 7; i++) s += v[i] * v[i];
    return sqrt(s);
}

// pose_math.hpp's sinc_cos for small angles (the two Horner chains as FMAs, interleaved by the scheduler)
__device__ __forceinline__ void lmw_sinc_cos(double a, double &sinc, double &c)
{
    if (a < 0.5) {
        const double z = a * a;
        double s = 1.0 / 51090942171709440000.0;  // 1/21!
        s = 1.0 / 121645100408832000.0 - z * s;   // 1/19!
        s = 1.0 / 355687428096000.0 - z * s;      // 1/17!
        s = 1.0 / 1307674368000.0 - z * s;        // 1/15!
        s = 1.0 / 6227020800.0 - z * s;           // 1/13!
        s = 1.0 / 39916800.0 - z * s;             // 1/11!
        s = 1.0 / 362880.0 - z * s;               // 1/9!
        s = 1.0 / 5040.0 - z * s;                 // 1/7!
        s = 1.0 / 120.0 - z * s;                  // 1/5!
        s = 1.0 / 6.0 - z * s;                    // 1/3!
        sinc = 1.0 - z * s;
        double k = 1.0 / 2432902008176640000.0;   // 1/20!
        k = 1.0 / 6402373705728000.0 - z * k;     // 1/18!
        k = 1.0 / 20922789888000.0 - z * k;       // 1/16!
        k = 1.0 / 87178291200.0 - z * k;          // 1/14!
        k = 1.0 / 479001600.0 - z * k;            // 1/12!
        k = 1.0 / 3628800.0 - z * k;              // 1/10!
        k = 1.0 / 40320.0 - z * k;                // 1/8!
        k = 1.0 / 720.0 - z * k;                  // 1/6!
        k = 1.0 / 24.0 - z * k;                   // 1/4!
        k = 1.0 / 2.0 - z * k;                    // 1/2!
        c = 1.0 - z * k;
    } else {
        sinc = sin(a) / a;
        c = cos(a);
    }
}

// ---- lm_core.hpp's policy, executed by one wave ---------------------------------------------
// Same decisions, same operation order as lm_begin_head / lm_feed_head / lm_propose, but a single
// lane issuing ~1500 f64 instructions costs ~4 us per step on a 64-wide SIMD.  Here lane r < 6
// owns row r of the 6x6 system; values another row needs travel by v_readlane (uniform
// broadcasts), so a step is ~400 instructions.  Differences from the serial code are at rounding
// level only: the 6x6 system is solved by Gauss-Jordan elimination on the lanes' rows with reciprocals from
// v_rcp_f64 + two Newton steps (lm_core.hpp: Cholesky with the correctly rounded library sqrt and division).
// All 64 lanes run the code (uniform control flow); lanes >= 6 compute unused values and never store.
__device__ __forceinline__ double lane_bcast(double v, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void sqrt_and_inverse(double d, double &root, double &inv)
{
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    const double dd = __builtin_fma(-g, g, d);
    root = __builtin_fma(dd, h, g);
    inv = h + h;
}

__device__ __forceinline__ double fast_rcp(double d)  // v_rcp_f64 + two Newton steps
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}

// manifold_plus (pose_math.hpp) with the square root from v_rsq_f64: one dependent chain shorter
__device__ __forceinline__ void manifold_plus_fast(const double x[7], const double d[6], double out[7])
{
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 == 0.0) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = x[i];
    } else {
        double nd, unused;
        sqrt_and_inverse(n2, nd, unused);
        double s, c;
        lmw_sinc_cos(nd, s, c);
        const double z[4] = {c, s * d[0], s * d[1], s * d[2]};
        out[0] = z[0] * x[0] - z[1] * x[1] - z[2] * x[2] - z[3] * x[3];
        out[1] = z[0] * x[1] + z[1] * x[0] + z[2] * x[3] - z[3] * x[2];
        out[2] = z[0] * x[2] - z[1] * x[3] + z[2] * x[0] + z[3] * x[1];
        out[3] = z[0] * x[3] + z[1] * x[2] - z[2] * x[1] + z[3] * x[0];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + d[3 + i];
}

// lm_assemble for lane r: its row of A, its g, and the (uniform) cost
__device__ __forceinline__ void lmw_assemble(const double *s, const double *xx, const double *prior_b, int r,
                                             double Arow[6], double &g_r, double &cost)
{
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int a = r < j ? r : j, b = r < j ? j : r;
        Arow[j] = s[a * 6 - (a * (a - 1)) / 2 + (b - a)];  // upper-triangle index of (a,b)
    }
    g_r = s[21 + r];
    cost = s[27];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double res = kPriorW * (xx[4 + a] - prior_b[a]);
        if (r == 3 + a) {
            Arow[3 + a] += kPriorW * kPriorW;
            g_r += kPriorW * res;
        }
        cost += 0.5 * res * res;
    }
}

__device__ __forceinline__ double lmw_pick(const double v[6], int r)  // v[r], r varies by lane
{
    // Conditional moves, kept as such: the compiler recognises a select chain over one index as v[r]
    // and implements that as an array in scratch memory (stores + a dependent scratch load in the
    // middle of the policy's chain).  An opaque copy of the index per comparison prevents it.
    double o = v[0];
#pragma unroll
    for (int j = 1; j < 6; j++) {
        int rj = r;
        asm volatile("" : "+v"(rj));
        o = (rj == j) ? v[j] : o;
    }
    return o;
}

__device__ __forceinline__ double lmw_gmax(double g_r)
{
    double m = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) m = fmax(m, fabs(lane_bcast(g_r, c)));
    return m;
}

__device__ __forceinline__ void lmw_store_normal(LmState &S, int lane, int r, const double Arow[6], double g_r,
                                                 double cost)
{
    if (lane < 6) {
#pragma unroll
        for (int j = 0; j < 6; j++) S.N.A[r][j] = Arow[j];
        S.N.g[r] = g_r;
    }
    if (lane == 0) S.N.cost = cost;
}

// lm_begin_head
__device__ __forceinline__ int lmw_begin(LmState &S, const double *first, const double *x, const double *prior_b,
                                         int lane)
{
    const int r = lane < 6 ? lane : 5;
    double Arow[6], g_r, cost;
    double pb[3] = {prior_b[0], prior_b[1], prior_b[2]};
    lmw_assemble(first, x, pb, r, Arow, g_r, cost);
    const double scale_r = 1.0 / (1.0 + sqrt(lmw_pick(Arow, r)));
    double xs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) xs[i] = x[i];
    const double x_norm = lmw_norm7(xs);
    lmw_store_normal(S, lane, r, Arow, g_r, cost);
    if (lane < 6) {
        S.scale[r] = scale_r;
        S.diag[r] = 0.0;
    }
    if (lane < 7) {
        S.x[lane] = x[lane];
        S.cand[lane] = x[lane];
    }
    if (lane < 3) S.prior_b[lane] = prior_b[lane];
    if (lane == 0) {
        S.x_norm = x_norm;
        S.radius = 1e4;
        S.decrease_factor = 2.0;
        S.reuse_diag = 0;
        S.invalid_run = 0;
        S.iter = 1;
        S.recorded = 1;
        S.evaluations = 1;
        S.last_step_norm = 0.0;
        S.model_change = 0.0;
        S.cost = cost;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lmw_gmax(g_r) <= kLmGtol) return LM_DONE;
    return LM_PROPOSE;
}

// lm_feed_head
__device__ __forceinline__ int lmw_feed(LmState &S, const double *sums, int lane)
{
    const int r = lane < 6 ? lane : 5;
    double xs[7], cs[7], pb[3];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        xs[i] = S.x[i];
        cs[i] = S.cand[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) pb[i] = S.prior_b[i];
    const double n_cost = S.N.cost, x_norm = S.x_norm, model_change = S.model_change;
    double radius = S.radius, decrease_factor = S.decrease_factor;
    const int evaluations = S.evaluations + 1, recorded = S.recorded, iter = S.iter;
    double Crow[6], cg_r, c_cost;
    lmw_assemble(sums, cs, pb, r, Crow, cg_r, c_cost);
    double d7[7];
#pragma unroll
    for (int i = 0; i < 7; i++) d7[i] = xs[i] - cs[i];
    const double sn = lmw_norm7(d7);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has read the state: lane 0 may now update it
    if (lane == 0) {
        S.evaluations = evaluations;
        S.cost = n_cost;
    }
    if (sn <= kLmPtol * (x_norm + kLmPtol)) return LM_DONE;           // parameter tolerance: not recorded
    const double cost_change = n_cost - c_cost;
    if (fabs(cost_change) <= kLmFtol * n_cost) return LM_DONE;        // function tolerance: not recorded
    const double rel_dec = cost_change / model_change;
    const bool accept = rel_dec > kLmMinRelDec;
    double g_now = S.N.g[r];
    double cost_now = n_cost;
    __builtin_amdgcn_wave_barrier();
    if (accept) {
        lmw_store_normal(S, lane, r, Crow, cg_r, c_cost);
        if (lane < 7) S.x[lane] = S.cand[lane];
        const double d3 = 2.0 * rel_dec - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - d3 * d3 * d3);
        radius = fmin(kLmMaxRadius, radius);
        decrease_factor = 2.0;
        g_now = cg_r;
        cost_now = c_cost;
    } else {
        radius /= decrease_factor;
        decrease_factor *= 2.0;
    }
    if (lane == 0) {
        if (accept) S.x_norm = lmw_norm7(cs);
        S.radius = radius;
        S.decrease_factor = decrease_factor;
        S.reuse_diag = accept ? 0 : 1;
        S.recorded = recorded + 1;
        S.last_step_norm = sn;
        S.cost = cost_now;
    }
    const bool done = lmw_gmax(g_now) <= kLmGtol;
    if (!done && lane == 0) S.iter = iter + 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return done ? LM_DONE : LM_PROPOSE;
}

// lm_propose
__device__ __forceinline__ int lmw_propose(LmState &S, int lane)
{
    const int r = lane < 6 ? lane : 5;
    double Arow[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Arow[j] = S.N.A[r][j];
    const double g_r = S.N.g[r], scale_r = S.scale[r];
    double diag_r = S.diag[r];
    double radius = S.radius, decrease_factor = S.decrease_factor;
    int reuse_diag = S.reuse_diag, invalid_run = S.invalid_run, iter = S.iter, recorded = S.recorded;
    double last_step_norm = S.last_step_norm;
    double xs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) xs[i] = S.x[i];
    double scale_c[6];
#pragma unroll
    for (int j = 0; j < 6; j++) scale_c[j] = lane_bcast(scale_r, j);
    double As[6];
#pragma unroll
    for (int j = 0; j < 6; j++) As[j] = Arow[j] * scale_r * scale_c[j];
    const double gs_r = g_r * scale_r;
    int result = LM_DONE;
    double cand[7] = {0, 0, 0, 0, 0, 0, 0}, model_change = 0.0;
    while (iter <= kLmMaxIter) {
        if (!reuse_diag) diag_r = fmin(fmax(lmw_pick(As, r), kLmMinDiag), kLmMaxDiag);
        const double inv_radius = fast_rcp(radius);
        double M[6];
#pragma unroll
        for (int j = 0; j < 6; j++) M[j] = (r == j) ? As[j] + diag_r * inv_radius : As[j];
        // Gauss-Jordan on the augmented rows [M_r | gs_r], lane r owns row r: per pivot one reciprocal, the pivot
        // row travels by v_readlane, and every lane updates its whole row with INDEPENDENT multiply-adds -- no
        // triangular substitutions afterwards (their 12 dependent broadcast-multiply-add steps were as long as
        // the factorisation).  Same elimination order as the L D L^T it replaces: the pivots are its D, so
        // "all pivots positive" is still "the Cholesky factor of lm_core.hpp exists".
        double rhs = gs_r;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double piv[6];
#pragma unroll
            for (int j = k; j < 6; j++) piv[j] = lane_bcast(M[j], k);
            const double prhs = lane_bcast(rhs, k);
            if (!(piv[k] > 0.0)) ok = false;
            const double f = (r == k) ? 0.0 : M[k] * fast_rcp(piv[k]);
#pragma unroll
            for (int j = k; j < 6; j++) M[j] -= f * piv[j];
            rhs -= f * prhs;
        }
        double y[6];
        if (ok) {
            const double y_r = rhs * fast_rcp(lmw_pick(M, r));
#pragma unroll
            for (int m = 0; m < 6; m++) y[m] = lane_bcast(y_r, m);
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (!lm_finite(y[i])) ok = false;
        }
        reuse_diag = 1;
        double step[6];
        model_change = 0.0;
        if (ok) {
            double gsdot = 0.0, quad = 0.0, row = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) step[c] = -y[c];
#pragma unroll
            for (int b = 0; b < 6; b++) row += As[b] * step[b];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                gsdot += lane_bcast(gs_r, a) * step[a];
                quad += step[a] * lane_bcast(row, a);
            }
            model_change = -gsdot - 0.5 * quad;
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid_run >= 5) break;
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            recorded++;
            last_step_norm = 0.0;
            iter++;
            continue;
        }
        invalid_run = 0;
        double delta[6];
#pragma unroll
        for (int c = 0; c < 6; c++) delta[c] = step[c] * scale_c[c];
        manifold_plus_fast(xs, delta, cand);
        result = LM_EVAL;
        break;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every lane has read the state
    if (lane < 6) S.diag[r] = diag_r;
    if (lane == 0) {
        S.radius = radius;
        S.decrease_factor = decrease_factor;
        S.reuse_diag = reuse_diag;
        S.invalid_run = invalid_run;
        S.iter = iter;
        S.recorded = recorded;
        S.last_step_norm = last_step_norm;
        if (result == LM_EVAL)
            S.model_change = model_change;
        else
            S.cost = S.N.cost;
    }
    if (result == LM_EVAL && lane < 7) {  // one store: lane i writes cand[i]
        double v = cand[0];
#pragma unroll
        for (int i = 1; i < 7; i++) v = (lane == i) ? cand[i] : v;
        S.cand[lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return result;
}

// =====================================================================================================
// Second form of the same policy (the one k_lm runs): the state of a solve stays in the REGISTERS of the wave
// that runs it, across all evaluations of the solve, and the step is written for its instruction count:
//   * every 16-lane row of the wave is a copy of row 0 -- lane l works on row r = min(l % 16, 5) of the system --
//     so whatever is the same for the six rows is wave-uniform without any broadcast;
//   * a value of row k reaches the others by DPP row_newbcast (two v_mov_b32_dpp, no SGPR round trip);
//   * lane-dependent picks (the own diagonal element, the own pivot's reciprocal) are carried along instead of
//     selected afterwards; sums and maxima over the six rows are DPP row reductions;
//   * the sums come from LDS through per-lane indices computed once per solve.
// tools/microbench/policy.hip replays real solves through both forms and checks them against lm_core.hpp.
template <int k>
__device__ __forceinline__ double row_bcast(double v)  // lane k of the row, to every lane of the row
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + k, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + k, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int kCtrl>
__device__ __forceinline__ double row_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// butterflies over the 16 lanes of a row: xor 1, xor 2, the two quads of a half, the two halves
__device__ __forceinline__ double row_max16(double v)
{
    v = fmax(v, row_dpp_f64<0xB1>(v));
    v = fmax(v, row_dpp_f64<0x4E>(v));
    v = fmax(v, row_dpp_f64<0x141>(v));
    return fmax(v, row_dpp_f64<0x140>(v));
}
__device__ __forceinline__ double row_sum16(double v)
{
    v += row_dpp_f64<0xB1>(v);
    v += row_dpp_f64<0x4E>(v);
    v += row_dpp_f64<0x141>(v);
    return v + row_dpp_f64<0x140>(v);
}

struct LmWave {  // registers of the wave that runs the solve, live across its evaluations
    // per lane: row r = min(lane % 16, 5) of the normal equations at x (prior included)
    double A[6], Ad, g;      // the row, its diagonal element, its gradient entry
    double scale, diag;      // Jacobi scale and LM diagonal of row r
    int r;
    bool first6;             // lane % 16 < 6: this lane's row counts in sums over the rows (lanes 6..15 repeat row 5)
    // the same in every lane
    double n_cost, radius;
    int reuse_diag, iter;
};
// What is the same for all rows and not needed in every instruction.  kReg = false: ONE copy in LDS, written by lane 0
// (the 512-thread kernels have no registers to spare); kReg = true: a copy in every lane's registers -- every lane
// computes these values anyway, so every lane keeps them and the LDS round trips of the step go (3,850 -> 3,224 cycles
// per step with a solve in tools/microbench/policy.hip).
struct LmShared {
    double x[7];   // the current iterate
    double x_norm, dec, model_change;
    double cost, last_step_norm;
    int recorded, evaluations, invalid_run;
};

__device__ __forceinline__ void lmw2_sync()  // LDS written by one lane, read by the others of this wave
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lm_assemble for lane r: its row, diagonal, g and the (uniform) cost of the sums `s` taken at point xx
__device__ __forceinline__ void lmw2_assemble(const LmWave &W, const double *s, const double *xx, const double *prior_b,
                                              double Arow[6], double &Ad, double &g_r, double &cost)
{
    const int r = W.r;
#pragma unroll
    for (int j = 0; j < 6; j++) {  // where the sums of row r sit in a 32-double block: upper-triangle index of (a, b)
        const int a = r < j ? r : j, b = r < j ? j : r;
        Arow[j] = s[a * 6 - (a * (a - 1)) / 2 + (b - a)];
    }
    Ad = s[r * 6 - (r * (r - 1)) / 2] + (r >= 3 ? kPriorW * kPriorW : 0.0);
    g_r = s[21 + W.r];
    cost = s[27];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double res = kPriorW * (xx[4 + a] - prior_b[a]);
        const double mine = (W.r == 3 + a) ? 1.0 : 0.0;
        Arow[3 + a] += mine * (kPriorW * kPriorW);
        g_r += mine * (kPriorW * res);
        cost += 0.5 * res * res;
    }
}

__device__ __forceinline__ bool lmw2_gradient_converged(double g_r)  // lm_gmax(N) <= gradient tolerance
{
    return __builtin_amdgcn_readfirstlane((int)(row_max16(fabs(g_r)) <= kLmGtol)) != 0;
}

// sin(a) / a and cos(a) of a step's half-angle: LM steps are small, six series terms cover a < 0.05 to the last bit
__device__ __forceinline__ void lmw2_sinc_cos(double a, double &sinc, double &c)
{
    if (a < 0.05) {
        const double z = a * a;
        double s = 1.0 / 39916800.0;  // 1/11!
        s = 1.0 / 362880.0 - z * s;   // 1/9!
        s = 1.0 / 5040.0 - z * s;     // 1/7!
        s = 1.0 / 120.0 - z * s;      // 1/5!
        s = 1.0 / 6.0 - z * s;        // 1/3!
        sinc = 1.0 - z * s;
        double k = 1.0 / 3628800.0;   // 1/10!
        k = 1.0 / 40320.0 - z * k;    // 1/8!
        k = 1.0 / 720.0 - z * k;      // 1/6!
        k = 1.0 / 24.0 - z * k;       // 1/4!
        k = 1.0 / 2.0 - z * k;        // 1/2!
        c = 1.0 - z * k;
    } else {
        lmw_sinc_cos(a, sinc, c);
    }
}

// one pivot of the Gauss-Jordan elimination of lmw2_propose
template <int k>
__device__ __forceinline__ void lmw2_pivot(double M[6], double &rhs, double &my_rinv, bool &ok, int r)
{
    double piv[6];
#pragma unroll
    for (int j = k; j < 6; j++) piv[j] = row_bcast<k>(M[j]);
    const double prhs = row_bcast<k>(rhs);
    ok = ok && (piv[k] > 0.0);
    const double rinv = fast_rcp(piv[k]);
    const bool mine = r == k;
    my_rinv = mine ? rinv : my_rinv;
    const double f = mine ? 0.0 : M[k] * rinv;
#pragma unroll
    for (int j = k + 1; j < 6; j++) M[j] -= f * piv[j];
    rhs -= f * prhs;
}

// lm_propose: from the current iterate, solve for steps until one is worth evaluating (LM_EVAL: the point in
// cand[0..6] (LDS, written by lane 0), W.model_change set) or the iteration / invalid-step budgets are used up (LM_DONE)
template <bool kReg>
__device__ __forceinline__ int lmw2_propose(LmWave &W, LmShared &S, double *cand, int lane)
{
    const int r = W.r;
    double sc[6];
    sc[0] = row_bcast<0>(W.scale);
    sc[1] = row_bcast<1>(W.scale);
    sc[2] = row_bcast<2>(W.scale);
    sc[3] = row_bcast<3>(W.scale);
    sc[4] = row_bcast<4>(W.scale);
    sc[5] = row_bcast<5>(W.scale);
    double As[6];
#pragma unroll
    for (int j = 0; j < 6; j++) As[j] = W.A[j] * W.scale * sc[j];
    const double Asd = W.Ad * W.scale * W.scale;
    const double gs_r = W.g * W.scale;
    int recorded_add = 0, invalid_run = S.invalid_run;
    double last_step_norm = 0.0, dec = S.dec, model_change_out = 0.0;
    bool touched = false;
    int result = LM_DONE;
    while (W.iter <= kLmMaxIter) {
        if (!W.reuse_diag) W.diag = fmin(fmax(Asd, kLmMinDiag), kLmMaxDiag);
        const double dterm = W.diag * fast_rcp(W.radius);
        double M[6];
#pragma unroll
        for (int j = 0; j < 6; j++) M[j] = (r == j) ? As[j] + dterm : As[j];
        // Gauss-Jordan on the augmented rows [M_r | gs_r], lane r owns row r: per pivot one reciprocal (of the
        // broadcast pivot: every lane computes it), the rest of the pivot row travels by DPP, every lane updates its
        // row with independent FMAs.  The pivots are the D of the L D L^T that lm_core.hpp's Cholesky implies, so
        // "all pivots positive" is "its factor exists".  Lane r keeps the reciprocal of its own pivot.
        double rhs = gs_r, my_rinv = 0.0;
        bool ok = true;
        lmw2_pivot<0>(M, rhs, my_rinv, ok, r);
        lmw2_pivot<1>(M, rhs, my_rinv, ok, r);
        lmw2_pivot<2>(M, rhs, my_rinv, ok, r);
        lmw2_pivot<3>(M, rhs, my_rinv, ok, r);
        lmw2_pivot<4>(M, rhs, my_rinv, ok, r);
        lmw2_pivot<5>(M, rhs, my_rinv, ok, r);
        const double y_r = rhs * my_rinv;
        double step[6];
        step[0] = -row_bcast<0>(y_r);
        step[1] = -row_bcast<1>(y_r);
        step[2] = -row_bcast<2>(y_r);
        step[3] = -row_bcast<3>(y_r);
        step[4] = -row_bcast<4>(y_r);
        step[5] = -row_bcast<5>(y_r);
#pragma unroll
        for (int j = 0; j < 6; j++) ok = ok && lm_finite(step[j]);
        W.reuse_diag = 1;
        // -(J s).(r + J s / 2) = -g.s - s^T A s / 2 (scaled space) = sum over the rows of s_r (-g_r - (A s)_r / 2)
        double rowv = As[0] * step[0];
#pragma unroll
        for (int j = 1; j < 6; j++) rowv += As[j] * step[j];
        const double t_r = -y_r * (-gs_r - 0.5 * rowv);
        const double model_change = row_sum16(W.first6 ? t_r : 0.0);
        const bool good = __builtin_amdgcn_readfirstlane((int)(ok && model_change > 0.0)) != 0;
        if (!good) {
            touched = true;
            if (++invalid_run >= 5) break;
            W.radius /= dec;
            dec *= 2.0;
            recorded_add++;
            last_step_norm = 0.0;
            W.iter++;
            continue;
        }
        if (invalid_run != 0) touched = true;
        invalid_run = 0;
        double delta[6];
#pragma unroll
        for (int c = 0; c < 6; c++) delta[c] = step[c] * sc[c];
        // manifold_plus (pose_math.hpp): Ceres QuaternionManifold::Plus, then the translation
        double x[7], out[7];
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = S.x[i];
        const double n2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
        if (n2 == 0.0) {
#pragma unroll
            for (int i = 0; i < 4; i++) out[i] = x[i];
        } else {
            double nd, unused, sn, cs;
            sqrt_and_inverse(n2, nd, unused);
            lmw2_sinc_cos(nd, sn, cs);
            const double z0 = cs, z1 = sn * delta[0], z2 = sn * delta[1], z3 = sn * delta[2];
            out[0] = z0 * x[0] - z1 * x[1] - z2 * x[2] - z3 * x[3];
            out[1] = z0 * x[1] + z1 * x[0] + z2 * x[3] - z3 * x[2];
            out[2] = z0 * x[2] - z1 * x[3] + z2 * x[0] + z3 * x[1];
            out[3] = z0 * x[3] + z1 * x[2] - z2 * x[1] + z3 * x[0];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + delta[3 + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 7; i++) cand[i] = out[i];
        }
        model_change_out = model_change;
        result = LM_EVAL;
        break;
    }
    if (kReg || lane == 0) {
        if (touched) {  // steps without a decrease were recorded on the way (rare)
            S.invalid_run = invalid_run;
            S.recorded += recorded_add;
            S.dec = dec;
            if (recorded_add) S.last_step_norm = last_step_norm;
        }
        if (result == LM_DONE)
            S.cost = W.n_cost;
        else
            S.model_change = model_change_out;
    }
    lmw2_sync();
    return result;
}

// lm_begin_head: `first` = the sums of the evaluation at x (iteration 0), in LDS; x = the point, in LDS.  Returns
// LM_DONE or LM_PROPOSE (= call lmw2_propose next; the caller does, so that the solve exists once in the kernel).
template <bool kReg>
__device__ __forceinline__ int lmw2_begin(LmWave &W, LmShared &S, const double *first, const double *x,
                                          const double *prior_b, int lane)
{
    const int l16 = lane & 15;
    const int r = l16 < 6 ? l16 : 5;
    W.r = r;
    W.first6 = l16 < 6;
    double xs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) xs[i] = x[i];
    double cost;
    lmw2_assemble(W, first, xs, prior_b, W.A, W.Ad, W.g, cost);
    // Jacobi scaling, computed once at iteration 0: 1 / (1 + ||column||)
    {
        double root, unused;
        sqrt_and_inverse(W.Ad, root, unused);
        W.scale = fast_rcp(1.0 + (W.Ad > 0.0 ? root : 0.0));
    }
    W.diag = 0.0;
    const double x_norm = lmw_norm7(xs);
    W.radius = 1e4;
    W.reuse_diag = 0;
    W.iter = 1;
    W.n_cost = cost;
    if constexpr (kReg) {
#pragma unroll
        for (int i = 0; i < 7; i++) S.x[i] = xs[i];
    } else if (lane < 7) {
        S.x[lane] = x[lane];
    }
    if (kReg || lane == 0) {
        S.x_norm = x_norm;
        S.dec = 2.0;
        S.model_change = 0.0;
        S.invalid_run = 0;
        S.recorded = 1;
        S.evaluations = 1;
        S.last_step_norm = 0.0;
        S.cost = cost;
    }
    lmw2_sync();
    if (lmw2_gradient_converged(W.g)) return LM_DONE;
    return LM_PROPOSE;
}

// lm_feed_head: `sums` = the evaluation at the candidate `cand` (both in LDS).  One evaluation at the candidate
// serves the accept test (cost) and, if accepted, the next iteration (Jacobian).  Returns LM_DONE or LM_PROPOSE.
template <bool kReg>
__device__ __forceinline__ int lmw2_feed(LmWave &W, LmShared &S, const double *sums, const double *cand,
                                         const double *prior_b, int lane)
{
    double cs[7], xs[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        cs[i] = cand[i];
        xs[i] = S.x[i];
    }
    double C[6], Cd, cg, c_cost;
    lmw2_assemble(W, sums, cs, prior_b, C, Cd, cg, c_cost);
    double d7[7];
#pragma unroll
    for (int i = 0; i < 7; i++) d7[i] = xs[i] - cs[i];
    const double sn = lmw_norm7(d7);
    const double cost_change = W.n_cost - c_cost;
    const double x_norm = S.x_norm, model_change = S.model_change, dec = S.dec;
    const bool stop = __builtin_amdgcn_readfirstlane((int)(sn <= kLmPtol * (x_norm + kLmPtol) ||             // parameter tolerance
                                                           fabs(cost_change) <= kLmFtol * W.n_cost)) != 0;  // function tolerance
    if (stop) {  // neither is recorded
        if (kReg || lane == 0) {
            S.evaluations++;
            S.cost = W.n_cost;
        }
        lmw2_sync();
        return LM_DONE;
    }
    const double rel_dec = cost_change / model_change;
    const bool accept = __builtin_amdgcn_readfirstlane((int)(rel_dec > kLmMinRelDec)) != 0;
    if (accept) {
#pragma unroll
        for (int j = 0; j < 6; j++) W.A[j] = C[j];
        W.Ad = Cd;
        W.g = cg;
        W.n_cost = c_cost;
        const double cand_norm = lmw_norm7(cs);
        const double d3 = 2.0 * rel_dec - 1.0;
        W.radius = fmin(kLmMaxRadius, W.radius / fmax(1.0 / 3.0, 1.0 - d3 * d3 * d3));
        W.reuse_diag = 0;
        if constexpr (kReg) {
#pragma unroll
            for (int i = 0; i < 7; i++) S.x[i] = cs[i];
        } else if (lane < 7) {
            S.x[lane] = cand[lane];
        }
        if (kReg || lane == 0) {
            S.x_norm = cand_norm;
            S.dec = 2.0;
        }
    } else {
        W.radius /= dec;
        W.reuse_diag = 1;
        if (kReg || lane == 0) S.dec = dec * 2.0;
    }
    if (kReg || lane == 0) {
        S.evaluations++;
        S.recorded++;
        S.last_step_norm = sn;
        S.cost = W.n_cost;
    }
    lmw2_sync();
    if (lmw2_gradient_converged(W.g)) return LM_DONE;
    W.iter++;
    return LM_PROPOSE;
}

#pragma clang fp contract(off)

}  // namespace lom
