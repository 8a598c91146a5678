// Host side of CloudMatcher::align (reference src/cloud_matcher.cpp:105-178):
// the <=35-iteration outer loop and the Levenberg-Marquardt policy that the
// reference gets from ceres::Solve (DENSE_QR, max_num_iterations 4,
// function_tolerance 1e-5), restated on the reduced 6x6 / 6x1 normal equations
// that the HIP kernels (or any other evaluator behind lom_align_hooks) return.
// No Ceres, no Eigen: 6x6 Cholesky in f64.
//
// What differs from the reference by construction: the reference solves the
// stacked least-squares system by QR over all residual rows, here the same
// step comes from (S A S + D^2) y = S g on the reduced system.  Both are the
// minimiser of the same damped quadratic; poses agree far inside the 1e-4 m /
// 1e-4 rad bar (tests/test_align_driver.py, tests/test_gpu_parity.py).
#include <cfloat>
#include <cmath>
#include <cstring>

#include "../../include/lidar_odometry_amd.h"
#include "pose_math.hpp"

namespace {

constexpr double kPriorW = 10.0;  // cloud_matcher.cpp:153  diag(0.1,0.1,0.1).inverse()

struct Normal {
    double A[6][6];
    double g[6];
    double cost;
};

// sums (device layout, prior excluded) -> full normal equations incl. the
// NormalPrior on translation (residual 10 (t - t_guess), Jacobian 10 I).
void assemble(const double s[LOM_NSUMS], const double x[7], const double prior_b[3], Normal &n)
{
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
            n.A[a][b] = s[k];
            n.A[b][a] = s[k];
            k++;
        }
    for (int a = 0; a < 6; a++) n.g[a] = s[21 + a];
    n.cost = s[27];
    for (int a = 0; a < 3; a++) {
        const double r = kPriorW * (x[4 + a] - prior_b[a]);
        n.A[3 + a][3 + a] += kPriorW * kPriorW;
        n.g[3 + a] += kPriorW * r;
        n.cost += 0.5 * r * r;
    }
}

bool cholesky_solve6(const double M[6][6], const double b[6], double y[6])
{
    double L[6][6] = {};
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j <= i; j++) {
            double s = M[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i][i] = std::sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    }
    double z[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i][k] * z[k];
        z[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = z[i];
        for (int k = i + 1; k < 6; k++) s -= L[k][i] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 0; i < 6; i++)
        if (!std::isfinite(y[i])) return false;
    return true;
}

struct LmResult {
    int recorded = 1;  // iteration 0
    int evaluations = 0;
    double last_step_norm = 0.0;
    double cost = 0.0;
};

struct Evaluator {
    const lom_align_hooks *h;
    int evaluate(bool fresh, const float pose_t[3], const float pose_q[4], const double x[7],
                 double sums[LOM_NSUMS])
    {
        int rc = fresh ? h->match_eval(h->user, pose_t, pose_q, x, x + 4, sums)
                       : h->eval_fixed(h->user, x, x + 4, sums);
        if (rc != 0) return LOM_ERR_HOOK;
        if (h->allreduce) {
            rc = h->allreduce(h->user, sums, LOM_NSUMS);
            if (rc != 0) return LOM_ERR_HOOK;
        }
        return LOM_OK;
    }
};

// Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, library defaults
// except max_num_iterations = 4, function_tolerance = 1e-5
// (cloud_matcher.cpp:109-112).  `first` holds the sums of iteration 0.
int lm_solve(Evaluator &ev, const double first[LOM_NSUMS], double x[7], const double prior_b[3],
             LmResult &out)
{
    const int max_iter = 4;
    const double ftol = 1e-5, gtol = 1e-10, ptol = 1e-8;
    const double min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diag = false;
    double scale[6], diag[6];
    Normal N;
    assemble(first, x, prior_b, N);
    // Jacobi scaling, computed once at iteration 0: 1 / (1 + ||column||)
    for (int c = 0; c < 6; c++) scale[c] = 1.0 / (1.0 + std::sqrt(N.A[c][c]));
    auto gmax_of = [](const Normal &n) {
        double m = 0.0;
        for (int c = 0; c < 6; c++) m = std::fmax(m, std::fabs(n.g[c]));
        return m;
    };
    auto norm7 = [](const double *v) {
        double s = 0.0;
        for (int i = 0; i < 7; i++) s += v[i] * v[i];
        return std::sqrt(s);
    };
    double x_norm = norm7(x);
    out.cost = N.cost;
    if (gmax_of(N) <= gtol) return LOM_OK;
    int invalid_run = 0;
    for (int iter = 1; iter <= max_iter; iter++) {
        double As[6][6], gs[6];
        for (int a = 0; a < 6; a++) {
            gs[a] = N.g[a] * scale[a];
            for (int b = 0; b < 6; b++) As[a][b] = N.A[a][b] * scale[a] * scale[b];
        }
        if (!reuse_diag)
            for (int c = 0; c < 6; c++) diag[c] = std::fmin(std::fmax(As[c][c], min_diag), max_diag);
        double M[6][6];
        std::memcpy(M, As, sizeof M);
        for (int c = 0; c < 6; c++) M[c][c] += diag[c] / radius;
        double y[6], step[6];
        const bool ok = cholesky_solve6(M, gs, y);
        reuse_diag = true;
        double model_change = 0.0;
        if (ok) {
            for (int c = 0; c < 6; c++) step[c] = -y[c];
            // -(J s).(r + J s / 2) = -g.s - s^T A s / 2   (scaled space)
            double gsdot = 0.0, quad = 0.0;
            for (int a = 0; a < 6; a++) {
                gsdot += gs[a] * step[a];
                double row = 0.0;
                for (int b = 0; b < 6; b++) row += As[a][b] * step[b];
                quad += step[a] * row;
            }
            model_change = -gsdot - 0.5 * quad;
        }
        if (!ok || !(model_change > 0.0)) {
            if (++invalid_run >= 5) break;
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            out.recorded++;
            out.last_step_norm = 0.0;
            continue;
        }
        invalid_run = 0;
        double delta[6], cand[7];
        for (int c = 0; c < 6; c++) delta[c] = step[c] * scale[c];
        lom::manifold_plus(x, delta, cand);
        // one evaluation at the candidate serves the accept test (cost) and, if
        // accepted, the next iteration (Jacobian) -- the reference evaluates the
        // cost first and the Jacobian after acceptance; same numbers, one pass.
        double sums[LOM_NSUMS];
        int rc = ev.evaluate(false, nullptr, nullptr, cand, sums);
        if (rc != LOM_OK) return rc;
        out.evaluations++;
        Normal C;
        assemble(sums, cand, prior_b, C);
        double d7[7];
        for (int i = 0; i < 7; i++) d7[i] = x[i] - cand[i];
        const double sn = norm7(d7);
        if (sn <= ptol * (x_norm + ptol)) break;           // parameter tolerance: not recorded
        const double cost_change = N.cost - C.cost;
        if (std::fabs(cost_change) <= ftol * N.cost) break;  // function tolerance: not recorded
        const double rel_dec = cost_change / model_change;
        if (rel_dec > min_rel_dec) {
            std::memcpy(x, cand, sizeof cand);
            x_norm = norm7(x);
            N = C;
            const double d3 = 2.0 * rel_dec - 1.0;
            radius = radius / std::fmax(1.0 / 3.0, 1.0 - d3 * d3 * d3);
            radius = std::fmin(max_radius, radius);
            decrease_factor = 2.0;
            reuse_diag = false;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            reuse_diag = true;
        }
        out.recorded++;
        out.last_step_norm = sn;
        out.cost = N.cost;
        if (gmax_of(N) <= gtol) break;
    }
    out.cost = N.cost;
    return LOM_OK;
}

}  // namespace

extern "C" {

int lom_abi_version(void) { return LOM_ABI_VERSION; }

void lom_pose_identity(lom_pose *out)
{
    out->t[0] = out->t[1] = out->t[2] = 0.f;
    out->q[0] = 1.f;
    out->q[1] = out->q[2] = out->q[3] = 0.f;
}

void lom_pose_compose(const lom_pose *a, const lom_pose *b, lom_pose *out) { lom::pose_compose(*a, *b, *out); }

void lom_pose_inverse(const lom_pose *a, lom_pose *out) { lom::pose_inverse(*a, *out); }

void lom_pose_relative_to(const lom_pose *a, const lom_pose *target, lom_pose *out)
{
    lom_pose inv;
    lom::pose_inverse(*a, inv);
    lom::pose_compose(inv, *target, *out);
}

void lom_pose_rotation_matrix(const lom_pose *a, float R[9]) { lom::rotation_matrix(a->q, R); }

int lom_transform_points(const lom_pose *pose, const float *xyz_in, const float *nrm_in, size_t n,
                         size_t stride_in, float *xyz_out, float *nrm_out, size_t stride_out)
{
    if (!pose || (n && (!xyz_in || !xyz_out)) || stride_in < 12 || stride_out < 12) return LOM_ERR_ARG;
    float R[9];
    lom::rotation_matrix(pose->q, R);
    for (size_t i = 0; i < n; i++) {
        const float *p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz_in) + i * stride_in);
        float *o = reinterpret_cast<float *>(reinterpret_cast<char *>(xyz_out) + i * stride_out);
        const float p0 = p[0], p1 = p[1], p2 = p[2];
        o[0] = lom::sum3(R[0] * p0, R[1] * p1, R[2] * p2) + pose->t[0];
        o[1] = lom::sum3(R[3] * p0, R[4] * p1, R[5] * p2) + pose->t[1];
        o[2] = lom::sum3(R[6] * p0, R[7] * p1, R[8] * p2) + pose->t[2];
        if (nrm_in && nrm_out) {
            const float *q = reinterpret_cast<const float *>(reinterpret_cast<const char *>(nrm_in) + i * stride_in);
            float *no = reinterpret_cast<float *>(reinterpret_cast<char *>(nrm_out) + i * stride_out);
            const float n0 = q[0], n1 = q[1], n2 = q[2];
            no[0] = lom::sum3(R[0] * n0, R[1] * n1, R[2] * n2);
            no[1] = lom::sum3(R[3] * n0, R[4] * n1, R[5] * n2);
            no[2] = lom::sum3(R[6] * n0, R[7] * n1, R[8] * n2);
        }
    }
    return LOM_OK;
}

int lom_align_with_hooks(const lom_align_hooks *hooks, const float guess_t[3], const float guess_q[4],
                         float out_t[3], float out_q[4], lom_align_stats *stats)
{
    if (!hooks || !hooks->match_eval || !hooks->eval_fixed || !guess_t || !guess_q || !out_t || !out_q)
        return LOM_ERR_ARG;
    lom_align_stats st;
    std::memset(&st, 0, sizeof st);
    Evaluator ev{hooks};
    float pt[3] = {guess_t[0], guess_t[1], guess_t[2]};              // cloud_matcher.cpp:107
    float pq[4] = {guess_q[0], guess_q[1], guess_q[2], guess_q[3]};
    const double prior_b[3] = {(double)guess_t[0], (double)guess_t[1], (double)guess_t[2]};  // :153
    for (int i = 0; i < 35; i++) {                                   // :117
        double x[7] = {(double)pq[0], (double)pq[1], (double)pq[2], (double)pq[3],  // :122-126
                       (double)pt[0], (double)pt[1], (double)pt[2]};                // :129-131
        double sums[LOM_NSUMS];
        int rc = ev.evaluate(true, pt, pq, x, sums);                 // :138-139 + iteration 0
        if (rc != LOM_OK) return rc;
        LmResult lr;
        lr.evaluations = 1;
        rc = lm_solve(ev, sums, x, prior_b, lr);                     // :157-158
        if (rc != LOM_OK) return rc;
        st.outer_iterations = i + 1;
        st.match_launches = i + 1;
        st.lm_iterations += lr.recorded;
        st.evaluations += lr.evaluations;
        st.valid_last = (int64_t)sums[28];
        st.cand_total += (int64_t)sums[29];
        st.occ_total += (int64_t)sums[30];
        st.queries += (int64_t)sums[31];
        // B(q) = 12 + 27*16 + 12*cand(q) + 12*valid(q)
        st.algorithmic_bytes += 444.0 * sums[31] + 12.0 * sums[29] + 12.0 * sums[28];
        st.final_cost = lr.cost;
        st.last_step_norm = lr.last_step_norm;
        for (int a = 0; a < 4; a++) pq[a] = (float)x[a];             // :161-164
        for (int a = 0; a < 3; a++) pt[a] = (float)x[4 + a];         // :165-167
        if (lr.last_step_norm < 1e-4 && i > 3) break;                // :169-172
    }
    {   // :175 rotation.normalize(), f32
        const float n2 = (pq[0] * pq[0] + pq[1] * pq[1]) + (pq[2] * pq[2] + pq[3] * pq[3]);
        const float nn = std::sqrt(n2);
        for (int a = 0; a < 4; a++) pq[a] = pq[a] / nn;
    }
    std::memcpy(out_t, pt, sizeof pt);
    std::memcpy(out_q, pq, sizeof pq);
    if (stats) *stats = st;
    return LOM_OK;
}

}  // extern "C"
