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
// 1e-4 rad bar (tests/test_host_exchange.py, tests/test_gpu_parity.py).  The policy itself lives in
// lm_core.hpp, shared with the device-resident loop.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "../../include/lidar_odometry_amd.h"
#include "lm_core.hpp"
#include "pose_math.hpp"

namespace {

using lom::LmState;

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

// one ceres::Solve (cloud_matcher.cpp:157-158): lm_core.hpp's state machine, fed by the hooks.
// `first` holds the sums of iteration 0.
int lm_solve(Evaluator &ev, const double first[LOM_NSUMS], double x[7], const double prior_b[3], LmState &S)
{
    int action = lom::lm_begin(S, first, x, prior_b);
    while (action == lom::LM_EVAL) {
        double sums[LOM_NSUMS];
        const int rc = ev.evaluate(false, nullptr, nullptr, S.cand, sums);
        if (rc != LOM_OK) return rc;
        action = lom::lm_feed(S, sums);
    }
    for (int i = 0; i < 7; i++) x[i] = S.x[i];
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
        LmState lr;
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
