// Scan-matching kernels: the MI355X counterpart of
//   VoxelGrid::getCorrespondence / findMatchingPairs  (src/voxel_grid.h:164-234)
//   PointToPlaneErrorAnalytic::Evaluate               (src/cloud_matcher.cpp:38-103)
// plus the reduction of the robustified normal equations that Ceres performs
// inside ceres::Solve (DENSE_QR) for the reference.
//
//   k_match   per source point: f64 transform -> f32 query -> 27-neighbour
//             voxel lookup -> nearest stored point (strict-min, scan order
//             ix,iy,iz then insertion order) -> winner's point+normal.
//             HBM/L2-bound gather; no MFMA (nothing here is a contraction).
//   k_eval    per valid correspondence: r = (q*p + t - o).n, 1x6 tangent
//             Jacobian, Huber(0.15) IRLS weight; f64 wave reduction of
//             sum w J J^T (21), sum w J r (6), sum 0.5 rho (1) -> one partial
//             per workgroup.
//             The last workgroup to arrive sums the partials in fixed order and
//             publishes LOM_NSUMS doubles straight into pinned host memory.
//
// Built with -ffp-contract=off (see voxel_map.hip).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "lom_internal.hpp"
#include "pose_math.hpp"

namespace lom {

constexpr int kMatchThreads = 256;             // 4 waves
constexpr int kMatchG = 16;                    // lanes per query: four queries per wave
constexpr int kGroupsPerBlock = kMatchThreads / kMatchG;
constexpr int kEvalThreads = 256;

// per-query device record written by k_match
struct __attribute__((aligned(8))) QStat {
    float sq_dist;
    uint32_t n_cand;
    uint32_t n_occ;
    uint32_t pad;
};

// ---------------------------------------------------------------------------
// k_match<G>: one query per group of G lanes (G = 16: four queries per wave).
//
//  1. probe    lane l takes neighbours b = l, l+G, ... < 27 in the reference's scan
//              order ix, iy, iz (voxel_grid.h:175-179): one 16-byte slot load each.
//  2. prune    a neighbour voxel whose nearest possible coordinate is provably
//              farther than max_dist cannot hold a point with d2 < max_sq
//              (voxel_grid.h:186), so its points are not read.  Exact: such points
//              never win in the reference either.  Counts stay the reference's.
//  3. flatten  the remaining voxels' points form one candidate sequence in scan
//              order (inclusive prefix of the counts in LDS); lane l takes
//              candidates l, l+G, ... and finds each one's voxel by a 5-step
//              binary search -- every load of a query is in flight at once instead
//              of one dependent round trip per occupied voxel.
//  4. select   private strict minimum per lane (candidates arrive in scan order),
//              then the lexicographic minimum of (sq_dist, candidate ordinal) over
//              the group == "first encountered wins" of voxel_grid.h:183-191.
// ---------------------------------------------------------------------------
__device__ inline float axis_gap(float q, int i, float vs)
{
    // coordinates x with (int)(x / vs) == i lie in [lo, hi] (truncation: index 0 is
    // double width); returns a lower bound of |q - x| over that range
    const float lo = (i > 0) ? (float)i * vs : (float)(i - 1) * vs;
    const float hi = (i < 0) ? (float)i * vs : (float)(i + 1) * vs;
    const float g = fmaxf(fmaxf(lo - q, q - hi), 0.f);
    // slack for the f32 rounding of x / vs at the voxel faces and of the distance itself
    return fmaxf(g - (1e-4f * vs + 1e-6f * fabsf(q)), 0.f);
}

template <int G>
__global__ __launch_bounds__(kMatchThreads) void k_match(MapView map, const char *__restrict__ src, size_t stride,
                                                         uint32_t n, PoseArgs P, int32_t *__restrict__ out_idx,
                                                         float *__restrict__ out_on, QStat *__restrict__ out_stat,
                                                         uint32_t *__restrict__ block_counters)
{
    constexpr int kGroups = kMatchThreads / G;
    constexpr int kSets = (27 + G - 1) / G;
    __shared__ uint32_t s_pref[kGroups][32];  // inclusive prefix of scanned counts, scan order; padded with total
    __shared__ uint32_t s_base[kGroups][32];  // slab * K - exclusive prefix: point index = s_base[b] + c
    __shared__ uint32_t s_cnt[kGroups][3];
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const uint32_t groups_total = gridDim.x * kGroups;
    uint32_t acc_cand = 0, acc_occ = 0, acc_valid = 0;

    for (uint32_t q = blockIdx.x * kGroups + grp; q < n; q += groups_total) {
        const float *sp = reinterpret_cast<const float *>(src + (size_t)q * stride);
        const double p0 = (double)sp[0], p1 = (double)sp[1], p2 = (double)sp[2];
        // voxel_grid.h:220-223: R*p + t in f64 (Eigen order a0 + (a1 + a2)), cast to f32
        const float qx = (float)((P.R[0] * p0 + (P.R[1] * p1 + P.R[2] * p2)) + P.t[0]);
        const float qy = (float)((P.R[3] * p0 + (P.R[4] * p1 + P.R[5] * p2)) + P.t[1]);
        const float qz = (float)((P.R[6] * p0 + (P.R[7] * p1 + P.R[8] * p2)) + P.t[2]);
        int ix = 0, iy = 0, iz = 0;
        const bool inr = voxel_index(qx, map.voxel_size, ix) && voxel_index(qy, map.voxel_size, iy) &&
                         voxel_index(qz, map.voxel_size, iz);
        uint32_t cnt[kSets], scan_cnt[kSets], slab[kSets];
#pragma unroll
        for (int s = 0; s < kSets; s++) {
            cnt[s] = 0;
            scan_cnt[s] = 0;
            slab[s] = 0;
            const int b = gl + s * G;
            if (inr && b < 27) {
                const int nx = ix + b / 9 - 1, ny = iy + (b / 3) % 3 - 1, nz = iz + b % 3 - 1;
                // stored indices lie in (-2^20, 2^20); anything outside cannot exist
                if (nx > -kIdxBias && nx < kIdxBias && ny > -kIdxBias && ny < kIdxBias && nz > -kIdxBias &&
                    nz < kIdxBias) {
                    const unsigned long long key = pack_key(nx, ny, nz);
                    uint32_t h = hash_key(key, map.shift) & map.mask;
                    for (uint32_t probe = 0; probe <= map.mask; probe++) {
                        const uint4 raw = *reinterpret_cast<const uint4 *>(&map.table[h]);
                        const unsigned long long k = ((unsigned long long)raw.y << 32) | raw.x;
                        if (k == key) {
                            cnt[s] = raw.z;
                            slab[s] = raw.w;
                            break;
                        }
                        if (k == kEmptyKey) break;
                        h = (h + 1) & map.mask;
                    }
                    if (cnt[s]) {
                        const float gx = axis_gap(qx, nx, map.voxel_size);
                        const float gy = axis_gap(qy, ny, map.voxel_size);
                        const float gz = axis_gap(qz, nz, map.voxel_size);
                        const float lower = gx * gx + (gy * gy + gz * gz);
                        scan_cnt[s] = (lower > P.max_sq * 1.0001f) ? 0u : cnt[s];
                    }
                }
            }
        }
        // group-wide prefix over the neighbours in scan order (set 0 = b < G, set 1 = b >= G)
        uint32_t n_occ = 0, n_cand = 0, run = 0;
#pragma unroll
        for (int s = 0; s < kSets; s++) {
            uint32_t inc = scan_cnt[s], all = cnt[s], occ = cnt[s] ? 1u : 0u;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, G);
                if (gl >= d) inc += o;
                all += __shfl_xor(all, d, G);
                occ += __shfl_xor(occ, d, G);
            }
            const uint32_t set_total = __shfl(inc, G - 1, G);
            const int b = gl + s * G;
            if (b < 32) {
                s_pref[grp][b] = (b < 27) ? run + inc : 0xFFFFFFFFu;
                s_base[grp][b] = slab[s] * map.K - (run + inc - scan_cnt[s]);
            }
            run += set_total;
            n_cand += all;
            n_occ += occ;
        }
        if (kSets * G < 32) {  // G = 16 covers b < 32 with two sets; other G: pad the tail
            for (int b = kSets * G + gl; b < 32; b += G) s_pref[grp][b] = 0xFFFFFFFFu;
        }
        const uint32_t T = run;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float best = INFINITY;
        uint32_t best_c = 0xFFFFFFFFu, best_idx = 0;
        const uint32_t *pref = s_pref[grp];
        for (uint32_t c = gl; c < T; c += G) {
            // smallest b with pref[b] > c
            uint32_t b = 0;
            b += (pref[b + 15] <= c) ? 16u : 0u;
            b += (pref[b + 7] <= c) ? 8u : 0u;
            b += (pref[b + 3] <= c) ? 4u : 0u;
            b += (pref[b + 1] <= c) ? 2u : 0u;
            b += (pref[b] <= c) ? 1u : 0u;
            const uint32_t pi = s_base[grp][b] + c;
            const float *vp = map.pts + (size_t)pi * 3;
            const float ax = vp[0], ay = vp[1], az = vp[2];
            const float dx = qx - ax, dy = qy - ay, dz = qz - az;
            const float d2 = dx * dx + (dy * dy + dz * dz);  // voxel_grid.h:184 f32 squaredNorm
            if (d2 < P.max_sq && d2 < best) {                // :186-187 strict
                best = d2;
                best_c = c;
                best_idx = pi;
            }
        }
        // lexicographic min over the group; d2 >= 0 so its bit pattern orders like the value
        unsigned long long keyv = ((unsigned long long)__float_as_uint(best) << 32) | best_c;
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(keyv, d, G);
            keyv = o < keyv ? o : keyv;
        }
        const uint32_t w_c = (uint32_t)keyv;
        const bool valid = w_c != 0xFFFFFFFFu;
        const uint32_t w_idx = __shfl(best_idx, valid ? (int)(w_c % G) : 0, G);  // the lane that scanned it
        if (gl == 0) {
            int32_t idx = -1;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
            if (valid) {
                const size_t pi = w_idx;
                idx = (int32_t)pi;
                o0 = map.pts[pi * 3 + 0];  // voxel_grid.h:197-198
                o1 = map.pts[pi * 3 + 1];
                o2 = map.pts[pi * 3 + 2];
                n0 = map.nrm[pi * 3 + 0];
                n1 = map.nrm[pi * 3 + 1];
                n2 = map.nrm[pi * 3 + 2];
            }
            out_idx[q] = idx;
            float *on = out_on + (size_t)q * 6;
            on[0] = o0;
            on[1] = o1;
            on[2] = o2;
            on[3] = n0;
            on[4] = n1;
            on[5] = n2;
            if (out_stat) {
                QStat st;
                st.sq_dist = valid ? __uint_as_float((uint32_t)(keyv >> 32)) : 0.f;
                st.n_cand = n_cand;
                st.n_occ = n_occ;
                st.pad = 0;
                out_stat[q] = st;
            }
            acc_cand += n_cand;
            acc_occ += n_occ;
            acc_valid += valid ? 1 : 0;
        }
        // the LDS tables are rewritten next iteration: all reads above are complete for this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // per-block counters, summed in fixed order by k_finish (no same-address atomics:
    // one word saturates at ~88 atomics/us, MI355X_MICROARCH.md "dequeue")
    if (gl == 0) {
        s_cnt[grp][0] = acc_valid;
        s_cnt[grp][1] = acc_cand;
        s_cnt[grp][2] = acc_occ;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t v = 0;
        for (int g = 0; g < kGroups; g++) v += s_cnt[g][threadIdx.x];
        block_counters[blockIdx.x * 4 + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------
// k_eval: residual + Jacobian + robust weight + reduction + in-launch finish.
// One lane per source point, grid-stride; 28 f64 accumulators per lane; one
// partial per workgroup.  The workgroup that takes the last ticket sums the
// partials in workgroup order (bitwise run-to-run reproducible, independent of
// arrival order) and publishes LOM_NSUMS doubles: to `out` in HBM and, when
// `mail` is set, straight into coherent pinned host memory followed by a
// system-scope release store of `seq` (the host polls that word; no copy kernel,
// no stream synchronisation per residual evaluation).
//
// Inter-workgroup hand-off follows cdna_hip_programming.md Guideline 16:
// producer stores -> s_waitcnt vmcnt(0) -> barrier -> one lane: agent release,
// s_waitcnt vmcnt(0), returning ticket atomic; last arriver: agent acquire,
// s_waitcnt vmcnt(0), barrier, plain loads.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(kEvalThreads) void k_eval(const char *__restrict__ src, size_t stride, uint32_t n,
                                                       const int32_t *__restrict__ idx,
                                                       const float *__restrict__ on, EvalArgs E,
                                                       double *partials, uint32_t *ticket,
                                                       const uint32_t *__restrict__ block_counters,
                                                       uint32_t n_match_blocks, double *__restrict__ out,
                                                       double *mail, unsigned long long seq)
{
    __shared__ double s_red[kEvalThreads / 64][28];
    __shared__ double s_part[8][33];
    __shared__ unsigned long long s_cnt[kEvalThreads / 64][3];
    __shared__ uint32_t s_last;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.0;
    const double q0 = E.q[0], q1 = E.q[1], q2 = E.q[2], q3 = E.q[3];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // all three streams are issued before anything depends on them
        const int32_t ci = idx[i];
        const float *sp = reinterpret_cast<const float *>(src + (size_t)i * stride);
        const float s0 = sp[0], s1 = sp[1], s2 = sp[2];
        const float *c = on + (size_t)i * 6;
        const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
        if (ci < 0) continue;
        const double p[3] = {(double)s0, (double)s1, (double)s2};
        const double o[3] = {(double)c0, (double)c1, (double)c2};
        const double nn[3] = {(double)c3, (double)c4, (double)c5};
        // cloud_matcher.cpp:54  (rot*local_point + t - plane_origin).dot(plane_normal)
        double uv0 = q2 * p[2] - q3 * p[1];
        double uv1 = q3 * p[0] - q1 * p[2];
        double uv2 = q1 * p[1] - q2 * p[0];
        uv0 += uv0;
        uv1 += uv1;
        uv2 += uv2;
        const double rp0 = (p[0] + q0 * uv0) + (q2 * uv2 - q3 * uv1);
        const double rp1 = (p[1] + q0 * uv1) + (q3 * uv0 - q1 * uv2);
        const double rp2 = (p[2] + q0 * uv2) + (q1 * uv1 - q2 * uv0);
        const double e0 = rp0 + E.t[0] - o[0], e1 = rp1 + E.t[1] - o[1], e2 = rp2 + E.t[2] - o[2];
        const double r = e0 * nn[0] + (e1 * nn[1] + e2 * nn[2]);
        // cloud_matcher.cpp:64-91: ambient d r / d q_i = (dR/dq_i p).n
        double v0, v1, v2, ja[4];
        v0 = 2.0 * q0 * p[0] + 2.0 * -q3 * p[1] + 2.0 * q2 * p[2];
        v1 = 2.0 * q3 * p[0] + 2.0 * q0 * p[1] + 2.0 * -q1 * p[2];
        v2 = 2.0 * -q2 * p[0] + 2.0 * q1 * p[1] + 2.0 * q0 * p[2];
        ja[0] = v0 * nn[0] + (v1 * nn[1] + v2 * nn[2]);
        v0 = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        v1 = 2.0 * q2 * p[0] + 2.0 * -q1 * p[1] + 2.0 * -q0 * p[2];
        v2 = 2.0 * q3 * p[0] + 2.0 * q0 * p[1] + 2.0 * -q1 * p[2];
        ja[1] = v0 * nn[0] + (v1 * nn[1] + v2 * nn[2]);
        v0 = 2.0 * -q2 * p[0] + 2.0 * q1 * p[1] + 2.0 * q0 * p[2];
        v1 = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        v2 = 2.0 * -q0 * p[0] + 2.0 * q3 * p[1] + 2.0 * -q2 * p[2];
        ja[2] = v0 * nn[0] + (v1 * nn[1] + v2 * nn[2]);
        v0 = 2.0 * -q3 * p[0] + 2.0 * -q0 * p[1] + 2.0 * q1 * p[2];
        v1 = 2.0 * q0 * p[0] + 2.0 * -q3 * p[1] + 2.0 * q2 * p[2];
        v2 = 2.0 * q1 * p[0] + 2.0 * q2 * p[1] + 2.0 * q3 * p[2];
        ja[3] = v0 * nn[0] + (v1 * nn[1] + v2 * nn[2]);
        // Ceres QuaternionManifold plus-Jacobian (4x3): ambient -> tangent
        double J[6];
        J[0] = ja[0] * -q1 + ja[1] * q0 + ja[2] * -q3 + ja[3] * q2;
        J[1] = ja[0] * -q2 + ja[1] * q3 + ja[2] * q0 + ja[3] * -q1;
        J[2] = ja[0] * -q3 + ja[1] * -q2 + ja[2] * q1 + ja[3] * q0;
        J[3] = nn[0];  // cloud_matcher.cpp:96-98
        J[4] = nn[1];
        J[5] = nn[2];
        // ceres::HuberLoss(0.15) (cloud_matcher.cpp:134); rho'' <= 0 -> plain IRLS weight rho'
        const double s = r * r;
        double rho0 = s, w = 1.0;
        if (s > 0.15 * 0.15) {
            const double rr = sqrt(s);
            rho0 = 2.0 * 0.15 * rr - 0.15 * 0.15;
            w = fmax(DBL_MIN, 0.15 / rr);
        }
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double wa = w * J[a];
#pragma unroll
            for (int b = a; b < 6; b++) acc[k++] += wa * J[b];
        }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += w * J[a] * r;
        acc[27] += 0.5 * rho0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 28; k++) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kEvalThreads / 64; w++) v += s_red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 28 + threadIdx.x] = v;
    }
    // ---- hand-off: publish this workgroup's partial, take a ticket ----------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave's stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- last arriver: fixed-order final sum ---------------------------------------
    const uint32_t nb = gridDim.x;
    {
        const int k = threadIdx.x & 31, part = threadIdx.x >> 5;  // 8 strided parts per output
        double v = 0.0;
        if (k < 28) {
            uint32_t b = part;
            for (; b + 24 < nb; b += 32) {  // four independent loads in flight
                const double a0 = partials[(size_t)b * 28 + k], a1 = partials[(size_t)(b + 8) * 28 + k];
                const double a2 = partials[(size_t)(b + 16) * 28 + k], a3 = partials[(size_t)(b + 24) * 28 + k];
                v += a0;
                v += a1;
                v += a2;
                v += a3;
            }
            for (; b < nb; b += 8) v += partials[(size_t)b * 28 + k];
        }
        s_part[part][k] = v;
    }
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
    for (uint32_t b = threadIdx.x; b < n_match_blocks; b += kEvalThreads) {
        const uint4 r = *reinterpret_cast<const uint4 *>(block_counters + (size_t)b * 4);
        c0 += r.x;
        c1 += r.y;
        c2 += r.z;
    }
    if (n_match_blocks) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            c0 += __shfl_xor(c0, d, 64);
            c1 += __shfl_xor(c1, d, 64);
            c2 += __shfl_xor(c2, d, 64);
        }
        if (lane == 0) {
            s_cnt[wave][0] = c0;
            s_cnt[wave][1] = c1;
            s_cnt[wave][2] = c2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        if (threadIdx.x < 28) {
#pragma unroll
            for (int p = 0; p < 8; p++) t += s_part[p][threadIdx.x];
        } else if (threadIdx.x < 31) {
            unsigned long long c = 0;
            if (n_match_blocks)
                for (int w = 0; w < kEvalThreads / 64; w++) c += s_cnt[w][threadIdx.x - 28];
            t = (double)c;
        } else {
            t = (double)n;
        }
        out[threadIdx.x] = t;
        if (threadIdx.x == 0) *ticket = 0u;  // every workgroup has arrived: re-arm for the next launch
        if (mail) {
            mail[threadIdx.x] = t;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: payload before the flag
            // the 32 lanes are one wave: lane 0 publishes after the wave's stores
            if (threadIdx.x == 0)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(mail + 32), seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void pose_args(const float t[3], const float q[4], float max_dist, PoseArgs &P)
{
    float R[9];
    rotation_matrix(q, R);  // voxel_grid.h:212 transform.rotationMatrix().cast<double>()
    for (int i = 0; i < 9; i++) P.R[i] = (double)R[i];
    for (int i = 0; i < 3; i++) P.t[i] = (double)t[i];
    P.max_sq = max_dist * max_dist;  // voxel_grid.h:215
}

constexpr uint32_t kMaxMatchBlocks = 256u * 8u;
static uint32_t match_grid(uint32_t n)
{
    const uint32_t need = (n + kGroupsPerBlock - 1) / kGroupsPerBlock;
    return std::max(1u, std::min(need, kMaxMatchBlocks));
}

static uint32_t eval_grid(uint32_t n)
{
    const uint32_t need = (n + kEvalThreads - 1) / kEvalThreads;
    return std::max(1u, std::min(need, 512u));
}

struct ScanCtx {
    lom_map *m;
    const char *d_src;
    size_t stride;
    uint32_t n;
    uint32_t match_blocks;
    int prof_used = 0;
};

static int scan_buffers(lom_map *m, uint32_t n, bool want_stats)
{
    int rc;
    const size_t nn = std::max<uint32_t>(n, 1);
    if ((rc = ensure(m, m->scan_idx, nn * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scan_on, nn * 24)) != LOM_OK) return rc;
    if (want_stats && (rc = ensure(m, m->scan_stats, nn * sizeof(QStat))) != LOM_OK) return rc;
    if ((rc = ensure(m, m->partials, (size_t)512 * 28 * 8)) != LOM_OK) return rc;
    const void *before = m->results.p;
    if ((rc = ensure(m, m->results, 1024 + (size_t)kMaxMatchBlocks * 16)) != LOM_OK) return rc;
    if (m->results.p != before)  // new allocation: sums and the hand-off ticket start at zero
        LOM_HIP(m, hipMemsetAsync(m->results.p, 0, 1024, m->stream));
    return LOM_OK;
}

static uint32_t *d_block_counters(lom_map *m) { return (uint32_t *)((char *)m->results.p + 1024); }
static double *d_sums(lom_map *m) { return (double *)m->results.p; }
static uint32_t *d_ticket(lom_map *m) { return (uint32_t *)((char *)m->results.p + 512); }

static int launch_match(ScanCtx &c, const float t[3], const float q[4], float max_dist, bool stats)
{
    lom_map *m = c.m;
    PoseArgs P;
    pose_args(t, q, max_dist, P);
    c.match_blocks = c.n ? match_grid(c.n) : 0;
    if (c.n) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (m->profiling) {
            // one event pair per launch, read back once at the end of the align
            while (m->prof_events.size() < (size_t)(c.prof_used + 1) * 2) {
                hipEvent_t e;
                LOM_HIP(m, hipEventCreate(&e));
                m->prof_events.push_back(e);
            }
            e0 = m->prof_events[(size_t)c.prof_used * 2];
            e1 = m->prof_events[(size_t)c.prof_used * 2 + 1];
            c.prof_used++;
            LOM_HIP(m, hipEventRecord(e0, m->stream));
        }
        hipLaunchKernelGGL(k_match<kMatchG>, dim3(c.match_blocks), dim3(kMatchThreads), 0, m->stream, view_of(m), c.d_src,
                           c.stride, c.n, P, (int32_t *)m->scan_idx.p, (float *)m->scan_on.p,
                           stats ? (QStat *)m->scan_stats.p : (QStat *)nullptr, d_block_counters(m));
        LOM_HIP(m, hipGetLastError());
        if (m->profiling) LOM_HIP(m, hipEventRecord(e1, m->stream));
    }
    return LOM_OK;
}

// evaluation at (q,t) -> host sums (rank-local, or rank-ordered total with a communicator)
static int launch_eval(ScanCtx &c, const double q[4], const double t[3], bool fresh_match, double out[LOM_NSUMS])
{
    lom_map *m = c.m;
    EvalArgs E;
    for (int i = 0; i < 4; i++) E.q[i] = q[i];
    for (int i = 0; i < 3; i++) E.t[i] = t[i];
    const uint32_t nb = c.n ? eval_grid(c.n) : 0;
    const bool mailbox = (m->comm == nullptr);
    if (nb == 0) {
        // nothing to evaluate on this rank: all sums are zero
        if (mailbox) {
            std::memset(out, 0, LOM_NSUMS * 8);
            for (int k = 0; k < 4; k++) m->last_counters[k] = 0.0;
            return LOM_OK;
        }
        LOM_HIP(m, hipMemsetAsync(d_sums(m), 0, LOM_NSUMS * 8, m->stream));
    }
    const unsigned long long seq = ++m->mail_seq;
    if (nb) {
        hipLaunchKernelGGL(k_eval, dim3(nb), dim3(kEvalThreads), 0, m->stream, c.d_src, c.stride, c.n,
                           (const int32_t *)m->scan_idx.p, (const float *)m->scan_on.p, E, (double *)m->partials.p,
                           d_ticket(m), (const uint32_t *)d_block_counters(m), fresh_match ? c.match_blocks : 0u,
                           d_sums(m), mailbox ? m->d_mail : (double *)nullptr, seq);
        LOM_HIP(m, hipGetLastError());
    }
    if (!mailbox) {
        int rc = ensure(m, m->gather, (size_t)m->nranks * LOM_NSUMS * 8);
        if (rc != LOM_OK) return rc;
        rc = comm_allgather_sums(m, d_sums(m), (double *)m->gather.p, LOM_NSUMS);
        if (rc != LOM_OK) return rc;
        LOM_HIP(m, hipMemcpyAsync(m->h_results, m->gather.p, (size_t)m->nranks * LOM_NSUMS * 8,
                                  hipMemcpyDeviceToHost, m->stream));
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        for (int k = 0; k < LOM_NSUMS; k++) {
            double v = 0.0;
            for (int r = 0; r < m->nranks; r++) v += m->h_results[(size_t)r * LOM_NSUMS + k];  // rank order
            out[k] = v;
        }
    } else {
        // poll the mailbox; fall back to the stream's status so a failed launch cannot hang us
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(m->h_mail + 32);
        uint64_t spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0) {
                const hipError_t e = hipStreamQuery(m->stream);
                if (e == hipSuccess) {
                    if (*flag == seq) break;
                    if (spins > (1ull << 26)) return set_error(m, LOM_ERR_HIP, "mailbox not written by k_eval");
                } else if (e != hipErrorNotReady) {
                    return set_error(m, LOM_ERR_HIP, "stream failed while waiting for k_eval", e);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        std::memcpy(out, (const void *)m->h_mail, LOM_NSUMS * 8);
    }
    // counters of the last k_match are summed on its first evaluation only
    if (fresh_match) {
        for (int k = 0; k < 4; k++) m->last_counters[k] = out[28 + k];
    } else {
        for (int k = 0; k < 3; k++) out[28 + k] = m->last_counters[k];
    }
    return LOM_OK;
}

static int hook_match_eval(void *user, const float pt[3], const float pq[4], const double q[4], const double t[3],
                           double out[LOM_NSUMS])
{
    ScanCtx &c = *(ScanCtx *)user;
    int rc = launch_match(c, pt, pq, 0.3f, false);  // cloud_matcher.cpp:139
    if (rc != LOM_OK) return rc;
    return launch_eval(c, q, t, true, out);
}

static int hook_eval_fixed(void *user, const double q[4], const double t[3], double out[LOM_NSUMS])
{
    ScanCtx &c = *(ScanCtx *)user;
    return launch_eval(c, q, t, false, out);
}

static int align_device(lom_map *m, const char *d_src, size_t n, size_t stride, const float guess_t[3],
                        const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    if (n >= 0x7FFFFFFFull) return set_error(m, LOM_ERR_ARG, "too many source points");
    int rc = scan_buffers(m, (uint32_t)n, false);
    if (rc != LOM_OK) return rc;
    ScanCtx c{m, d_src, stride, (uint32_t)n, 0};
    lom_align_hooks hooks;
    hooks.user = &c;
    hooks.match_eval = hook_match_eval;
    hooks.eval_fixed = hook_eval_fixed;
    hooks.allreduce = nullptr;  // the rank-ordered all-gather sits inside launch_eval
    lom_align_stats st;
    rc = lom_align_with_hooks(&hooks, guess_t, guess_q, out_t, out_q, &st);
    if (rc != LOM_OK) {
        if (m->last_error.empty()) set_error(m, rc, "align failed");
        return rc == LOM_ERR_HOOK ? LOM_ERR_HIP : rc;
    }
    if (m->profiling && c.prof_used) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        for (int i = 0; i < c.prof_used; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->prof_events[(size_t)i * 2], m->prof_events[(size_t)i * 2 + 1]) == hipSuccess)
                st.match_kernel_ms += ms;
        }
    }
    if (stats) *stats = st;
    return LOM_OK;
}

static int stage_scan(lom_map *m, const float *src, size_t n, size_t stride, const char **d_src)
{
    const size_t bytes = n ? (n - 1) * stride + 12 : 0;
    int rc = ensure(m, m->scan_src, std::max<size_t>(bytes, 16));
    if (rc != LOM_OK) return rc;
    if (bytes) LOM_HIP(m, hipMemcpyAsync(m->scan_src.p, src, bytes, hipMemcpyHostToDevice, m->stream));
    *d_src = (const char *)m->scan_src.p;
    return LOM_OK;
}

}  // namespace lom

using namespace lom;

extern "C" {

int64_t lom_match_find_pairs(lom_map *m, const float *src, size_t n, size_t stride, const float t[3],
                             const float q[4], float max_dist, lom_correspondence *out)
{
    if (!m || (n && (!src || !out)) || !t || !q || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    if (n == 0) return 0;
    LOM_HIP(m, hipSetDevice(m->device));
    const char *d_src = nullptr;
    int rc = stage_scan(m, src, n, stride, &d_src);
    if (rc != LOM_OK) return rc;
    if ((rc = scan_buffers(m, (uint32_t)n, true)) != LOM_OK) return rc;
    ScanCtx c{m, d_src, stride, (uint32_t)n, 0};
    if ((rc = launch_match(c, t, q, max_dist, true)) != LOM_OK) return rc;
    std::vector<int32_t> idx(n);
    std::vector<float> on(n * 6);
    std::vector<QStat> st(n);
    LOM_HIP(m, hipMemcpyAsync(idx.data(), m->scan_idx.p, n * 4, hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipMemcpyAsync(on.data(), m->scan_on.p, n * 24, hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipMemcpyAsync(st.data(), m->scan_stats.p, n * sizeof(QStat), hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    int64_t valid = 0;
    for (size_t i = 0; i < n; i++) {
        lom_correspondence &o = out[i];
        o.index = idx[i];
        std::memcpy(o.origin, &on[i * 6], 12);
        std::memcpy(o.normal, &on[i * 6 + 3], 12);
        o.sq_dist = st[i].sq_dist;
        o.n_cand = st[i].n_cand;
        o.n_occ = st[i].n_occ;
        valid += idx[i] >= 0;
    }
    return valid;
}

int lom_match_align_device(lom_map *m, const float *d_src, size_t n, size_t stride, const float guess_t[3],
                           const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    if (!m || (n && !d_src) || !guess_t || !guess_q || !out_t || !out_q || stride < 12 || (stride & 3))
        return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    m->last_error.clear();
    return align_device(m, (const char *)d_src, n, stride, guess_t, guess_q, out_t, out_q, stats);
}

int lom_match_align(lom_map *m, const float *src, size_t n, size_t stride, const float guess_t[3],
                    const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    if (!m || (n && !src) || !guess_t || !guess_q || !out_t || !out_q || stride < 12 || (stride & 3))
        return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    m->last_error.clear();
    const char *d_src = nullptr;
    int rc = stage_scan(m, src, n, stride, &d_src);
    if (rc != LOM_OK) return rc;
    return align_device(m, d_src, n, stride, guess_t, guess_q, out_t, out_q, stats);
}

}  // extern "C"
