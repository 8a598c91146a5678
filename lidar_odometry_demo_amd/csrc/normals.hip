// Normal estimation helper: the per-point plane fit the reference's matcher test feeds the map with --
// pcl::NormalEstimation with setRadiusSearch(0.25) and the default viewpoint (0, 0, 0) (test/test.cpp:196-205):
// covariance of all points within the radius (the point itself included), eigenvector of the smallest
// eigenvalue, flipped towards the viewpoint; NaN where fewer than 3 neighbours are found (the test drops
// those points, :219-221).  This is the only place where the reference's data flow holds a plane /
// covariance accumulation; it is OUTSIDE the align path, which uses the
// normals it is given.
//
// Device design: the cloud is its own spatial index -- a voxel map with voxel size = radius, so the 27-voxel
// neighbourhood of a point's voxel covers its radius ball exactly.  One query per 16-lane DPP row as in
// k_match: 27 slot probes, the neighbours' points as one flattened candidate sequence, every lane adds the
// first and second moments of its candidates (relative to the query point, f64), the row adds them up with
// DPP butterflies, one lane solves the symmetric 3x3 eigenproblem (cyclic Jacobi) and writes the normal.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "lom_internal.hpp"

namespace lom {

constexpr int kNrmThreads = 256, kNrmG = 16;

template <int kCtrl>
__device__ __forceinline__ double nrm_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double nrm_row_sum(double v)  // all 16 lanes of the row end with the total
{
    v += nrm_dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += nrm_dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += nrm_dpp_f64<0x141>(v);  // row_half_mirror
    v += nrm_dpp_f64<0x140>(v);  // row_mirror
    return v;
}

// eigenvector of the smallest eigenvalue of the symmetric matrix {a00 a01 a02; . a11 a12; . . a22}: cyclic Jacobi
__device__ inline void smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22,
                                            double out[3])
{
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; sweep++) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-18 * diag || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; k++) {  // A <- A G
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {  // A <- G^T A
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {  // V <- V G
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    out[0] = m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
    out[1] = m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
    out[2] = m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

__global__ __launch_bounds__(kNrmThreads) void k_normals(MapView map, const char *__restrict__ xyz, size_t stride, uint32_t n,
                                                         float radius, float *__restrict__ out_nrm,
                                                         uint32_t *__restrict__ out_neigh)
{
    constexpr int kGroups = kNrmThreads / kNrmG;
    __shared__ uint32_t s_pref[kGroups][32], s_base[kGroups][32];
    const int gl = threadIdx.x % kNrmG, grp = threadIdx.x / kNrmG;
    const double r2 = (double)radius * (double)radius;
    for (uint32_t q = blockIdx.x * kGroups + grp; q < n; q += gridDim.x * kGroups) {
        const float *sp = reinterpret_cast<const float *>(xyz + (size_t)q * stride);
        const float qx = sp[0], qy = sp[1], qz = sp[2];
        int ix = 0, iy = 0, iz = 0;
        const bool inr = voxel_index(qx, map.voxel_size, ix) && voxel_index(qy, map.voxel_size, iy) &&
                         voxel_index(qz, map.voxel_size, iz);
        uint32_t cnt[2] = {0, 0}, slab[2] = {0, 0};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int b = gl + s * kNrmG;
            const int nx = ix + b / 9 - 1, ny = iy + (b / 3) % 3 - 1, nz = iz + b % 3 - 1;
            const bool act = inr && b < 27 && nx > -kIdxBias && nx < kIdxBias && ny > -kIdxBias && ny < kIdxBias &&
                             nz > -kIdxBias && nz < kIdxBias;
            if (act) {
                const unsigned long long key = pack_key(nx, ny, nz);
                uint32_t h = hash_key(key, map.shift) & map.mask;
                for (uint32_t probe = 0; probe <= map.mask; probe++) {
                    const Slot sl = map.table[h];
                    if (sl.key == key) {
                        cnt[s] = sl.count;
                        slab[s] = sl.slab;
                        break;
                    }
                    if (sl.key == kEmptyKey) break;
                    h = (h + 1) & map.mask;
                }
            }
        }
        // inclusive prefix of the counts in scan order over the row (two sets of 16)
        uint32_t run = 0;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32_t inc = cnt[s];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 16);
                if (gl >= d) inc += o;
            }
            const int b = gl + s * kNrmG;
            s_pref[grp][b] = (b < 27) ? run + inc : 0xFFFFFFFFu;
            s_base[grp][b] = slab[s] * map.K - (run + inc - cnt[s]);
            run += __shfl(inc, 15, 16);
        }
        const uint32_t T = run;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double m0 = 0.0, s1x = 0.0, s1y = 0.0, s1z = 0.0, sxx = 0.0, sxy = 0.0, sxz = 0.0, syy = 0.0, syz = 0.0, szz = 0.0;
        const uint32_t *pref = s_pref[grp];
        for (uint32_t c = gl; c < T; c += kNrmG) {
            uint32_t b = 0;  // smallest b with pref[b] > c
            b += (pref[b + 15] <= c) ? 16u : 0u;
            b += (pref[b + 7] <= c) ? 8u : 0u;
            b += (pref[b + 3] <= c) ? 4u : 0u;
            b += (pref[b + 1] <= c) ? 2u : 0u;
            b += (pref[b] <= c) ? 1u : 0u;
            const float *vp = map.pts + (size_t)(s_base[grp][b] + c) * 3;
            const double dx = (double)vp[0] - (double)qx, dy = (double)vp[1] - (double)qy, dz = (double)vp[2] - (double)qz;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 <= r2) {  // radius search, the point itself included (d = 0)
                m0 += 1.0;
                s1x += dx, s1y += dy, s1z += dz;
                sxx += dx * dx, sxy += dx * dy, sxz += dx * dz, syy += dy * dy, syz += dy * dz, szz += dz * dz;
            }
        }
        m0 = nrm_row_sum(m0);
        s1x = nrm_row_sum(s1x), s1y = nrm_row_sum(s1y), s1z = nrm_row_sum(s1z);
        sxx = nrm_row_sum(sxx), sxy = nrm_row_sum(sxy), sxz = nrm_row_sum(sxz);
        syy = nrm_row_sum(syy), syz = nrm_row_sum(syz), szz = nrm_row_sum(szz);
        if (gl == 0) {
            float o0 = __uint_as_float(0x7FC00000u), o1 = o0, o2 = o0;  // NaN: fewer than 3 neighbours (test.cpp:219-221 drops the point)
            if (m0 >= 3.0) {
                const double inv = 1.0 / m0;
                const double mx = s1x * inv, my = s1y * inv, mz = s1z * inv;
                double v[3];
                smallest_eigenvector(sxx * inv - mx * mx, sxy * inv - mx * my, sxz * inv - mx * mz, syy * inv - my * my,
                                     syz * inv - my * mz, szz * inv - mz * mz, v);
                // flipNormalTowardsViewpoint, viewpoint (0, 0, 0): (vp - p) . n must not be negative
                const double side = -((double)qx * v[0] + (double)qy * v[1] + (double)qz * v[2]);
                const double sgn = side < 0.0 ? -1.0 : 1.0;
                o0 = (float)(sgn * v[0]), o1 = (float)(sgn * v[1]), o2 = (float)(sgn * v[2]);
            }
            out_nrm[(size_t)q * 3 + 0] = o0;
            out_nrm[(size_t)q * 3 + 1] = o1;
            out_nrm[(size_t)q * 3 + 2] = o2;
            if (out_neigh) out_neigh[q] = (uint32_t)m0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace lom

using namespace lom;

extern "C" {

int64_t lom_estimate_normals(const float *xyz, size_t n, size_t stride, float radius, int device, float *nrm_out,
                             uint32_t *neighbours_out)
{
    if ((n && (!xyz || !nrm_out)) || stride < 12 || (stride & 3) || !(radius > 0.f)) return LOM_ERR_ARG;
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    // the cloud as its own index: voxel = radius; the cap must hold the fullest voxel (the search has to see
    // every point), so it is raised until nothing was dropped
    lom_map *m = nullptr;
    int rc = LOM_OK;
    const size_t caps[] = {64, 256, 1024, 4096, 16384, 65535};
    bool complete = false;
    for (size_t cap : caps) {
        if ((rc = lom_map_create(radius, cap, n / 4 + 1024, device, &m)) != LOM_OK) return rc;
        if ((rc = lom_map_add_points(m, xyz, nullptr, n, stride)) != LOM_OK) break;
        const int64_t stored = lom_map_point_count(m);
        if (stored < 0) {
            rc = (int)stored;
            break;
        }
        if ((size_t)stored == n) {
            complete = true;
            break;
        }
        lom_map_destroy(m);
        m = nullptr;
    }
    if (rc == LOM_OK && !complete) rc = LOM_ERR_RANGE;  // more than 65535 points within one radius-sized voxel
    if (rc != LOM_OK) {
        if (m) lom_map_destroy(m);
        return rc;
    }
    float *d_xyz = nullptr, *d_nrm = nullptr;
    uint32_t *d_cnt = nullptr;
    const size_t bytes = (n - 1) * stride + 12;
    hipError_t e = hipMalloc((void **)&d_xyz, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_nrm, n * 12);
    if (e == hipSuccess && neighbours_out) e = hipMalloc((void **)&d_cnt, n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, bytes, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) {
        const uint32_t groups = kNrmThreads / kNrmG;
        const uint32_t blocks = (uint32_t)std::min<size_t>((n + groups - 1) / groups, 256u * 8u);
        hipLaunchKernelGGL(k_normals, dim3(blocks), dim3(kNrmThreads), 0, m->stream, view_of(m), (const char *)d_xyz, stride,
                           (uint32_t)n, radius, d_nrm, d_cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(nrm_out, d_nrm, n * 12, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess && neighbours_out) e = hipMemcpyAsync(neighbours_out, d_cnt, n * 4, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (d_xyz) (void)hipFree(d_xyz);
    if (d_nrm) (void)hipFree(d_nrm);
    if (d_cnt) (void)hipFree(d_cnt);
    lom_map_destroy(m);
    if (e != hipSuccess) return LOM_ERR_HIP;
    int64_t valid = 0;
    for (size_t i = 0; i < n; i++) valid += nrm_out[3 * i] == nrm_out[3 * i] ? 1 : 0;
    return valid;
}

}  // extern "C"
