// Per-frame front end of LidarOdometry::processCloud on the device -- the callers of the scan-matching
// path, so that a frame stays in HBM from its upload to its pose:
//
//   utils::pointTimeNormalize            reference src/utils/point_time_normalize.h:15-39
//   CloudTransformer::transformNonRigid  reference src/utils/cloud_transform.h:15-40   (deskew)
//   CloudClassifier::classify            reference src/utils/cloud_classifier.h:19-168
//   utils::rangeFilter                   reference src/utils/range_filter.h:13-28
//
// Four kernels, no host decision in between (sizes the host does not know travel as device words):
//   k_fe_stats     min / max of the stamps, points per ring (uint8 ring id, cloud_classifier.h:23)
//   k_fe_deskew    time normalisation, per-point slerp + weighted translation, ring row + azimuth cell,
//                  "last writer wins" per cell as an atomicMax over input indices (:47-55)
//   k_fe_curv      organised cloud (zero points in empty cells) + 9-tap curvature over the FLATTENED
//                  array (:76-103; the window crosses ring boundaries, as in the reference)
//   k_fe_planar    normals from the previous ring (:105-165), range filter, and the compaction of the
//                  surviving planar points in ray-major order (one in-kernel scan, grid_scan.hpp)
//
// Bit-exactness against the host code of odometry.cpp (= oracle/pipeline.c): every f32 / f64 operation
// below is the host's, in the host's order (-ffp-contract=off).  The three library calls are handled
// like this: acos / sin of the FRAME's rotation angle are computed once on the host (glibc); the
// per-point sin((1 - t) theta), sin(t theta) use glibc's own sinf algorithm restated below
// (exhaustively equal to libm's on this image for 0 <= x <= pi/2, tools/check_sinf.c); the double
// atan2 of the azimuth comes from the device library; the host's value lies within 2e-14 of it, and a
// point whose azimuth bin is not the same at both ends of that band raises a flag -- the frame is then
// redone by the host stages (a band of 4e-14 rad against bins of 2 pi / W: ~1e-11 per point).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "grid_scan.hpp"
#include "lom_internal.hpp"
#include "pose_math.hpp"

namespace lom {

constexpr double kPi = 3.14159265358979323846;
constexpr int kFeItems = 4;  // cells per thread of k_fe_planar: up to 4 * 65536 cells per frame

// ---- glibc 2.35 sinf (sysdeps/ieee754/flt-32/s_sinf.c, sincosf.h: ARM optimized-routines) for
// |x| < 120: double-precision polynomial on the reduced argument, result rounded to f32 once ----------
__host__ __device__ inline float glibc_sinf(float y)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    uint32_t bits;
    memcpy(&bits, &y, 4);
    const uint32_t top = (bits >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3F4u) {  // abstop12(y) < abstop12(pi/4)
        const double s = x * x;
        if (top < 0x398u) return y;  // |y| < 2^-12
        const double x3 = x * s;
        const double s1 = S2 + s * S3;
        const double x7 = x3 * s;
        const double ss = x + x3 * S1;
        return (float)(ss + x7 * s1);
    }
    // reduce_fast: quadrant in bits 24..31 of x * (2/pi * 2^24)
    const double r = x * HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * HPI;
    const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
    const double neg = (n & 2) ? -1.0 : 1.0;                         // second table: negated coefficients
    const double xs = x * sg, x2 = x * x;
    if ((n & 1) == 0) {
        const double x3 = xs * x2;
        const double s1 = neg * S2 + x2 * (neg * S3);
        const double x7 = x3 * x2;
        const double ss = xs + x3 * (neg * S1);
        return (float)(ss + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = neg * C3 + x2 * (neg * C4);
    const double c1 = neg * C0 + x2 * (neg * C1);
    const double x6 = x4 * x2;
    const double c = c1 + x4 * (neg * C2);
    return (float)(c + x6 * c2);
}

// what the host prepares per frame: the two poses of transformNonRigid and the frame-level pieces of
// Eigen's Quaternionf::slerp (cloud_transform.h:27)
struct FrameConst {
    float sq[4], eq[4];  // start / end rotation
    float st[3], et[3];  // start / end translation
    float theta, sin_theta;
    int linear;  // |dot| >= 1 - eps: the coefficients are 1 - t and t
    int negate;  // dot < 0: the second coefficient changes sign
    float min_sq, max_sq;  // rangeFilter bounds, squared in f32 (range_filter.h:18-19)
};

// per-frame statistics: two sets, a frame uses set (frame & 1) and clears the other one for its successor
struct FeStats {
    uint32_t ring_count[256];
    uint32_t tmin, tmax;  // order-preserving images of the f32 stamps
    uint32_t pad[6];
};

__device__ __forceinline__ uint32_t f32_ordered(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// words written for the host / the consumers:  [0] planar points  [1] filtered points  [2] H  [3] W
//   [4] fall-back flag (sequence number of the frame that must be redone on the host)  [5] grid error
constexpr int kFeWords = 8;

// `in` may be the pinned host buffer the frame was staged in (read over the host link, once): the kernel then leaves
// the frame in HBM (`keep`) for the kernels behind it -- the upload and the first pass over the frame are one pass,
// without a copy engine's start-up in front of them.
__global__ __launch_bounds__(kThreads) void k_fe_stats(const lom_point_xyzirt *__restrict__ in, uint32_t n, FeStats *mine,
                                                       FeStats *next, lom_point_xyzirt *__restrict__ keep)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_min[kThreads / 64], s_max[kThreads / 64];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const lom_point_xyzirt p = in[i];
        if (keep) keep[i] = p;
        if (p.time == p.time) {  // the host's `<` / `>` scans skip NaN stamps
            const uint32_t o = f32_ordered(p.time);
            lo = o < lo ? o : lo;
            hi = o > hi ? o : hi;
        }
        atomicAdd(&s_hist[(uint8_t)p.ring], 1u);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lo;
        s_max[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&mine->ring_count[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; w++) {
            lo = s_min[w] < lo ? s_min[w] : lo;
            hi = s_max[w] > hi ? s_max[w] : hi;
        }
        atomicMin(&mine->tmin, lo);
        atomicMax(&mine->tmax, hi);
    }
    if (blockIdx.x == 0) {  // the other set goes back to rest for the next frame
        next->ring_count[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            next->tmin = 0xFFFFFFFFu;
            next->tmax = 0u;
        }
    }
}

// rows of the organised cloud: ring ids in ascending order of the uint8 key (std::map<uint8_t, ...>,
// cloud_classifier.h:23,56-66); W = the largest ring (:33-39).  Every workgroup derives them from the
// 256 counters; thread r holds ring r.
__device__ __forceinline__ void ring_layout(const FeStats *st, uint32_t *s_row, uint32_t *s_tmp, uint32_t &H, uint32_t &W)
{
    const uint32_t cnt = st->ring_count[threadIdx.x];
    const uint32_t has = cnt ? 1u : 0u;
    // inclusive scan of `has` and max of `cnt` over the 256 threads
    uint32_t inc = has, mx = cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 63) s_tmp[wave] = inc;
    if (lane == 0) s_tmp[4 + wave] = mx;
    __syncthreads();
    uint32_t off = 0, tot = 0, w = 0;
    for (int k = 0; k < kThreads / 64; k++) {
        if (k < wave) off += s_tmp[k];
        tot += s_tmp[k];
        w = s_tmp[4 + k] > w ? s_tmp[4 + k] : w;
    }
    s_row[threadIdx.x] = off + inc - has;
    H = tot;
    W = w;
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void k_fe_deskew(const lom_point_xyzirt *__restrict__ in, uint32_t n, FrameConst F,
                                                        const FeStats *st, lom_point_xyzirt *__restrict__ desk,
                                                        uint32_t *win, uint32_t cell_cap, uint32_t seq, uint32_t *words)
{
    __shared__ uint32_t s_row[256], s_tmp[8];
    uint32_t H, W;
    ring_layout(st, s_row, s_tmp, H, W);
    const unsigned long long total = (unsigned long long)H * W;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        words[2] = H;
        words[3] = W;
        if (total > cell_cap) __hip_atomic_store(words + 4, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float lo = f32_unordered(st->tmin), hi = f32_unordered(st->tmax);
    const float range = hi - lo;  // point_time_normalize.h:27
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        lom_point_xyzirt p = in[i];
        const float t = (p.time - lo) / range;  // :33
        // Eigen Quaternionf::slerp(t, end) of start (cloud_transform.h:27)
        float s0, s1;
        if (F.linear) {
            s0 = 1.0f - t;
            s1 = t;
        } else {
            s0 = glibc_sinf((1.0f - t) * F.theta) / F.sin_theta;
            s1 = glibc_sinf(t * F.theta) / F.sin_theta;
        }
        if (F.negate) s1 = -s1;
        float q[4], r[3];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = s0 * F.sq[k] + s1 * F.eq[k];
        const float v[3] = {p.x, p.y, p.z};
        quat_rotate<float>(q, v, r);
        const float w1 = (float)(1.0 - (double)t);  // :30
        // the reference weights start.translation by time and end.translation by (1 - time)
        p.x = (r[0] + F.st[0] * t) + F.et[0] * w1;
        p.y = (r[1] + F.st[1] * t) + F.et[1] * w1;
        p.z = (r[2] + F.st[2] * t) + F.et[2] * w1;
        p.time = t;
        desk[i] = p;
        // cloud_classifier.h:49-50: azimuth = atan2(-y, x) + pi (double) narrowed to f32, then the bin
        const double az_d = atan2((double)-p.y, (double)p.x) + kPi;
        const float az = (float)az_d;
        const double binf = fabs((double)(az * (float)W) / (2.0 * kPi));
        {   // could a last-bits difference between this atan2 and the host's change the cell?  The bin is a
            // monotone function of the azimuth: evaluate it at both ends of the band the host's value lies in
            const float az_lo = (float)(az_d - 2e-14), az_hi = (float)(az_d + 2e-14);
            const double b_lo = fabs((double)(az_lo * (float)W) / (2.0 * kPi)), b_hi = fabs((double)(az_hi * (float)W) / (2.0 * kPi));
            const bool in_lo = b_lo < (double)W, in_hi = b_hi < (double)W;
            if (in_lo != in_hi || (in_lo && (uint32_t)b_lo != (uint32_t)b_hi))
                __hip_atomic_store(words + 4, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (binf < (double)W && total <= cell_cap) {  // :52 approx_point_index < max_row_width
            const uint32_t idx = (uint32_t)binf;
            const uint32_t cell = s_row[(uint8_t)p.ring] * W + idx;
            atomicMax(&win[cell], i + 1u);  // the sequential loop's last writer = the largest input index
        }
    }
}

// organised cloud as {x, y, z, curvature} per cell
__global__ __launch_bounds__(kThreads) void k_fe_curv(const lom_point_xyzirt *__restrict__ desk, const uint32_t *__restrict__ win,
                                                      const uint32_t *__restrict__ words, uint32_t cell_cap,
                                                      float4 *__restrict__ org)
{
    const uint32_t total = words[2] * words[3];
    if ((unsigned long long)words[2] * words[3] > cell_cap) return;
    const uint32_t c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= total) return;
    auto cell_point = [&](uint32_t cc, float &x, float &y, float &z, float &inten) {
        const uint32_t j = win[cc];
        x = y = z = inten = 0.f;  // PointType(): empty cells are zero points (:45)
        if (j) {
            const lom_point_xyzirt p = desk[j - 1u];
            x = p.x, y = p.y, z = p.z, inten = p.intensity;
        }
    };
    float x, y, z, inten;
    cell_point(c, x, y, z, inten);
    const uint32_t cw = 4;
    if (c >= cw && c + cw < total) {  // :79 for (i = w; i < size - w; i++)
        const float range = x * x + y * y + z * z;  // :81 (pow(v, 2) is v * v in the reference build)
        if ((double)range < 0.1) {
            inten = 1000.0f;  // :82-85
        } else {
            float dx = (float)((double)(-x) * 9.0);  // :87-89
            float dy = (float)((double)(-y) * 9.0);
            float dz = (float)((double)(-z) * 9.0);
            for (uint32_t w = c - cw; w <= c + cw; w++) {  // :91-95, the centre included
                float ax, ay, az, ai;
                if (w == c)
                    ax = x, ay = y, az = z;
                else
                    cell_point(w, ax, ay, az, ai);
                dx += ax;
                dy += ay;
                dz += az;
            }
            inten = (float)(sqrt((double)(dx * dx + dy * dy + dz * dz)) / (double)range);  // :97
        }
    }
    org[c] = make_float4(x, y, z, inten);
}

template <int kItems>  // cells per thread: 1 while a frame's cells fit one resident grid of 65536 threads, else kFeItems
__global__ __launch_bounds__(kThreads) void k_fe_planar(const float4 *__restrict__ org, uint32_t *win, uint32_t cell_cap,
                                                        FrameConst F, float *__restrict__ out_xyz,
                                                        float *__restrict__ out_nrm, Granule *agg, uint32_t seq,
                                                        uint32_t *words, uint32_t test_fail_from)
{
    __shared__ unsigned long long s_w[8];
    const uint32_t H = words[2], W = words[3];
    const bool overflow = (unsigned long long)H * W > cell_cap;
    const uint32_t total = overflow ? 0u : H * W;
    const uint32_t base = (blockIdx.x * kThreads + threadIdx.x) * kItems;
    const float flat = 0.05f;
    const double flat10 = (double)flat * 10.0;  // :121 flatness_threshold * 10.0
    bool keep[kItems];
    float px[kItems], py[kItems], pz[kItems], nx[kItems], ny[kItems], nz[kItems];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t c = base + k;
        keep[k] = false;
        px[k] = py[k] = pz[k] = nx[k] = ny[k] = nz[k] = 0.f;
        if (c >= total) continue;
        win[c] = 0u;  // the cell table goes back to rest (k_fe_curv has read it)
        const uint32_t ray = c / W, col = c % W;
        if (ray < 1u || col < 4u || col + 4u >= W) continue;  // :107-108
        const float4 pt = org[c];
        if (!(pt.w < flat)) continue;  // :110
        const float4 *row = org + (size_t)(ray - 1u) * W;
        int found = 0;
        float L0 = 0.f, L1 = 0.f, L2 = 0.f, R0 = 0.f, R1 = 0.f, R2 = 0.f;
        // the eight neighbours of the previous ring, all loaded before any is looked at (a loop that stops at the first
        // hit asks for them one round trip after the other); then the reference's two scans over the loaded values
        float4 nbl[4], nbr[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            nbl[t] = row[col - 4u + (uint32_t)t];  // :116 q = col - 4 .. col - 1
            nbr[t] = row[col + 4u - (uint32_t)t];  // :125 q = col + 4 .. col + 1
        }
        bool hit = false;
#pragma unroll
        for (int t = 0; t < 4; t++) {  // :116-123 first from the left
            if (!hit && (double)nbl[t].w < flat10) {
                L0 = nbl[t].x, L1 = nbl[t].y, L2 = nbl[t].z;
                hit = true;
            }
        }
        found += hit ? 1 : 0;
        hit = false;
#pragma unroll
        for (int t = 0; t < 4; t++) {  // :125-132 first from the right
            if (!hit && (double)nbr[t].w < flat10) {
                R0 = nbr[t].x, R1 = nbr[t].y, R2 = nbr[t].z;
                hit = true;
            }
        }
        found += hit ? 1 : 0;
        if (found != 2) continue;
        const float a0 = L0 - pt.x, a1 = L1 - pt.y, a2 = L2 - pt.z;
        const float b0 = R0 - pt.x, b1 = R1 - pt.y, b2 = R2 - pt.z;
        float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;  // :136
        const float zz = sum3(c0 * c0, c1 * c1, c2 * c2);
        if (zz > 0.f) {  // Eigen normalized()
            const float s = sqrtf(zz);
            c0 /= s, c1 /= s, c2 /= s;
        }
        mine += 1ull << 32;  // a planar point (:138-148)
        const float r2 = pt.x * pt.x + pt.y * pt.y + pt.z * pt.z;  // range_filter.h:20-22
        if (r2 >= F.min_sq && r2 <= F.max_sq) {
            keep[k] = true;
            mine += 1ull;
            px[k] = pt.x, py[k] = pt.y, pz[k] = pt.z;
            nx[k] = c0, ny[k] = c1, nz[k] = c2;
        }
    }
    unsigned long long tot;
    const unsigned long long excl = block_scan64(mine, s_w, tot);
    bool gave_up;  // no prefix: nothing is written (the cell table is at rest already); the host stages redo the frame
    const unsigned long long before = grid_prefix64(tot, agg, seq, words + 5, s_w, gave_up, test_fail_from);
    uint32_t at = (uint32_t)(before + excl);  // low word: filtered points before this thread
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        if (!keep[k] || gave_up) continue;
        float *o = out_xyz + (size_t)at * 3, *no = out_nrm + (size_t)at * 3;
        o[0] = px[k], o[1] = py[k], o[2] = pz[k];
        no[0] = nx[k], no[1] = ny[k], no[2] = nz[k];
        at++;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        // a grid that gave up hands an empty cloud on -- also when it was a workgroup in the MIDDLE that gave up and this
        // one still got its prefix (the slow predecessor published in between): part of the output was never written
        const bool hole = __hip_atomic_load(words + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq;
        const unsigned long long all = (gave_up || hole) ? 0ull : before + tot;
        words[0] = (uint32_t)(all >> 32);  // planar points
        words[1] = (uint32_t)all;          // after the range filter
    }
}

__global__ void k_debug_sinf(float *x, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = glibc_sinf(x[i]);
}

}  // namespace lom

using namespace lom;

// ---- host side ------------------------------------------------------------------------------------
struct lom_frontend {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void *h_stage = nullptr;  // pinned bounce buffer for the raw frame
    void *d_stage_view = nullptr;  // ... as the device sees it (looked up once per allocation, not per frame)
    size_t h_stage_bytes = 0;
    hipEvent_t stage_ev = nullptr, done_ev = nullptr;
    lom_point_xyzirt *d_in = nullptr, *d_desk = nullptr;
    uint32_t *d_win = nullptr;
    float4 *d_org = nullptr;
    float *d_xyz = nullptr, *d_nrm = nullptr;
    size_t cap_pts = 0, cap_cells = 0;
    FeStats *d_stats = nullptr;  // [2]
    uint32_t *d_words = nullptr; // kFeWords + aggregates
    uint32_t *h_words = nullptr; // pinned copy
    uint32_t seq = 0;
    uint32_t n_last = 0;
    int test_grid_give_up = -1;  // LOM_OPT_TEST_GRID_GIVE_UP (one shot)
    bool dma_upload = false;     // LOM_FE_DMA_UPLOAD=1 at create
    std::string error;
};

namespace {

int fe_fail(lom_frontend *f, int code, const char *what, hipError_t e = hipSuccess)
{
    f->error = what;
    if (e != hipSuccess) {
        f->error += ": ";
        f->error += hipGetErrorString(e);
    }
    return code;
}

#define FE_HIP(f, expr)                                              \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return fe_fail((f), LOM_ERR_HIP, #expr, _e); \
    } while (0)

Granule *fe_agg(lom_frontend *f) { return reinterpret_cast<Granule *>(f->d_words + 64); }

int fe_reserve(lom_frontend *f, size_t n)
{
    if (n <= f->cap_pts) return LOM_OK;
    FE_HIP(f, hipStreamSynchronize(f->stream));
    for (void *p : {(void *)f->d_in, (void *)f->d_desk, (void *)f->d_win, (void *)f->d_org, (void *)f->d_xyz, (void *)f->d_nrm})
        if (p) (void)hipFree(p);
    f->d_in = f->d_desk = nullptr;
    f->d_win = nullptr;
    f->d_org = nullptr;
    f->d_xyz = f->d_nrm = nullptr;
    f->cap_pts = 0;
    const size_t cap = n + n / 2 + 4096;
    const size_t cells = 3 * cap + 4096;  // rings of unequal size make H * W exceed n; beyond this the host stages take the frame
    FE_HIP(f, hipMalloc((void **)&f->d_in, cap * sizeof(lom_point_xyzirt)));
    FE_HIP(f, hipMalloc((void **)&f->d_desk, cap * sizeof(lom_point_xyzirt)));
    FE_HIP(f, hipMalloc((void **)&f->d_win, cells * 4));
    FE_HIP(f, hipMalloc((void **)&f->d_org, cells * sizeof(float4)));
    FE_HIP(f, hipMalloc((void **)&f->d_xyz, cells * 12));
    FE_HIP(f, hipMalloc((void **)&f->d_nrm, cells * 12));
    FE_HIP(f, hipMemsetAsync(f->d_win, 0, cells * 4, f->stream));  // at rest: k_fe_planar clears what a frame set
    f->cap_pts = cap;
    f->cap_cells = cells;
    return LOM_OK;
}

// the frame-level part of Eigen's slerp (cloud_transform.h:27) with the host's libm
void frame_const(const lom_pose &start, const lom_pose &end, float min_range, float max_range, FrameConst &F)
{
    for (int k = 0; k < 4; k++) F.sq[k] = start.q[k], F.eq[k] = end.q[k];
    for (int k = 0; k < 3; k++) F.st[k] = start.t[k], F.et[k] = end.t[k];
    const float *a = start.q, *b = end.q;
    const float one = 1.0f - 1.1920928955078125e-07f;
    const float d = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    const float ad = std::fabs(d);
    F.linear = ad >= one ? 1 : 0;
    F.negate = d < 0.0f ? 1 : 0;
    F.theta = 0.f;
    F.sin_theta = 1.f;
    if (!F.linear) {
        F.theta = std::acos(ad);
        F.sin_theta = std::sin(F.theta);
    }
    F.min_sq = min_range * min_range;
    F.max_sq = max_range * max_range;
}

}  // namespace

extern "C" {

int lom_frontend_create(int device, void *hip_stream, lom_frontend **out)
{
    if (!out) return LOM_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return LOM_ERR_NO_DEVICE;
    }
    lom_frontend *f = new (std::nothrow) lom_frontend();
    if (!f) return LOM_ERR_OOM;
    f->device = device;
    f->dma_upload = getenv("LOM_FE_DMA_UPLOAD") != nullptr;
    const size_t wbytes = 64 * 4 + 256 * 2 * sizeof(Granule);
    if (hipSetDevice(device) != hipSuccess || (hip_stream == nullptr && hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipMalloc((void **)&f->d_stats, 2 * sizeof(FeStats)) != hipSuccess || hipMalloc((void **)&f->d_words, wbytes) != hipSuccess ||
        hipHostMalloc((void **)&f->h_words, 64 * 4, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&f->stage_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f->done_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        lom_frontend_destroy(f);
        return LOM_ERR_HIP;
    }
    if (hip_stream)
        f->stream = (hipStream_t)hip_stream;
    else
        f->own_stream = true;
    FeStats init[2];
    std::memset(init, 0, sizeof init);
    init[0].tmin = init[1].tmin = 0xFFFFFFFFu;
    if (hipMemcpy(f->d_stats, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(f->d_words, 0, wbytes) != hipSuccess) {
        (void)hipGetLastError();
        lom_frontend_destroy(f);
        return LOM_ERR_HIP;
    }
    *out = f;
    return LOM_OK;
}

void lom_frontend_destroy(lom_frontend *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    for (void *p : {(void *)f->d_in, (void *)f->d_desk, (void *)f->d_win, (void *)f->d_org, (void *)f->d_xyz, (void *)f->d_nrm,
                    (void *)f->d_stats, (void *)f->d_words})
        if (p) (void)hipFree(p);
    if (f->h_stage) (void)hipHostFree(f->h_stage);
    if (f->h_words) (void)hipHostFree(f->h_words);
    if (f->stage_ev) (void)hipEventDestroy(f->stage_ev);
    if (f->done_ev) (void)hipEventDestroy(f->done_ev);
    if (f->own_stream && f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

const char *lom_frontend_last_error(const lom_frontend *f) { return f ? f->error.c_str() : ""; }

int lom_frontend_set_option(lom_frontend *f, int option, int64_t value)
{
    if (!f) return LOM_ERR_ARG;
    if (option == LOM_OPT_TEST_GRID_GIVE_UP && value >= -1 && (value & ~(int64_t)kGridFailOnlyOne) <= 65535) {
        f->test_grid_give_up = (int)value;
        return LOM_OK;
    }
    return LOM_ERR_ARG;
}

int lom_frontend_process(lom_frontend *f, const lom_point_xyzirt *pts, size_t n, const lom_pose *start, const lom_pose *end,
                         float min_range, float max_range)
{
    if (!f || (n && !pts) || !start || !end) return LOM_ERR_ARG;
    // the organised cloud is scanned by one grid of at most kFeItems * 65536 cells, sized n + n / 2 + 4096
    if (n + n / 2 + 4096 > (size_t)kFeItems * kOnePassMax) return fe_fail(f, LOM_ERR_ARG, "frame too large for the device front end");
    FE_HIP(f, hipSetDevice(f->device));
    f->error.clear();
    int rc = fe_reserve(f, std::max<size_t>(n, 1));
    if (rc != LOM_OK) return rc;
    const uint32_t N = (uint32_t)n;
    const uint32_t seq = ++f->seq;
    f->n_last = N;
    // raw frame -> pinned bounce buffer -> HBM (the caller's buffer is free when this returns); a caller that
    // has written the frame into lom_frontend_stage()'s buffer itself passes that pointer and skips the copy
    const size_t bytes = n * sizeof(lom_point_xyzirt);
    if (static_cast<const void *>(pts) != f->h_stage || bytes > f->h_stage_bytes) {
        lom_point_xyzirt *stage = nullptr;
        if ((rc = lom_frontend_stage(f, n, &stage)) != LOM_OK) return rc;
        if (bytes) std::memcpy(static_cast<void *>(stage), pts, bytes);
    }
    // the frame reaches HBM through k_fe_stats, which reads the pinned buffer itself (LOM_FE_DMA_UPLOAD=1 at create: through
    // a copy of its own in front of the kernels, as until round 3 -- on C5 that was ~10 us more per frame)
    const lom_point_xyzirt *stats_in = f->d_in;
    lom_point_xyzirt *stats_keep = nullptr;
    if (f->dma_upload || !bytes) {
        if (bytes) FE_HIP(f, hipMemcpyAsync(f->d_in, f->h_stage, bytes, hipMemcpyHostToDevice, f->stream));
        FE_HIP(f, hipEventRecord(f->stage_ev, f->stream));
    } else {
        if (!f->d_stage_view) FE_HIP(f, hipHostGetDevicePointer(&f->d_stage_view, f->h_stage, 0));
        stats_in = static_cast<const lom_point_xyzirt *>(f->d_stage_view);
        stats_keep = f->d_in;
    }
    FrameConst F;
    frame_const(*start, *end, min_range, max_range, F);
    FeStats *mine = f->d_stats + (seq & 1u), *next = f->d_stats + ((seq + 1u) & 1u);
    const uint32_t cell_cap = (uint32_t)std::min<size_t>(f->cap_cells, (size_t)kFeItems * kOnePassMax);
    const uint32_t pt_blocks = std::max(1u, std::min(blocks_for(N), 1024u));
    hipLaunchKernelGGL(k_fe_stats, dim3(pt_blocks), dim3(kThreads), 0, f->stream, stats_in, N, mine, next, stats_keep);
    if (stats_keep) FE_HIP(f, hipEventRecord(f->stage_ev, f->stream));  // the staging buffer is free once this kernel has read it
    // the organised cloud has H * W cells, known on the device only: the grids cover what a frame of n points
    // normally needs (rings of equal size: H * W ~ n) with a margin; a larger cloud raises the fall-back flag
    const uint32_t cells_bound = (uint32_t)std::min<size_t>(cell_cap, (size_t)N + N / 2 + 4096);
    hipLaunchKernelGGL(k_fe_deskew, dim3(pt_blocks), dim3(kThreads), 0, f->stream, f->d_in, N, F, mine, f->d_desk, f->d_win,
                       cells_bound, seq, f->d_words);
    hipLaunchKernelGGL(k_fe_curv, dim3(blocks_for(cells_bound)), dim3(kThreads), 0, f->stream, f->d_desk, f->d_win,
                       f->d_words, cells_bound, f->d_org);
    const uint32_t fail_from = f->test_grid_give_up < 0 ? 0xFFFFFFFFu : (uint32_t)f->test_grid_give_up;
    f->test_grid_give_up = -1;
    if (cells_bound <= kOnePassMax)
        hipLaunchKernelGGL(k_fe_planar<1>, dim3(blocks_for(cells_bound)), dim3(kThreads), 0, f->stream, f->d_org, f->d_win,
                           cells_bound, F, f->d_xyz, f->d_nrm, fe_agg(f), seq, f->d_words, fail_from);
    else
        hipLaunchKernelGGL(k_fe_planar<kFeItems>, dim3(blocks_for((cells_bound + kFeItems - 1) / kFeItems)), dim3(kThreads), 0,
                           f->stream, f->d_org, f->d_win, cells_bound, F, f->d_xyz, f->d_nrm, fe_agg(f), seq, f->d_words,
                           fail_from);
    FE_HIP(f, hipGetLastError());
    FE_HIP(f, hipEventRecord(f->done_ev, f->stream));
    return LOM_OK;
}

// pinned staging buffer for a frame of n points (valid until the next lom_frontend_stage / process of a larger
// frame); waits until the previous frame's upload has read it
int lom_frontend_stage(lom_frontend *f, size_t n, lom_point_xyzirt **out)
{
    if (!f || !out) return LOM_ERR_ARG;
    FE_HIP(f, hipSetDevice(f->device));
    const size_t bytes = n * sizeof(lom_point_xyzirt);
    FE_HIP(f, hipEventSynchronize(f->stage_ev));
    if (bytes > f->h_stage_bytes) {
        if (f->h_stage) FE_HIP(f, hipHostFree(f->h_stage));
        f->h_stage = nullptr;
        f->d_stage_view = nullptr;
        f->h_stage_bytes = 0;
        const size_t nb = std::max(bytes + bytes / 2, (size_t)1 << 20);
        FE_HIP(f, hipHostMalloc(&f->h_stage, nb, hipHostMallocDefault));
        f->h_stage_bytes = nb;
    }
    *out = static_cast<lom_point_xyzirt *>(f->h_stage);
    return LOM_OK;
}

// hipEvent_t recorded behind the last frame's kernels: a consumer on another stream waits for it
void *lom_frontend_done_event(lom_frontend *f) { return f ? (void *)f->done_ev : nullptr; }

// device-side results of the last lom_frontend_process: the filtered planar cloud (packed xyz, normals),
// an upper bound of its size known to the host, and the words {planar, filtered, H, W, fall-back, error}
int lom_frontend_results(lom_frontend *f, const float **d_xyz, const float **d_nrm, const uint32_t **d_counts,
                         uint32_t *bound)
{
    if (!f) return LOM_ERR_ARG;
    if (d_xyz) *d_xyz = f->d_xyz;
    if (d_nrm) *d_nrm = f->d_nrm;
    if (d_counts) *d_counts = f->d_words;
    if (bound) *bound = f->n_last;
    return LOM_OK;
}

void *lom_frontend_stream(lom_frontend *f) { return f ? (void *)f->stream : nullptr; }
uint32_t lom_frontend_sequence(const lom_frontend *f) { return f ? f->seq : 0u; }

// test hook: the device's restatement of glibc's sinf on n host values
int lom_debug_sinf(lom_frontend *f, const float *x, size_t n, float *out)
{
    if (!f || (n && (!x || !out))) return LOM_ERR_ARG;
    FE_HIP(f, hipSetDevice(f->device));
    float *d = nullptr;
    FE_HIP(f, hipMalloc((void **)&d, std::max<size_t>(n, 1) * 4));
    hipError_t e = hipMemcpyAsync(d, x, n * 4, hipMemcpyHostToDevice, f->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_sinf, dim3(blocks_for(std::max<size_t>(n, 1))), dim3(kThreads), 0, f->stream, d, (uint32_t)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, n * 4, hipMemcpyDeviceToHost, f->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fe_fail(f, LOM_ERR_HIP, "lom_debug_sinf", e);
    return LOM_OK;
}

// waits for the frame and returns its verdict: LOM_OK, or 1 when the frame has to be redone by the host stages
// (an azimuth on a rounding boundary, an organised cloud larger than the device buffers), or a negative status
int lom_frontend_wait(lom_frontend *f, uint32_t counts_out[4])
{
    if (!f) return LOM_ERR_ARG;
    FE_HIP(f, hipSetDevice(f->device));
    FE_HIP(f, hipMemcpyAsync(f->h_words, f->d_words, kFeWords * 4, hipMemcpyDeviceToHost, f->stream));
    FE_HIP(f, hipStreamSynchronize(f->stream));
    if (counts_out)
        for (int k = 0; k < 4; k++) counts_out[k] = f->h_words[k];
    // a grid that gave up has written nothing and left the cell table at rest: the frame goes to the host stages
    return (f->h_words[4] == f->seq || f->h_words[5] == f->seq) ? 1 : LOM_OK;
}

// copies of the device results for callers on the host (getTempCloud, tests): what = 0 the deskewed cloud
// (n records), 1 the filtered planar cloud (xyz + normals)
int64_t lom_frontend_fetch(lom_frontend *f, int what, void *out_a, void *out_b, size_t cap)
{
    if (!f || what < 0 || what > 1) return LOM_ERR_ARG;
    FE_HIP(f, hipSetDevice(f->device));
    if (what == 0) {
        const size_t n = f->n_last, take = std::min(n, cap);
        if (take && out_a)
            FE_HIP(f, hipMemcpyAsync(out_a, f->d_desk, take * sizeof(lom_point_xyzirt), hipMemcpyDeviceToHost, f->stream));
        FE_HIP(f, hipStreamSynchronize(f->stream));
        return (int64_t)n;
    }
    uint32_t counts[4];
    const int rc = lom_frontend_wait(f, counts);
    if (rc < 0) return rc;
    const size_t n = counts[1], take = std::min(n, cap);
    if (take && out_a) FE_HIP(f, hipMemcpyAsync(out_a, f->d_xyz, take * 12, hipMemcpyDeviceToHost, f->stream));
    if (take && out_b) FE_HIP(f, hipMemcpyAsync(out_b, f->d_nrm, take * 12, hipMemcpyDeviceToHost, f->stream));
    FE_HIP(f, hipStreamSynchronize(f->stream));
    return (int64_t)n;
}

}  // extern "C"
