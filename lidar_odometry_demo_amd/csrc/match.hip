// Scan-matching kernels: the MI355X counterpart of
//   VoxelGrid::getCorrespondence / findMatchingPairs  (src/voxel_grid.h:164-234)
//   PointToPlaneErrorAnalytic::Evaluate               (src/cloud_matcher.cpp:38-103)
// plus the reduction of the robustified normal equations that Ceres performs
// inside ceres::Solve (DENSE_QR) for the reference.
//
//   k_match   per source point: f64 transform -> f32 query -> 27-neighbour
//             voxel lookup -> nearest stored point (strict-min, scan order
//             ix,iy,iz then insertion order) -> winner's point+normal.
//             A gather that is VALU-issue- and latency-bound at scan sizes;
//             no MFMA (nothing here is a contraction).
//   k_lm      single GPU: one whole ceres::Solve per launch.  Per evaluation and valid
//             correspondence: r = (q*p + t - o).n, 1x6 tangent Jacobian, Huber(0.15) IRLS
//             weight; f64 reduction of sum w J J^T (21), sum w J r (6), sum 0.5 rho (1) per
//             workgroup, exchange between the workgroups through HBM, then the
//             Levenberg-Marquardt policy of lm_core.hpp on one wave.  The pose of the next
//             k_match travels through AlignState in HBM: no host round trip inside an align.
//   k_eval_server / k_eval
//             the same evaluation for the host-driven loop (ranks that exchange sums,
//             LOM_HOST_LM=1): the <= 64 records land in pinned host memory and the host adds
//             them in workgroup order, or stay in HBM for the RCCL all-gather.
//
// Built with -ffp-contract=off (see voxel_map.hip).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "lm_core.hpp"
#include "lm_wave.hpp"
#include "lom_internal.hpp"
#include "pose_math.hpp"

namespace lom {

constexpr int kMatchThreads = 256;             // 4 waves
constexpr int kMatchG = 16;                    // lanes per query: four queries per wave
constexpr int kMatchRows = 4;                  // consecutive rows of a voxel per chunk: one search and 48 bytes per lane and trip
constexpr int kMatchMinWaves = 7;              // waves per SIMD the register budget is held to (72 VGPRs)
constexpr int kEvalThreads = 512;

// what k_match leaves behind for the evaluations of one outer iteration: source point,
// winner's stored point and normal, 48 bytes = three dwordx4 (coalesced for k_eval)
// (the winner's point and the valid flag share one dwordx4: the next outer iteration's k_match reads exactly that
// quarter back as its temporal pruning bound)
struct __attribute__((aligned(16))) MatchRec {
    float px, py, pz, nx;     // source_point_local (voxel_grid.h:226), plane_normal.x
    float ox, oy, oz, valid;  // plane_origin; valid: 0.0f = no match, else the bits kRecValid | the winner's row in the slabs
                              // (never zero, never a denormal: consumers test `!= 0.f`; the next search of the same scan
                              // reads the row back: a query whose winner has not changed leaves its record alone)
    float ny, nz, pad0, pad1;
};
static_assert(sizeof(MatchRec) == 48, "three dwordx4");
constexpr uint32_t kRecValid = 0x40000000u;  // rows below 2^30 are told apart (a larger map still matches, it only rewrites)

// per-query debug record written by k_match for lom_match_find_pairs
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
//  3. flatten  the remaining voxels' points, cut into chunks of up to four consecutive
//              rows of one voxel, form one chunk sequence in scan order (inclusive prefix
//              of the chunk counts in LDS); lane l takes chunks l, l+G, ... and finds each
//              one's voxel by a 5-step binary search -- one search, one address and 48 bytes
//              in flight (three dwordx4) per four candidates.
//  4. select   private strict minimum per lane (candidates arrive in scan order),
//              then the lexicographic minimum of (sq_dist, candidate ordinal) over
//              the group == "first encountered wins" of voxel_grid.h:183-191.
// ---------------------------------------------------------------------------
// Pruning bound along one axis, once per query: squared lower bounds of |q - x| over the
// coordinates x of the neighbour voxels i-1 (gm2) and i+1 (gp2).  Coordinates with
// (int)(x / vs) == j lie in [lo_j, hi_j] (truncation: index 0 is double width), so voxel
// i+1 starts at (i >= 0 ? i+1 : i) * vs and voxel i-1 ends at (i <= 0 ? i-1 : i) * vs.
// (slack_vs = 1e-4 * vs comes from the caller as a wave-uniform value in a scalar register: left to the compiler it was
// hoisted into a vector register and, under the register budget, spilled -- and the reload's s_waitcnt vmcnt(0) then
// waited for every global load in flight, the next query's prefetch included)
__device__ __forceinline__ void axis_gaps(float q, int i, float vs, float slack_vs, float &gm2, float &gp2)
{
    const float fi = (float)i;
    const float face_p = ((i >= 0) ? fi + 1.f : fi) * vs;
    const float face_m = ((i <= 0) ? fi - 1.f : fi) * vs;
    // slack for the f32 rounding of x / vs at the voxel faces and of the distance itself
    const float slack = slack_vs + 1e-6f * fabsf(q);
    const float gp = fmaxf((face_p - q) - slack, 0.f);
    const float gm = fmaxf((q - face_m) - slack, 0.f);
    gm2 = gm * gm;
    gp2 = gp * gp;
}

// One query group == one 16-lane DPP row: shifts, butterflies and mirrors inside the row are
// VALU operand modifiers (no LDS crossbar trip, no index registers).  Lanes shifted in from
// outside the row read 0.  Every lane of a row is active wherever these are used.
template <int kCtrl>
__device__ __forceinline__ uint32_t row_dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
__device__ __forceinline__ uint32_t row_scan_inclusive(uint32_t v)
{
    v += row_dpp<0x111>(v);  // row_shr:1
    v += row_dpp<0x112>(v);  // row_shr:2
    v += row_dpp<0x114>(v);  // row_shr:4
    v += row_dpp<0x118>(v);  // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t row_sum(uint32_t v)
{
    v += row_dpp<kDppXor1>(v);
    v += row_dpp<kDppXor2>(v);
    v += row_dpp<kDppHalfMirror>(v);  // pairs the two quads of a half
    v += row_dpp<kDppMirror>(v);      // pairs the two halves
    return v;
}
__device__ __forceinline__ uint32_t row_min32(uint32_t v)
{
    v = min(v, row_dpp<kDppXor1>(v));
    v = min(v, row_dpp<kDppXor2>(v));
    v = min(v, row_dpp<kDppHalfMirror>(v));
    return min(v, row_dpp<kDppMirror>(v));
}
// lane kLane (0..15) of the row, to every lane of the row (ds_swizzle bit mode: and 0x10, or kLane)
template <int kLane>
__device__ __forceinline__ uint32_t row_lane(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (kLane << 5));
}
// lane 15 of the row, to every lane of the row (ds_swizzle bit mode: and 0x10, or 0x0F)
__device__ __forceinline__ uint32_t row_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F0); }

template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v)  // the value of the DPP partner lane
{
    return __hiloint2double((int)row_dpp<kCtrl>((uint32_t)__double2hiint(v)),
                            (int)row_dpp<kCtrl>((uint32_t)__double2loint(v)));
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Two 16-byte slot loads in flight together, each ONE dwordx4 (the compiler otherwise splits a
// slot into a key load and a dependent count/slab load: two round trips per hit).
__device__ __forceinline__ void load_slots2(const Slot *a, const Slot *b, u32x4 &ra, u32x4 &rb)
{
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(a), "v"(b)
                 : "memory");
}
__device__ __forceinline__ u32x4 load_slot(const Slot *a)
{
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
    return r;
}

typedef float f32x3 __attribute__((ext_vector_type(3)));
// Four consecutive 12-byte rows = 48 bytes from one address as three 16-byte loads in flight together.  What the
// vector L1 charges a load instruction is, per four consecutive lanes, the 128-byte lines they touch
// (tools/microbench/tcp_lines.hip): three instructions over a lane's 48 bytes cost 3/4 of what four 12-byte loads do.
// The address is a multiple of 4, not of 16 unless K % 4 == 0: global_load_dwordx4 takes that on gfx950 (the driver
// runs the memory pipeline in unaligned mode; tools/microbench/unaligned_x4.hip checks it).
__device__ __forceinline__ void load_chunk48(const float *a, f32x3 (&r)[4])
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v0, v1, v2;
    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2)
                 : "v"(a)
                 : "memory");
    r[0] = f32x3{v0.x, v0.y, v0.z};
    r[1] = f32x3{v0.w, v1.x, v1.y};
    r[2] = f32x3{v1.z, v1.w, v2.x};
    r[3] = f32x3{v2.y, v2.z, v2.w};
}

// Two 12-byte loads issued back to back and waited for together (the winner's point and normal).  The loads of a
// trip are asm blocks because, left to the compiler, the first use of load 1 was scheduled ahead of the address
// computation of load 2: the "two loads in flight" of round 2 were two dependent round trips (C2 / C3 / C4: 7.9 / 29.9
// / 53.7 us; issued together 6.8 / 28.1 / 47.7 us, profiles/r03_b_*).
__device__ __forceinline__ void load_points2(const float *a, const float *b, f32x3 &ra, f32x3 &rb)
{
    asm volatile("global_load_dwordx3 %0, %2, off\n\tglobal_load_dwordx3 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(a), "v"(b)
                 : "memory");
}

// kStamp = true is a diagnostic build (lom_debug_match_stamps): thread 0 of every workgroup records
// the shader clock after each phase of its first query, every wait fully drained before a stamp.
// Its run time is not representative; the product launches kStamp = false only.
template <bool kOn>
struct Stamper {  // product build: nothing
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void flush(unsigned long long *) {}
};
template <>
struct Stamper<true> {
    unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool first = true;
    __device__ __forceinline__ void mark(int i)
    {
        if (!first) return;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) t[i] = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void first_done() { first = false; }
    __device__ __forceinline__ void flush(unsigned long long *out)
    {
        if (threadIdx.x != 0) return;
        t[7] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; i++) out[(size_t)blockIdx.x * 8 + i] = t[i];
    }
};
#define LOM_STAMP(i) stamper.mark(i)

// kChained: the pose comes from the AlignState a previous k_lm left in HBM (read through the
// constant address space: scalar loads, like kernel arguments), and the launch does nothing once
// the outer loop has finished -- the host enqueues several outer iterations ahead.
// kPrev: the records of the PREVIOUS search of the same scan against the same map are still at out_rec (outer
// iterations >= 2 of an align): where the old winner still lies in the query's 27 voxels, its f32 distance at the new
// pose bounds this search's minimum from above, and a neighbour voxel whose nearest face is provably farther than that
// cannot hold the winner -- see "temporal bound" in the loop.  Exact.
// kCount: the reference-algorithm counts per query (occupied voxels among the 27, their stored points: SURVEY.md 8d's
// cand(q), the tests' n_cand / n_occ) need every one of the 27 slots.  Without them (the product's align, unless
// LOM_OPT_COUNT_CANDIDATES asks) a neighbour voxel that the bound prunes is not even looked up: its slot is neither
// hashed nor loaded -- the result cannot depend on whether a voxel exists whose points could not win.
template <int G, int kU, int kMinWaves, bool kStamp = false, bool kChained = false, bool kPrev = kChained, bool kCount = true>
__global__ __launch_bounds__(kMatchThreads, kMinWaves) void k_match(MapView map, const char *__restrict__ src, size_t stride,
                                                         uint32_t n, PoseArgs Parg, int32_t *__restrict__ out_idx,
                                                         MatchRec *__restrict__ out_rec,
                                                         QStat *__restrict__ out_stat,
                                                         uint32_t *__restrict__ block_counters,
                                                         unsigned long long *__restrict__ stamps = nullptr,
                                                         const AlignState *state = nullptr)
{
    static_assert(G == 16 && kU == 4, "one query per 16-lane DPP row, a chunk of four rows per lane and trip");
    constexpr uint32_t kRowsLog2 = 2;  // rows per chunk
    // the first query's source point is on its way before anything else: the chained form's pose comes through a
    // scalar-cache miss of its own, and the LDS tables below need a barrier -- one memory round trip instead of two
    // ahead of the first probe (the loop fetches the next query's point the same way, behind the current one's work)
    constexpr int kGroups0 = kMatchThreads / G;
    // (the chained form: the pointer to the state comes in with the kernel's first argument loads, not in a round trip
    // of its own between the point's load and the pose's)
    if constexpr (kChained) asm volatile("" ::"s"(state));
    const uint32_t q_first = blockIdx.x * kGroups0 + threadIdx.x / G;
    f32x3 sp_next = {0.f, 0.f, 0.f};
    // (without the counts the temporal bound decides which slots are loaded at all: the previous record travels with the
    // source point, one query ahead; with them it is only needed once the slots are back)
    constexpr bool kPrevEarly = kPrev && !kCount;
    float4 pv_next = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q_first < n) {
        sp_next = *reinterpret_cast<const f32x3 *>(src + (size_t)q_first * stride);
        if constexpr (kPrevEarly) pv_next = reinterpret_cast<const float4 *>(out_rec + q_first)[1];
    }
    struct {
        double R[9], t[3];
        float max_sq;
    } P;
    if constexpr (kChained) {
        typedef const __attribute__((address_space(4))) AlignState *ConstState;
        ConstState cs = (ConstState)(state);
        // pose and stop flags in ONE scalar round trip (the flags first and the pose behind the branch were two)
#pragma unroll
        for (int i = 0; i < 9; i++) P.R[i] = cs->P.R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) P.t[i] = cs->P.t[i];
        P.max_sq = cs->P.max_sq;
        const int stop = cs->finished | cs->error;
        asm volatile("" ::"s"(P.max_sq), "s"(stop), "s"(P.R[0]), "s"(P.R[1]), "s"(P.R[2]), "s"(P.R[3]), "s"(P.R[4]), "s"(P.R[5]),
                     "s"(P.R[6]), "s"(P.R[7]), "s"(P.R[8]), "s"(P.t[0]), "s"(P.t[1]), "s"(P.t[2]));  // all loaded before the branch
        if (stop) return;
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) P.R[i] = Parg.R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) P.t[i] = Parg.t[i];
        P.max_sq = Parg.max_sq;
    }
    Stamper<kStamp> stamper;
    LOM_STAMP(0);
    constexpr int kGroups = kMatchThreads / G;
    constexpr int kSets = 2;                  // neighbours b = gl (set 0) and 16 + gl (set 1) < 27
    // per neighbour b in scan order: .z inclusive prefix of the scanned CHUNKS (entries >= 27: never reached),
    // .x slab * K - 4 * exclusive prefix, so that chunk ch of the flattened sequence starts at row .x + 4 * ch,
    // .y count + 4 * exclusive prefix: .y - 4 * ch rows of the voxel remain from there
    __shared__ uint4 s_pb[kGroups][32];
    __shared__ uint32_t s_cnt[kGroups][4];
    __shared__ double s_pose[12];             // [component][R row (3), t]: what the component lanes multiply with
    __shared__ float s_gap[kGroups][12];      // per query [axis][to voxel i-1, 0, to voxel i+1]: squared pruning gaps
    const int gl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const uint32_t groups_total = gridDim.x * kGroups;
    // per-group counters live in LDS (one ds_add per counter and query by the writing lane):
    // four fewer live registers keep the kernel at 64 VGPRs without spilling
    if (gl < 4) s_cnt[grp][gl] = 0u;
    if (threadIdx.x < 12) {
        const int c = threadIdx.x >> 2, k = threadIdx.x & 3;
        double v = P.t[0];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                if (c == cc && k == kk) v = kk < 3 ? P.R[cc * 3 + kk] : P.t[cc];
        s_pose[threadIdx.x] = v;
    }
    __syncthreads();
    // Work that is the same for the lanes of a query is split over them instead of repeated by each: lanes
    // 0, 1, 2 of a row prepare the x, y, z component (f64 transform, f32 cast, truncating index, the two
    // pruning gaps of that axis) and hand the results to the row -- values through ds_swizzle broadcasts, the
    // gap table through 12 LDS words.  Lanes 3..15 repeat component z (same instruction stream, results unused).
    const int comp = gl < 2 ? gl : 2;
    const double *my_pose = s_pose + comp * 4;
    // this lane's neighbours (scan order ix, iy, iz): key and hash of a neighbour follow from the centre's by ADDING a lane
    // constant -- pack_key is a sum of shifted fields, and the Fibonacci hash multiplies by a constant modulo 2^64, so
    // hash(key0 + d) = (key0 * phi + d * phi) >> shift.  One 64-bit multiply per query, no per-neighbour packing.
    // (the products are kept opaque: under the 72-register budget the compiler otherwise folds prod0 + dprod back into
    // (key0 + dkey) * phi -- two quarter-rate multiplies and a 64-bit mad per neighbour -- to save their four registers;
    // the three gap-table addresses of a neighbour travel as byte offsets packed into one register instead)
    constexpr unsigned long long kPhi = 0x9E3779B97F4A7C15ull;
    unsigned long long dkey[kSets], dprod[kSets];
    uint32_t gap_off[kSets];
#pragma unroll
    for (int s = 0; s < kSets; s++) {
        const int b = gl + s * G;
        const int dx = b / 9 - 1, dy = (b / 3) % 3 - 1, dz = b % 3 - 1;
        dkey[s] = (unsigned long long)(((long long)dx << 42) + ((long long)dy << 21) + (long long)dz);
        dprod[s] = dkey[s] * kPhi;
        asm volatile("" : "+v"(dprod[s]));
        const uint32_t ox = 4u * (uint32_t)(0 + (b < 27 ? dx + 1 : 1)), oy = 4u * (uint32_t)(3 + (b < 27 ? dy + 1 : 1)),
                       oz = 4u * (uint32_t)(6 + (b < 27 ? dz + 1 : 1));
        gap_off[s] = ox | (oy << 8) | (oz << 16);
    }
    const char *gap_base = reinterpret_cast<const char *>(&s_gap[grp][0]);
    if (gl < 3) s_gap[grp][gl * 3 + 1] = 0.f;  // the centre column of the gap table never changes (own group, own wave)

    const float slack_vs = map.prune_slack;  // 1e-4f * voxel_size
    for (uint32_t q = blockIdx.x * kGroups + grp; q < n; q += groups_total) {
        f32x3 sp = sp_next;
        // the previous search's {winner point, valid} of this query: not needed before the slots are back, so it is
        // asked for here (one round trip beside theirs) rather than a query ahead (four more live registers)
        float4 pv = pv_next;
        if constexpr (kPrev && !kPrevEarly) pv = reinterpret_cast<const float4 *>(out_rec + q)[1];
        if (q + groups_total < n) {
            sp_next = *reinterpret_cast<const f32x3 *>(src + (size_t)(q + groups_total) * stride);
            if constexpr (kPrevEarly) pv_next = reinterpret_cast<const float4 *>(out_rec + (q + groups_total))[1];
        }
        const double p0 = (double)sp.x, p1 = (double)sp.y, p2 = (double)sp.z;
        // voxel_grid.h:220-223: R*p + t in f64 (Eigen order a0 + (a1 + a2)), cast to f32 -- this lane's component
        const float qc = (float)((my_pose[0] * p0 + (my_pose[1] * p1 + my_pose[2] * p2)) + my_pose[3]);
        int ic = 0;
        const bool okc = voxel_index_fast(qc, map.voxel_size, map.inv_voxel_size, ic);
        float gm2, gp2;
        axis_gaps(qc, ic, map.voxel_size, slack_vs, gm2, gp2);
        if (gl < 3) {
            s_gap[grp][gl * 3 + 0] = gm2;
            s_gap[grp][gl * 3 + 2] = gp2;
        }
        const int icc = okc ? ic : (int)0x80000000;  // out of range / not finite
        // temporal bound, part 1 (this lane's axis): does the old winner's own voxel index -- the expression the insert
        // stored it under -- lie within one of the new centre's?  (lanes 3..15 repeat axis z, as above)
        uint32_t near_c = 0u;
        if constexpr (kPrev) {
            const float oc = gl == 0 ? pv.x : (gl == 1 ? pv.y : pv.z);
            int io = 0;
            const bool oko = voxel_index_fast(oc, map.voxel_size, map.inv_voxel_size, io);
            near_c = (okc && oko && (uint32_t)(io - ic + 1) <= 2u) ? 1u : 0u;
        }
        const float qx = __uint_as_float(row_lane<0>(__float_as_uint(qc)));
        const float qy = __uint_as_float(row_lane<1>(__float_as_uint(qc)));
        const float qz = __uint_as_float(row_lane<2>(__float_as_uint(qc)));
        const int ix = (int)row_lane<0>((uint32_t)icc), iy = (int)row_lane<1>((uint32_t)icc),
                  iz = (int)row_lane<2>((uint32_t)icc);
        const bool inr = ix != (int)0x80000000 && iy != (int)0x80000000 && iz != (int)0x80000000;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        LOM_STAMP(1);  // source point loaded and transformed
        // ---- temporal bound, part 2 ----
        // The winner of the previous search (same scan, same map, the pose one solve earlier) is a stored point.  If its
        // voxel index lies within one of this query's centre index on every axis it is one of this query's candidates in
        // the reference (voxel_grid.h:175-183), so its distance d_prev2 -- the very f32 expression of the candidate loop
        // -- is an upper bound of this search's minimum B.  A voxel whose nearest face is provably farther than B holds
        // only points with d2 > B (the argument of the plain bound with B in place of max_sq, same slack): they can
        // neither win nor tie, so every candidate with d2 <= B -- the old winner among them -- is scanned in the
        // reference's order and the first strict minimum is the reference's (:183-191).  An old winner outside the 27
        // voxels, or beyond max_dist, or none: the plain bound.  With kCount the counts (n_cand, n_occ) stay the slot
        // counts of all 27 voxels.
        auto temporal_bound = [&]() -> float {
            float Bv = P.max_sq;
            if constexpr (kPrev) {
                const float ex = qx - pv.x, ey = qy - pv.y, ez = qz - pv.z;
                const float d_prev2 = ex * ex + (ey * ey + ez * ez);
                const bool near = row_min32(near_c) != 0u;  // all three axes (lanes 2..15 hold axis z)
                if (near && pv.w != 0.f && d_prev2 < P.max_sq) Bv = d_prev2;  // (NaN: no bound)
            }
            return Bv;
        };
        float B = P.max_sq;
        if constexpr (!kCount) B = temporal_bound();
        // ---- probe ----
        // stored indices lie in (-2^20, 2^20): a centre at least two voxels inside has all 27 neighbours in range
        const uint32_t kInner = (uint32_t)(2 * kIdxBias - 3);
        const bool safe = (uint32_t)(ix + (kIdxBias - 2)) < kInner && (uint32_t)(iy + (kIdxBias - 2)) < kInner &&
                          (uint32_t)(iz + (kIdxBias - 2)) < kInner;
        const unsigned long long key0 = inr ? pack_key(ix, iy, iz) : 0ull;
        const unsigned long long prod0 = key0 * kPhi;
        uint32_t cnt[kSets], scan_cnt[kSets], slab[kSets];
        unsigned long long key[kSets];
        uint32_t h[kSets];
        bool act[kSets];
        float lower[kSets];
#pragma unroll
        for (int s = 0; s < kSets; s++) {
            const int b = gl + s * G;
            act[s] = inr && b < 27;
            if (act[s] && !safe) {  // the outermost index layers: neighbours beyond the range cannot exist
                const int nx = ix + (b / 9 - 1), ny = iy + ((b / 3) % 3 - 1), nz = iz + (b % 3 - 1);
                act[s] = nx > -kIdxBias && nx < kIdxBias && ny > -kIdxBias && ny < kIdxBias && nz > -kIdxBias &&
                         nz < kIdxBias;
            }
            key[s] = act[s] ? key0 + dkey[s] : 0ull;
            h[s] = act[s] ? ((uint32_t)((prod0 + dprod[s]) >> map.shift) & map.mask) : 0u;
            lower[s] = *reinterpret_cast<const float *>(gap_base + (gap_off[s] & 0xFFu)) +
                       (*reinterpret_cast<const float *>(gap_base + ((gap_off[s] >> 8) & 0xFFu)) +
                        *reinterpret_cast<const float *>(gap_base + (gap_off[s] >> 16)));
            if constexpr (!kCount) {  // a pruned neighbour is not looked up
                if (lower[s] > B * 1.0001f) {
                    act[s] = false;
                    key[s] = 0ull;
                    h[s] = 0u;
                }
            }
        }
        // both sets' first slots in flight together
        u32x4 raw[kSets];
        load_slots2(map.table + h[0], map.table + h[1], raw[0], raw[1]);
#pragma unroll
        for (int s = 0; s < kSets; s++) {
            cnt[s] = 0;
            slab[s] = 0;
            if (act[s]) {
                u32x4 r = raw[s];
                uint32_t hh = h[s];
                for (uint32_t probe = 0; probe <= map.mask; probe++) {
                    const unsigned long long k = ((unsigned long long)r.y << 32) | r.x;
                    if (k == key[s]) {
                        cnt[s] = r.z;
                        slab[s] = r.w;
                        break;
                    }
                    if (k == kEmptyKey) break;
                    hh = (hh + 1) & map.mask;
                    r = load_slot(map.table + hh);
                }
            }
        }
        LOM_STAMP(2);  // slots probed
        uint32_t probed = 0;  // (lom_profile_match's tally launch: the slots this query asked for)
        if constexpr (!kCount && !kChained)
            if (out_stat) probed = row_sum((act[0] ? 1u : 0u) + (act[1] ? 1u : 0u));
        if constexpr (kCount) B = temporal_bound();
        uint32_t w_d, w_c, best_pi0, best_c, n_cand = 0, n_occ = 0, T = 0;  // T: points actually read
        float best;
        if constexpr (kCount) {
            // the reference's counts: occupied voxels (<= 27) above bit 26, stored points (<= 27 K, K < 2^16) below: one
            // row sum for both sets
            uint32_t mine = 0;
#pragma unroll
            for (int s = 0; s < kSets; s++) mine += cnt[s] | ((cnt[s] ? 1u : 0u) << 26);
            const uint32_t tot = row_sum(mine);
            n_cand = tot & ((1u << 26) - 1u);
            n_occ = tot >> 26;
        }
        const float bound = B * 1.0001f;
#pragma unroll
        for (int s = 0; s < kSets; s++) {
            // a neighbour voxel whose nearest face is provably farther than the bound is not read
            scan_cnt[s] = (lower[s] > bound) ? 0u : cnt[s];
        }
        // ---- group-wide prefix over the scanned neighbours in scan order ----
        // best starts at max_sq: "d2 < best" then implies voxel_grid.h:186's d2 < max_sq, and NaN never wins
        best = P.max_sq;
        best_c = 0xFFFFFFFFu;
        best_pi0 = 0;
        {
            // chunks of up to four consecutive points of one voxel: nch chunks per scanned voxel
            uint32_t nch[kSets], read = 0;
#pragma unroll
            for (int s = 0; s < kSets; s++) {
                nch[s] = (scan_cnt[s] + ((1u << kRowsLog2) - 1u)) >> kRowsLog2;
                read += scan_cnt[s];
            }
            uint32_t Tc = 0;  // chunks of this query
            if (map.K <= 16380u) {
                // both sets' chunk counts in one register (16 voxels x K / 4 < 2^16 each): ONE row scan, one broadcast
#pragma unroll
                for (int s = 0; s < kSets; s += 2) {
                    const uint32_t inc = row_scan_inclusive(nch[s] | (nch[s + 1] << 16));
                    const uint32_t last = row_last(inc);
                    const uint32_t tot_a = last & 0xFFFFu, inc_a = Tc + (inc & 0xFFFFu), inc_b = Tc + tot_a + (inc >> 16);
                    const uint32_t ex_a = (inc_a - nch[s]) << kRowsLog2, ex_b = (inc_b - nch[s + 1]) << kRowsLog2;
                    const int b_a = gl + s * G, b_b = b_a + G;
                    s_pb[grp][b_a] = make_uint4(slab[s] * map.K - ex_a, scan_cnt[s] + ex_a, (b_a < 27) ? inc_a : 0xFFFFFFFFu, 0u);
                    s_pb[grp][b_b] =
                        make_uint4(slab[s + 1] * map.K - ex_b, scan_cnt[s + 1] + ex_b, (b_b < 27) ? inc_b : 0xFFFFFFFFu, 0u);
                    Tc += tot_a + (last >> 16);
                }
            } else {
#pragma unroll
                for (int s = 0; s < kSets; s++) {
                    const uint32_t inc = row_scan_inclusive(nch[s]);
                    const int b = gl + s * G;
                    const uint32_t ex = (Tc + inc - nch[s]) << kRowsLog2;
                    s_pb[grp][b] = make_uint4(slab[s] * map.K - ex, scan_cnt[s] + ex, (b < 27) ? Tc + inc : 0xFFFFFFFFu, 0u);
                    Tc += row_last(inc);
                }
            }
            // points actually read (after the exact pruning): one more row sum (an LDS atomic per lane instead cost the
            // kernel's tail 0.3 us: sixteen lanes on one word)
            if constexpr (!kChained) T += row_sum(read);  // (only lom_profile_match reads it: not computed inside an align)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            LOM_STAMP(3);  // prefix in LDS
            // Binary search of a chunk's voxel: the first two of its five levels compare with three values read once
            // per query, the last reads the entry and its successor together -- three dependent LDS round trips
            // per chunk of four candidates.
            const uint4 *pb = s_pb[grp];
            const uint32_t p7 = pb[7].z, p15 = pb[15].z, p23 = pb[23].z;
            // lane l takes chunks l, l + 16, ... of the flattened sequence; a chunk's four points are consecutive rows:
            // one address, four 12-byte loads in flight, compared in ascending order (strict minimum per lane: first
            // wins).  Rows of a chunk beyond the voxel's count are read (they exist: the slab, the next one, or the
            // padding behind the last) and not compared.
            for (uint32_t ch = gl; ch < Tc; ch += G) {
                // smallest b with prefix[b] > ch
                uint32_t b = (p15 <= ch) ? 16u : 0u;
                b += ((b ? p23 : p7) <= ch) ? 8u : 0u;
                b += (pb[b + 3].z <= ch) ? 4u : 0u;
                b += (pb[b + 1].z <= ch) ? 2u : 0u;
                const uint4 e = pb[b];
                const uint2 nx = *reinterpret_cast<const uint2 *>(&pb[b + 1]);
                const bool up = e.z <= ch;
                const uint32_t c0 = ch << 2;
                const uint32_t pi0 = (up ? nx.x : e.x) + c0;   // first row of the chunk
                const uint32_t nv = (up ? nx.y : e.y) - c0;    // rows of the voxel from there on (>= 1)
                f32x3 pt[4];
                load_chunk48(map.pts + (size_t)pi0 * 3, pt);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const f32x3 a = pt[u];
                    const float dx = qx - a.x, dy = qy - a.y, dz = qz - a.z;
                    const float d2 = dx * dx + (dy * dy + dz * dz);  // voxel_grid.h:184 f32 squaredNorm
                    if ((u == 0 || nv > (uint32_t)u) && d2 < best) {  // :186-187 strict
                        best = d2;
                        best_c = c0 + (uint32_t)u;
                        best_pi0 = pi0;
                    }
                }
            }
        }
        LOM_STAMP(4);  // candidates scanned
        // lexicographic min over the group; d2 >= 0 so its bit pattern orders like the value
        // (as two 32-bit row minima -- the distance bits, then the ordinal among the lanes that hold that distance --:
        // half the instructions of four 64-bit compare-and-select steps)
        w_d = row_min32(__float_as_uint(best));
        w_c = row_min32(__float_as_uint(best) == w_d ? best_c : 0xFFFFFFFFu);
        const bool valid = w_c != 0xFFFFFFFFu;
        LOM_STAMP(5);  // group minimum known
        // the lane that scanned the winner reads its point again together with the normal (two loads, one round
        // trip: keeping the point in registers through the candidate loop cost three selects per candidate)
        if (valid ? (best_c == w_c) : (gl == 0)) {
            int32_t idx = -1;
            const size_t pi = (size_t)best_pi0 + (best_c & 3u);
            const uint32_t mark = valid ? (kRecValid | (uint32_t)pi) : 0u;
            if (valid) idx = (int32_t)pi;
            // Outer iterations >= 2: most queries find the winner they had (the pose moves by millimetres).  Such a query's
            // record -- point, normal, mark -- is what it would write again: neither the winner's point and normal are
            // fetched (the last of the query's dependent round trips) nor anything stored.
            bool same = false;
            if constexpr (kPrev) same = __float_as_uint(pv.w) == mark && pi < (size_t)kRecValid;
            if (!same) {
                f32x3 wp = {0.f, 0.f, 0.f}, wn = {0.f, 0.f, 0.f};
                if (valid) load_points2(map.pts + pi * 3, map.nrm + pi * 3, wp, wn);  // voxel_grid.h:197-198
                float4 *rec = reinterpret_cast<float4 *>(out_rec + q);
                if constexpr (kPrev)
                    reinterpret_cast<float *>(rec)[3] = wn.x;  // the source point is there since the first search of this scan
                else
                    rec[0] = make_float4(sp.x, sp.y, sp.z, wn.x);
                rec[1] = make_float4(wp.x, wp.y, wp.z, __uint_as_float(mark));
                rec[2] = make_float4(wn.y, wn.z, 0.f, 0.f);
            }
            if constexpr (!kChained) out_idx[q] = idx;  // (only lom_match_find_pairs reads it)
            if (!kChained && out_stat) {
                QStat st;
                st.sq_dist = valid ? best : 0.f;
                // with the counts: the reference algorithm's; without: what this launch itself read (rows) and looked up
                // (slots) for the query -- lom_profile_match's "requested bytes"; lom_match_find_pairs reports zeros then
                st.n_cand = kCount ? n_cand : T;
                st.n_occ = kCount ? n_occ : probed;
                st.pad = 0;
                out_stat[q] = st;
            }
            atomicAdd(&s_cnt[grp][0], valid ? 1u : 0u);
            if constexpr (kCount) {
                atomicAdd(&s_cnt[grp][1], n_cand);
                atomicAdd(&s_cnt[grp][2], n_occ);
            }
            if constexpr (!kChained) atomicAdd(&s_cnt[grp][3], T);  // candidates actually read (after the exact pruning)
        }
        LOM_STAMP(6);  // winner's normal loaded, record stored
        stamper.first_done();
        // the LDS tables are rewritten next iteration: all reads above are complete for this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    stamper.flush(stamps);
    // per-block counters, summed in fixed order by k_finish (no same-address atomics:
    // one word saturates at ~88 atomics/us, MI355X_MICROARCH.md "dequeue")
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t v = 0;
        for (int g = 0; g < kGroups; g++) v += s_cnt[g][threadIdx.x];
        block_counters[blockIdx.x * 4 + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------------
// Evaluation: residual + Jacobian + robust weight + reduction.  One lane per
// source point, grid-stride; 28 f64 accumulators per lane; each workgroup
// publishes ONE 256-byte record: [0..27] its sums, [28..30] its slice of the
// counters k_match left per workgroup, [31] the evaluation's sequence number.
//
//   k_eval         one evaluation per launch; records stay in HBM (multi-GPU path:
//                  k_sum_records folds them for the RCCL all-gather).
//   k_eval_server  host-driven path, one GPU per rank.  Launched once per outer iteration behind
//                  k_match, it evaluates at the launch pose, then stays resident and
//                  serves the LM iterations: the host writes {seq, op, pose} into
//                  pinned host memory, every workgroup polls that word, evaluates,
//                  and stores its record straight into coherent pinned host memory
//                  (payload, system-scope release, sequence word).  The host polls the
//                  <= 64 sequence words and adds the records in workgroup order --
//                  bitwise reproducible; per LM iteration there is no kernel launch,
//                  no inter-workgroup hand-off, no copy and no stream synchronisation.
//                  The first point of every lane stays in registers across
//                  evaluations.  Workgroups never wait on each other, and every spin
//                  is bounded by a wall-clock timeout (s_memrealtime), so the grid
//                  always drains; the host relaunches if a server timed out.
// ---------------------------------------------------------------------------
constexpr int kRecWords = 32;    // doubles per record
constexpr int kAccStride = kEvalThreads + 16;  // LDS row stride (doubles): rows k, k+1 land on disjoint banks

struct EvalCmd {  // pinned host memory, written by the host only
    unsigned long long seq;  // increases with every command
    unsigned int op;         // kCmdEval / kCmdStop
    unsigned int pad;
    double q[4];
    double t[3];
};
constexpr unsigned int kCmdEval = 1, kCmdStop = 2;
constexpr int kPublishPlain = 0, kPublishHost = 1, kPublishDevice = 2;
constexpr int kPairsAhead = 5;  // (k_match, k_lm) pairs enqueued before the host looks at a report
constexpr uint32_t kMaxLmBlocks = 64;    // workgroups of k_lm (one lane of a wave watches each record)
constexpr uint32_t kMaxLmBlocksBig = 128;  // ... of its variant for large clouds; also the size of an exchange set

// The f64 residual / Jacobian arithmetic below contracts a * b + c to one FMA (the library is built with
// -ffp-contract=off for the f32 index and distance expressions of the search, which must round like the reference's
// x86 build; these f64 sums are compared with the oracle's to 1e-12 of their scale, not bit for bit, and the order of
// the additions across points differs from any CPU's anyway): a third fewer instructions per point.
#pragma clang fp contract(fast)
struct PointTerms {
    double J[6], r;
};
// cloud_matcher.cpp:48-98 for one correspondence: residual and 1x6 tangent Jacobian
__device__ __forceinline__ void point_terms(const float4 ra, const float4 rb, const float4 rc, const double q0,
                                            const double q1, const double q2, const double q3, const double t0,
                                            const double t1, const double t2, PointTerms &T)
{
    const double p[3] = {(double)ra.x, (double)ra.y, (double)ra.z};
    const double o[3] = {(double)rb.x, (double)rb.y, (double)rb.z};
    const double nn[3] = {(double)ra.w, (double)rc.x, (double)rc.y};
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
    const double e0 = rp0 + t0 - o[0], e1 = rp1 + t1 - o[1], e2 = rp2 + t2 - o[2];
    T.r = e0 * nn[0] + (e1 * nn[1] + e2 * nn[2]);
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
    T.J[0] = ja[0] * -q1 + ja[1] * q0 + ja[2] * -q3 + ja[3] * q2;
    T.J[1] = ja[0] * -q2 + ja[1] * q3 + ja[2] * q0 + ja[3] * -q1;
    T.J[2] = ja[0] * -q3 + ja[1] * -q2 + ja[2] * q1 + ja[3] * q0;
    T.J[3] = nn[0];  // cloud_matcher.cpp:96-98
    T.J[4] = nn[1];
    T.J[5] = nn[2];
}
// ceres::HuberLoss(0.15) (cloud_matcher.cpp:134); rho'' <= 0 -> plain IRLS weight rho'; then the 28 sums
__device__ __forceinline__ void point_accumulate(const PointTerms &T, double acc[28])
{
    const double r = T.r, s = r * r;
    double rho0 = s, w = 1.0;
    if (s > 0.15 * 0.15) {
        const double rr = sqrt(s);
        rho0 = 2.0 * 0.15 * rr - 0.15 * 0.15;
        w = fmax(DBL_MIN, 0.15 / rr);
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double wa = w * T.J[a];
#pragma unroll
        for (int b = a; b < 6; b++) acc[k++] += wa * T.J[b];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += w * T.J[a] * r;
    acc[27] += 0.5 * rho0;
}
// cloud_matcher.cpp:48-102 for one correspondence, accumulated into the 28 sums
__device__ __forceinline__ void accumulate_point(const float4 ra, const float4 rb, const float4 rc,
                                                 const double q0, const double q1, const double q2, const double q3,
                                                 const double t0, const double t1, const double t2, double acc[28])
{
    PointTerms T;
    point_terms(ra, rb, rc, q0, q1, q2, q3, t0, t1, t2, T);
    point_accumulate(T, acc);
}
#pragma clang fp contract(off)

// Workgroup reduction of the 28 per-lane sums through LDS in a fixed order, plus the
// workgroup's slice of k_match's counters; one wave then writes the 256-byte record.
// s_acc: dynamic LDS, 28 rows of kAccStride doubles.
__device__ __forceinline__ void reduce_and_publish(const double acc[28], double *s_acc, unsigned long long *s_cnt,
                                                   const uint32_t *__restrict__ block_counters,
                                                   uint32_t n_match_blocks, double *out_rec,
                                                   unsigned long long seq, int mode)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 28; k++) s_acc[k * kAccStride + tid] = acc[k];
    if (wave == 0 && n_match_blocks) {
        const uint32_t chunk = (n_match_blocks + gridDim.x - 1) / gridDim.x;
        const uint32_t lo = blockIdx.x * chunk;
        const uint32_t hi = min(lo + chunk, n_match_blocks);
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        for (uint32_t b = lo + lane; b < hi; b += 64) {
            const uint4 r = *reinterpret_cast<const uint4 *>(block_counters + (size_t)b * 4);
            c0 += r.x;
            c1 += r.y;
            c2 += r.z;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            c0 += __shfl_xor(c0, d, 64);
            c1 += __shfl_xor(c1, d, 64);
            c2 += __shfl_xor(c2, d, 64);
        }
        if (lane == 0) {
            s_cnt[0] = c0;
            s_cnt[1] = c1;
            s_cnt[2] = c2;
        }
    }
    __syncthreads();
    // thread (k = tid / 16, j = tid % 16) adds row k's elements j, j+16, ... in order
    const int k = tid >> 4, j = tid & 15;
    double v = 0.0;
    if (k < 28) {
        const double *row = s_acc + k * kAccStride + j;
#pragma unroll 8
        for (int i = 0; i < kEvalThreads / 16; i++) v += row[i * 16];
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 16);
    __syncthreads();  // every read of s_acc is done: its first words become the staging row
    if (j == 0 && k < 28) s_acc[k] = v;
    __syncthreads();
    if (tid < 32) {  // one wave writes the whole 256-byte record
        double o = 0.0;
        if (tid < 28)
            o = s_acc[tid];
        else if (tid < 31)
            o = n_match_blocks ? (double)s_cnt[tid - 28] : 0.0;
        double *dst = out_rec + (size_t)blockIdx.x * kRecWords;
        if (mode == kPublishHost) {
            // payload as system-scope (write-through) stores, wait until they have left the wave,
            // then the sequence word: the same order a system-scope release gives, without its
            // L2 write-back pass (nothing this wave wrote is cached)
            if (tid < 31) __hip_atomic_store(dst + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 31)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + 31), seq, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (mode == kPublishDevice) {
            // to the other workgroups of this launch (any XCD): every store of the record
            // agent-coherent and drained before the sequence word; the readers use
            // agent-coherent loads for both
            if (tid < 31) __hip_atomic_store(dst + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 31)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + 31), seq, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        } else if (tid < 31) {
            dst[tid] = o;
        }
    }
    __syncthreads();  // s_acc / s_cnt may be rewritten by the next evaluation
}

__global__ __launch_bounds__(kEvalThreads) void k_eval(const MatchRec *__restrict__ rec, uint32_t n, EvalArgs E,
                                                       const uint32_t *__restrict__ block_counters,
                                                       uint32_t n_match_blocks, double *out_rec,
                                                       unsigned long long seq)
{
    extern __shared__ __attribute__((aligned(16))) double s_acc[];
    __shared__ unsigned long long s_cnt[3];
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rec + i);
        const float4 ra = r4[0], rb = r4[1], rc = r4[2];
        if (rb.w != 0.f) accumulate_point(ra, rb, rc, E.q[0], E.q[1], E.q[2], E.q[3], E.t[0], E.t[1], E.t[2], acc);
    }
    reduce_and_publish(acc, s_acc, s_cnt, block_counters, n_match_blocks, out_rec, seq, kPublishPlain);
}

__global__ __launch_bounds__(kEvalThreads) void k_eval_server(const MatchRec *__restrict__ rec, uint32_t n,
                                                              EvalArgs E0, const uint32_t *__restrict__ block_counters,
                                                              uint32_t n_match_blocks, double *out_rec,
                                                              unsigned long long seq0, const EvalCmd *cmd,
                                                              unsigned long long cmd_seen,
                                                              unsigned long long timeout_ticks)
{
    extern __shared__ __attribute__((aligned(16))) double s_acc[];
    __shared__ unsigned long long s_cnt[3];
    __shared__ EvalCmd s_cmd;
    const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    // this lane's first point stays in registers for every evaluation of the outer iteration
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    if (first < n) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rec + first);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2];
    }
    double q0 = E0.q[0], q1 = E0.q[1], q2 = E0.q[2], q3 = E0.q[3], t0 = E0.t[0], t1 = E0.t[1], t2 = E0.t[2];
    unsigned long long seq = seq0;
    uint32_t counters_from = n_match_blocks;  // counters are folded by the first evaluation only
    for (;;) {
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0.0;
        if (rb.w != 0.f) accumulate_point(ra, rb, rc, q0, q1, q2, q3, t0, t1, t2, acc);
        for (uint32_t i = first + step; i < n; i += step) {
            const float4 *r4 = reinterpret_cast<const float4 *>(rec + i);
            const float4 xa = r4[0], xb = r4[1], xc = r4[2];
            if (xb.w != 0.f) accumulate_point(xa, xb, xc, q0, q1, q2, q3, t0, t1, t2, acc);
        }
        reduce_and_publish(acc, s_acc, s_cnt, block_counters, counters_from, out_rec, seq, kPublishHost);
        counters_from = 0;
        // wait for the next command from the host (bounded: the grid always drains).  The first
        // wave reads the 72-byte command with ONE instruction per poll (lanes 0..8, one word each,
        // relaxed system-scope loads: no cache invalidate per poll), then once more after the
        // sequence word changed -- the host wrote the payload before the sequence word.
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            unsigned long long *words = reinterpret_cast<unsigned long long *>(const_cast<EvalCmd *>(cmd));
            unsigned long long *my = words + (lane < 9 ? lane : 0);
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool timed_out = false;
            for (;;) {
                const unsigned long long w = __hip_atomic_load(my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (__shfl(w, 0, 64) != cmd_seen) break;
                if (__builtin_amdgcn_s_memrealtime() - t_start > timeout_ticks) {
                    timed_out = true;  // host went away: leave
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            const unsigned long long w = __hip_atomic_load(my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane < 9) reinterpret_cast<unsigned long long *>(&s_cmd)[lane] = w;
            if (timed_out && lane == 0) s_cmd.op = kCmdStop;
        }
        __syncthreads();
        if (s_cmd.op != kCmdEval) return;  // uniform over the workgroup
        q0 = s_cmd.q[0];
        q1 = s_cmd.q[1];
        q2 = s_cmd.q[2];
        q3 = s_cmd.q[3];
        t0 = s_cmd.t[0];
        t1 = s_cmd.t[1];
        t2 = s_cmd.t[2];
        seq = s_cmd.seq;
        cmd_seen = s_cmd.seq;
        __syncthreads();  // s_cmd is rewritten by thread 0 in the next round
    }
}

// Exchange word of k_lm: a value and a check word = sequence number XOR the value's bits.  A reader
// accepts the pair only when check ^ bits == the sequence number it waits for, so the two 8-byte
// words need no ordering between them and no separate "record complete" flag: publishing is one
// memory round trip and reading is one more.
struct __attribute__((aligned(16))) XWord {
    unsigned long long bits, check;
};

// The pair travels as ONE 16-byte agent-coherent access each way (sc1: past this XCD's L2).  Nothing relies on the
// access being indivisible -- a reader that catches half a pair sees check ^ bits != seq and polls again -- it only
// halves the memory instructions of the exchange (two 8-byte atomics per word each way in round 2).
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void xword_store(XWord *dst, double v, unsigned long long seq)
{
    u64x2 w;
    w.x = (unsigned long long)__double_as_longlong(v);
    w.y = seq ^ w.x;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(w) : "memory");
}
__device__ __forceinline__ void xword_load_issue(const XWord *src, u64x2 &r)  // result valid after xword_load_wait
{
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r) : "v"(src) : "memory");
}
template <int kN>
__device__ __forceinline__ void xword_load_wait(u64x2 (&r)[kN])
{
    static_assert(kN == 4 || kN == 8, "loads in flight per lane");
    if constexpr (kN == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
}

// a' + b' after v_permlane{32,16}_swap(a, b): lanes of the lower half (of the wave / of each pair of rows) end with
// a[lane] + a[partner], lanes of the upper half with b[partner] + b[lane] -- two values folded by one addition
// (lane mapping verified on the device: tools/microbench/permlane_swap.hip)
template <int kWidth>
__device__ __forceinline__ double swap_add(double a, double b)
{
    unsigned int alo = (unsigned int)__double2loint(a), ahi = (unsigned int)__double2hiint(a);
    unsigned int blo = (unsigned int)__double2loint(b), bhi = (unsigned int)__double2hiint(b);
    if constexpr (kWidth == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
}

// k_lm's evaluation epilogue: workgroup reduction of the 28 per-lane sums through LDS in a fixed
// order, every row total published straight from the lane that holds it (plus the workgroup's
// slice of k_match's counters), then all workgroups' words gathered and added in workgroup order
// into s_tot[0..30] -- bitwise the same on every workgroup.  Only the first wave may read s_tot
// afterwards (no workgroup barrier behind the final sum); the caller's next __syncthreads()
// releases s_acc / s_part for the following evaluation.
//   s_acc: 32 doubles per wave;  s_part: kT doubles (kT = threads of the workgroup).
template <int kT, int kBlocks>
__device__ __forceinline__ void reduce_and_exchange(const double acc[28], double *s_acc, double *s_part,
                                                    const uint32_t *__restrict__ block_counters,
                                                    uint32_t n_match_blocks, XWord *set, uint32_t nb,
                                                    unsigned long long seq, unsigned long long timeout_ticks,
                                                    double *s_tot, int *s_failed, const int32_t *chain_error,
                                                    const uint4 pre, unsigned long long *dbg = nullptr)
{
#define RX_STAMP(k)                                                     \
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) {                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        dbg[k] = __builtin_amdgcn_s_memtime();                          \
    }
    RX_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    XWord *mine = set + (size_t)blockIdx.x * kRecWords;
    // Wave level, all in registers, as a reduce-scatter: v_permlane32_swap exchanges the upper half of one
    // register with the lower half of another, so ONE add folds two values at once -- the lower 32 lanes keep
    // value k, the upper 32 value k + 14 (28 -> 14 values per lane); v_permlane16_swap does the same between
    // the 16-lane rows (14 -> 7: row r now holds values k + 7 r); four DPP butterflies finish the 7 values
    // inside each row.  147 instructions instead of the 168 of a quad pre-sum plus an LDS pass over 28 x 128
    // doubles, and what goes through LDS is 28 doubles per wave.  The order of the additions is fixed.
    double s1[14];
#pragma unroll
    for (int k = 0; k < 14; k++) s1[k] = swap_add<32>(acc[k], acc[k + 14]);
    double s2[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        double v = swap_add<16>(s1[k], s1[k + 7]);
        v += dpp_f64<kDppXor1>(v);
        v += dpp_f64<kDppXor2>(v);
        v += dpp_f64<kDppHalfMirror>(v);
        v += dpp_f64<kDppMirror>(v);
        s2[k] = v;
    }
    if ((lane & 15) == 0) {  // the first lane of row r holds the wave's totals of values 7 r .. 7 r + 6
        double *dst = s_acc + wave * 32 + 7 * (lane >> 4);
#pragma unroll
        for (int k = 0; k < 7; k++) dst[k] = s2[k];
    }
    if (wave == kT / 64 - 1) {  // the workgroup's slice of k_match's counters (first evaluation of a launch only)
        // the slice of a workgroup is at most one block per lane when k_lm runs 28 workgroups or more: the caller
        // then loaded this lane's block with the kernel's start-up loads (`pre`); counts are exact in f64, and the
        // wave sum is the permlane-swap / DPP fold of the residual sums (36 LDS-crossbar shuffles in round 2)
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        if (n_match_blocks) {
            const uint32_t chunk = (n_match_blocks + gridDim.x - 1) / gridDim.x;
            if (chunk <= 64u) {
                uint4 r = pre;
                if constexpr (kT != 256) {  // (the 512-thread shapes have no registers to spare for the early load)
                    const uint32_t b = blockIdx.x * chunk + (uint32_t)lane;
                    r = make_uint4(0u, 0u, 0u, 0u);
                    if ((uint32_t)lane < chunk && b < n_match_blocks) r = *reinterpret_cast<const uint4 *>(block_counters + (size_t)b * 4);
                }
                d0 = (double)r.x;
                d1 = (double)r.y;
                d2 = (double)r.z;
            } else {
                const uint32_t lo = blockIdx.x * chunk;
                const uint32_t hi = min(lo + chunk, n_match_blocks);
                unsigned long long c0 = 0, c1 = 0, c2 = 0;
                for (uint32_t b = lo + lane; b < hi; b += 64) {
                    const uint4 r = *reinterpret_cast<const uint4 *>(block_counters + (size_t)b * 4);
                    c0 += r.x;
                    c1 += r.y;
                    c2 += r.z;
                }
                d0 = (double)c0;
                d1 = (double)c1;
                d2 = (double)c2;
            }
            // rows after the two swaps: 0 = d0, 1 = d2, 2 = d1, 3 = nothing; four butterflies finish each row
            double v = swap_add<16>(swap_add<32>(d0, d1), swap_add<32>(d2, 0.0));
            v += dpp_f64<kDppXor1>(v);
            v += dpp_f64<kDppXor2>(v);
            v += dpp_f64<kDppHalfMirror>(v);
            v += dpp_f64<kDppMirror>(v);
            d0 = v;
        }
        if (lane == 0 || lane == 16 || lane == 32) xword_store(mine + 28 + (lane == 0 ? 0 : (lane == 32 ? 1 : 2)), d0, seq);
    }
    __syncthreads();
    if (tid < 28) {  // the eight waves' totals, in wave order
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kT / 64; w++) v += s_acc[w * 32 + tid];
        RX_STAMP(1);
        xword_store(mine + tid, v, seq);
    }
    RX_STAMP(2);
    // gather: thread (g = tid / 32, k = tid % 32) takes word k of workgroups kPer g .. kPer g + kPer - 1
    {
        constexpr int kPer = kBlocks / (kT / 32);
        const int k = tid & 31, g = tid >> 5;
        unsigned long long vb[kPer];
        bool ok[kPer];
#pragma unroll
        for (int u = 0; u < kPer; u++) {
            vb[u] = 0;
            ok[u] = (k >= 31) || ((uint32_t)(g * kPer + u) >= nb);
        }
        // Let the words land before the first poll: a poll that comes too early is a wasted memory
        // round trip (and 52 workgroups x 512 lanes of them load the memory side).  Measured on C2:
        // no head start 0.1724 ms per align, s_sleep 8 / 12 / 16 / 20 / 24 -> 0.1668 / 0.1657 / 0.1645 /
        // 0.1650 / 0.1650; again after the round-2 reduction: 4 / 8 / 12 / 16 / 24 -> 0.1580 / 0.1562 / 0.1545 /
        // 0.1539 / 0.1548; round 3 (256-thread workgroups, two points per lane): 4 / 8 / 12 / 16 / 20 / 24 / 32 ->
        // 0.1349 / 0.1333 / 0.1315 / 0.1312 / 0.1325 / 0.1343 / 0.1366; at the end of round 3: 10 / 13 / 16 / 20 ->
        // 0.1355 / 0.1342 / 0.1332 / 0.1328.
        __builtin_amdgcn_s_sleep(16);
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        uint32_t polls = 0;
        // a patience shorter than the head start just slept (16 x 64 cycles, > 0.4 us = 40 ticks of 10 ns) cannot be
        // met whatever the first poll finds: the wait counts as timed out -- which is what makes a 1-tick patience a
        // deterministic way to force the give-up path (tests), not a race against the other workgroups' stores
        if (timeout_ticks < 40ull) {
            *s_failed = 1;
#pragma unroll
            for (int u = 0; u < kPer; u++) ok[u] = true;
        }
        const XWord *mine_src = set + (size_t)(g * kPer) * kRecWords + k;  // (a set holds kMaxLmBlocksBig records: in bounds)
        for (; timeout_ticks >= 40ull;) {
            u64x2 r[kPer];
#pragma unroll
            for (int u = 0; u < kPer; u++) xword_load_issue(mine_src + (size_t)u * kRecWords, r[u]);
            xword_load_wait(r);
            bool all = true;
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                if (!ok[u]) {
                    vb[u] = r[u].x;
                    ok[u] = (r[u].y ^ r[u].x) == seq;
                }
                all = all && ok[u];
            }
            if (all) break;
            if (__builtin_amdgcn_s_memrealtime() - t_start > timeout_ticks) {
                *s_failed = 1;  // some workgroup never published: give up (the grid drains)
                break;
            }
            // a wait that drags on: has a workgroup of this launch given up already?  Then the words this one waits
            // for will never come; it leaves now, not after its own patience.
            if ((++polls & 255u) == 0 && __hip_atomic_load(chain_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                *s_failed = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        double part = 0.0;
#pragma unroll
        for (int u = 0; u < kPer; u++) part += __longlong_as_double((long long)vb[u]);  // absent workgroups add +0.0
        s_part[tid] = part;
    }
    RX_STAMP(3);
    __syncthreads();
    if (tid < 64) {  // the first wave adds the kT / 32 partial sums in order and keeps the totals to itself
        if (tid < 31) {
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < kT / 32; g++) v += s_part[g * 32 + tid];
            s_tot[tid] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    RX_STAMP(4);
#undef RX_STAMP
}

// ---------------------------------------------------------------------------
// k_lm: one whole ceres::Solve (cloud_matcher.cpp:157-158) of the single-GPU align, resident
// on the GPU.  Launched behind k_match once per outer iteration.  Every evaluation the
// Levenberg-Marquardt policy (lm_core.hpp) asks for is done by the whole grid:
//   each workgroup reduces its points to one 256-byte record and publishes it in HBM
//   (agent-coherent stores, sequence word last; two record sets alternate), waits until
//   the records of all workgroups carry the evaluation's sequence number, and adds them
//   in workgroup order -- every workgroup holds the same totals bit for bit and runs the
//   same policy step (one lane), so no decision has to be broadcast and nothing returns to
//   the host between the evaluations of a solve.
// Workgroup 0 then writes the f32 pose back (:161-167), prepares the pose of the next
// k_match in AlignState, decides convergence (:169-172) and copies the state to the report in
// pinned host memory.  Every wait is bounded (s_memrealtime); a workgroup that gives up sets
// the error flags and leaves, the others follow.
// ---------------------------------------------------------------------------
// ---- ranks of one node: the ranks' totals exchanged by the GPUs themselves -------------------
// Every rank owns a small buffer in its HBM: [4 sets][kP2pMaxRanks][32] exchange words.  Inside one
// launch the sets alternate with the sequence number (the dependency chain of a solve keeps a rank at
// most one evaluation ahead of its peers); consecutive launches alternate between the set pairs
// {0,1} and {2,3}, so the first publish of the next k_lm can never overwrite a slot a lagging peer
// still polls for the previous kernel's last evaluation (the kernels of different ranks are not
// ordered against each other).  Rank r's
// workgroup 0 stores its 32 rank totals into slot r of EVERY rank's buffer (its own directly, the
// peers' through their IPC mappings: xGMI), system-coherent stores, same {bits, seq ^ bits} words as
// inside a GPU.  Every workgroup of every rank then reads its own GPU's buffer and adds the ranks'
// words in rank order: identical bits on all workgroups of all ranks, no host in the loop.
// Behind the four sets every buffer holds one ABORT word per rank: a rank whose kernel gives up (its workgroups not
// all resident, a peer that never published) stores the number of the align it abandons -- the same number on every
// rank -- into its word in EVERY rank's buffer.  A kernel waiting for that rank's totals looks at the abort words
// whenever a wait drags on and leaves at once, instead of after its own (ten times longer) patience: without that
// word the ranks reached the host-side agreement up to 100 s apart (round 2's three-rank failure).
constexpr size_t kP2pExchangeWords = (size_t)4 * kP2pMaxRanks * kRecWords;  // XWords before the abort words
constexpr size_t kP2pBufferBytes = kP2pExchangeWords * sizeof(XWord) + kP2pMaxRanks * sizeof(unsigned long long);
struct P2pArgs {
    XWord *peer[kP2pMaxRanks];  // peer[r]: rank r's buffer as seen from this GPU (peer[rank] = local)
    int rank, nranks;
    int set_base;  // 0 or 2: consecutive launches use disjoint pairs of exchange sets (see global_exchange)
    unsigned long long epoch;  // number of this device-to-device align (>= 1; ~0: the attach self-test)
};

__device__ __forceinline__ unsigned long long *p2p_abort_words(XWord *buffer)
{
    return reinterpret_cast<unsigned long long *>(buffer + kP2pExchangeWords);
}

// this rank abandons align `epoch`: tell every rank (own buffer included)
__device__ __forceinline__ void p2p_publish_abort(const P2pArgs &A)
{
    for (int r = 0; r < A.nranks; r++)
        __hip_atomic_store(p2p_abort_words(A.peer[r]) + A.rank, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void global_exchange(const P2pArgs &A, double *s_tot, unsigned long long seq,
                                                unsigned long long timeout_ticks, int *s_failed, bool publisher,
                                                int lane)
{
    const size_t set_off = (size_t)((unsigned)A.set_base + (unsigned)(seq & 1)) * kP2pMaxRanks * kRecWords;
    if (publisher && lane < kRecWords) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(s_tot[lane]);
        for (int r = 0; r < A.nranks; r++) {
            XWord *dst = A.peer[r] + set_off + (size_t)A.rank * kRecWords + lane;
            __hip_atomic_store(&dst->bits, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dst->check, seq ^ b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // lane (h = lane / 32, k = lane % 32) reads word k of ranks h, h + 2, h + 4, h + 6
    const XWord *local = A.peer[A.rank] + set_off;
    const int k = lane & 31, h = lane >> 5;
    unsigned long long vb[4] = {0, 0, 0, 0};
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long *aborts = p2p_abort_words(A.peer[A.rank]);
    bool failed = false;
    uint32_t polls = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = h + 2 * u;
            if (r < A.nranks) {
                const XWord *w = local + (size_t)r * kRecWords + k;
                const unsigned long long bits = __hip_atomic_load(&w->bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long chk = __hip_atomic_load(&w->check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                vb[u] = bits;
                all = all && ((chk ^ bits) == seq);
            }
        }
        if (__ballot(!all) == 0ull) break;
        if (__builtin_amdgcn_s_memrealtime() - t_start > timeout_ticks) {
            failed = true;  // a rank never published: give up (every grid drains)
            break;
        }
        if ((++polls & 63u) == 0) {  // a wait that drags on: has a rank abandoned this align?
            unsigned long long ab = 0;
            if (lane < A.nranks) ab = __hip_atomic_load(aborts + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (__ballot(lane < A.nranks && ab == A.epoch) != 0ull) {
                failed = true;
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    // rank order: lanes < 32 hold the even ranks, their partners (lane + 32) the odd ones
    double total = 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const double mine = __longlong_as_double((long long)vb[u]);
        const double other = __shfl_xor(mine, 32, 64);
        total += (h == 0) ? mine : other;   // rank 2u     (absent ranks add +0.0)
        total += (h == 0) ? other : mine;   // rank 2u + 1
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < kRecWords) s_tot[lane] = total;
    if (failed && lane == 0) *s_failed = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lom_comm_attach_p2p's self-test: `rounds` exchanges of known values between all ranks
__global__ __launch_bounds__(64) void k_p2p_selftest(P2pArgs A, unsigned long long seq_base, int rounds,
                                                     unsigned long long timeout_ticks, uint32_t *result)
{
    __shared__ double s_tot[kRecWords];
    __shared__ int s_failed;
    const int lane = threadIdx.x;
    if (lane == 0) s_failed = 0;
    __syncthreads();
    uint32_t bad = 0;
    for (int i = 0; i < rounds && !s_failed; i++) {
        if (lane < kRecWords) s_tot[lane] = (double)(A.rank + 1) * 1000.0 + (double)i + 0.5 * (double)lane;
        __syncthreads();
        // the first round also absorbs the start-up skew between the ranks' processes
        global_exchange(A, s_tot, seq_base + 1 + (unsigned long long)i, i == 0 ? timeout_ticks * 100 : timeout_ticks,
                        &s_failed, true, lane);
        if (lane < kRecWords && !s_failed) {
            double want = 0.0;
            for (int r = 0; r < A.nranks; r++) want += (double)(r + 1) * 1000.0 + (double)i + 0.5 * (double)lane;
            if (s_tot[lane] != want) bad++;
        }
        __syncthreads();
    }
    for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d, 64);
    if (lane == 0) {
        result[0] = bad;
        result[1] = (uint32_t)s_failed;
    }
}

struct LmInit {
    float t[3], q[4];   // initial guess (cloud_matcher.cpp:107), used when `first`
    double prior_b[3];  // NormalPrior anchor = the guess's translation (:153)
    float max_sq;       // max_correspondence_distance^2 of the searches (:139, voxel_grid.h:215)
};

// kRegPts: this lane's first points (first, first + step, ...) stay in registers for every evaluation of the solve
template <int kRegPts>
__device__ __forceinline__ void accumulate_all(const MatchRec *__restrict__ rec, uint32_t n, uint32_t first,
                                               uint32_t step, const float4 (&ra)[kRegPts], const float4 (&rb)[kRegPts],
                                               const float4 (&rc)[kRegPts], const double *x, double acc[28])
{
    const double q0 = x[0], q1 = x[1], q2 = x[2], q3 = x[3], t0 = x[4], t1 = x[5], t2 = x[6];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.0;
    // The register points go through unconditionally and stage by stage -- residuals and Jacobians of all of them, then
    // their sums -- so that the scheduler interleaves the independent chains (a wave alone on its SIMD issues a dependent
    // instruction every ~9 cycles, independent ones every ~5).  A lane without a match, or beyond the cloud, holds a zero
    // normal: every term it adds is exactly zero.
    PointTerms T[kRegPts];
#pragma unroll
    for (int p = 0; p < kRegPts; p++) point_terms(ra[p], rb[p], rc[p], q0, q1, q2, q3, t0, t1, t2, T[p]);
#pragma unroll
    for (int p = 0; p < kRegPts; p++) point_accumulate(T[p], acc);
    for (uint32_t i = first + (uint32_t)kRegPts * step; i < n; i += step) {
        const float4 *r4 = reinterpret_cast<const float4 *>(rec + i);
        const float4 xa = r4[0], xb = r4[1], xc = r4[2];
        if (xb.w != 0.f) accumulate_point(xa, xb, xc, q0, q1, q2, q3, t0, t1, t2, acc);
    }
}

// kPolicyTwice (LOM_DEBUG_LM_TWICE=1 at create, a measurement aid): the first wave runs every policy step twice -- the
// first time on state that is put back afterwards -- and the phase stamps time the second run: the same instructions
// on the same data, with the step's code already in the instruction cache.
template <int kT, int kBlocks = (int)kMaxLmBlocks, int kRegPts = 1, bool kPolicyTwice = false>
__global__ __launch_bounds__(kT) void k_lm(const MatchRec *__restrict__ rec, uint32_t n, AlignState *state,
                                                     LmInit init, int first_outer,
                                                     const uint32_t *__restrict__ block_counters,
                                                     uint32_t n_match_blocks, XWord *xrec,
                                                     unsigned long long seq_base, AlignReport *report,
                                                     unsigned long long report_seq,
                                                     unsigned long long timeout_ticks,
                                                     unsigned long long *dbg_stamps, P2pArgs px,
                                                     double *dbg_trace, int test_give_up)
{
    __shared__ double s_acc[(kT / 64) * 32];  // the waves' totals of one evaluation
    __shared__ double s_tot[kRecWords];
    __shared__ double s_part[kT];
    __shared__ double s_x[7];
    LmWave W;  // the solve's state: per-row part in the registers of the first wave, the rest in LDS (lm_wave.hpp)
    // the solve's uniform state: in every lane's registers in the 256-thread shapes, one copy in LDS in the 512-thread ones
    constexpr bool kRegState = kT == 256;
    __shared__ LmShared s_lm;
    LmShared r_lm;
    __shared__ int s_action, s_failed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = gridDim.x;
    const uint32_t first = blockIdx.x * blockDim.x + tid, step = nb * blockDim.x;
    // start-up loads issued together (one memory round trip, not three): this lane's first point --
    // it stays in registers for every evaluation of the solve --, the pose, the chain's stop flags
    // (the records' loads are ISSUED here and waited for behind the other start-up loads: left to the compiler, the second
    // register point's loads were scheduled behind the first one's wait -- two round trips where one will do; a lane
    // beyond the cloud reads record 0 and forgets it)
    typedef float RecQuarter __attribute__((ext_vector_type(4)));
    RecQuarter raw_a[kRegPts], raw_b[kRegPts], raw_c[kRegPts];
    float4 ra[kRegPts], rb[kRegPts], rc[kRegPts];
    bool have[kRegPts];
#pragma unroll
    for (int p = 0; p < kRegPts; p++) {
        const uint32_t i = first + (uint32_t)p * step;
        have[p] = i < n;
        const MatchRec *at = rec + (have[p] ? i : 0u);
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32"
                     : "=&v"(raw_a[p]), "=&v"(raw_b[p]), "=&v"(raw_c[p])
                     : "v"(at)
                     : "memory");
    }
    // ... and, in the last wave, this lane's block of k_match's counters (reduce_and_exchange folds them)
    // (both without a divergent branch around the load -- every lane loads, from a clamped address, and picks afterwards --:
    // the compiler waits for the loads of a divergent region where the region ends, which made these two more round
    // trips in a row behind the records')
    uint4 cnt_pre = make_uint4(0u, 0u, 0u, 0u);
    RecQuarter cnt_raw = {0.f, 0.f, 0.f, 0.f};
    bool cnt_want = false;
    if constexpr (kT == 256) {
        const uint32_t chunk = (n_match_blocks + nb - 1) / nb;
        const uint32_t b = blockIdx.x * chunk + (uint32_t)lane;
        cnt_want = wave == kT / 64 - 1 && n_match_blocks && chunk <= 64u && (uint32_t)lane < chunk && b < n_match_blocks;
        const uint32_t *at = block_counters + (size_t)(cnt_want ? b : 0u) * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(cnt_raw) : "v"(at) : "memory");  // (issued, like the records)
    }
    float x0;
    {
        const int k = tid < 7 ? tid : 0;
        if (first_outer) {  // (uniform)
            x0 = k < 4 ? init.q[k] : init.t[k - 4];
        } else {
            const float *from = k < 4 ? &state->pose_q[k] : &state->pose_t[k - 4];
            x0 = *from;
        }
    }
    // the previous outer iteration's tallies, read now (scalar loads, with everything else that starts the kernel) for
    // workgroup 0's write-back at the very end: read there they were one more memory round trip on the critical path
    typedef const __attribute__((address_space(4))) AlignState *ConstState;
    struct {
        int32_t outer_done, lm_iterations, evaluations;
        double valid_total, cand_total, occ_total, queries_total;
    } prev = {0, 0, 0, 0.0, 0.0, 0.0, 0.0};
    if (!first_outer) {
        ConstState cs = (ConstState)state;
        prev.outer_done = cs->outer_done;
        prev.lm_iterations = cs->lm_iterations;
        prev.evaluations = cs->evaluations;
        prev.valid_total = cs->valid_total;
        prev.cand_total = cs->cand_total;
        prev.occ_total = cs->occ_total;
        prev.queries_total = cs->queries_total;
        if (cs->finished | cs->error) return;  // chained launch after the end
    }
    // ... and parked in LDS until then: eleven scalar registers less to carry (or spill) through the solve
    __shared__ double s_prev[4];
    __shared__ int32_t s_prev_i[3];
    if (tid == 0) {
        s_prev[0] = prev.valid_total;
        s_prev[1] = prev.cand_total;
        s_prev[2] = prev.occ_total;
        s_prev[3] = prev.queries_total;
        s_prev_i[0] = prev.outer_done;
        s_prev_i[1] = prev.lm_iterations;
        s_prev_i[2] = prev.evaluations;
    }
    // the records are needed from here on
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cnt_raw)::"memory");
    if (cnt_want)
        cnt_pre = make_uint4(__float_as_uint(cnt_raw.x), __float_as_uint(cnt_raw.y), __float_as_uint(cnt_raw.z), __float_as_uint(cnt_raw.w));
#pragma unroll
    for (int p = 0; p < kRegPts; p++) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_a[p]), "+v"(raw_b[p]), "+v"(raw_c[p])::"memory");
        ra[p] = have[p] ? make_float4(raw_a[p].x, raw_a[p].y, raw_a[p].z, raw_a[p].w) : make_float4(0.f, 0.f, 0.f, 0.f);
        rb[p] = have[p] ? make_float4(raw_b[p].x, raw_b[p].y, raw_b[p].z, raw_b[p].w) : make_float4(0.f, 0.f, 0.f, 0.f);
        rc[p] = have[p] ? make_float4(raw_c[p].x, raw_c[p].y, raw_c[p].z, raw_c[p].w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 7) s_x[tid] = (double)x0;  // cloud_matcher.cpp:122-131
    if (tid == 0) s_failed = test_give_up;  // LOM_OPT_TEST_GIVE_UP_AT_OUTER: this launch behaves as if its waits had timed out
    __syncthreads();
    unsigned long long seq = seq_base;
    uint32_t counters_from = n_match_blocks;  // k_match's counters are folded by the first evaluation only
    double counters[4] = {0.0, 0.0, 0.0, 0.0};  // valid, cand, occ of the last k_match; queries (all ranks)
    int action = LM_EVAL;
    // LOM_DEBUG_LM: shader-clock stamps of workgroup 0's first lane in the first k_lm of the align
    // (0 start, 1 accumulated, 3 totals known, 4 policy done)
#define LM_STAMP(k)                                                                            \
    if (dbg_stamps && first_outer && blockIdx.x == 0 && tid == 0 && ev < 5) {                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                            \
        dbg_stamps[ev * 5 + (k)] = __builtin_amdgcn_s_memtime();                               \
    }
    for (int ev = 0; action == LM_EVAL; ev++) {
        double acc[28];
        LM_STAMP(0);
        accumulate_all<kRegPts>(rec, n, first, step, ra, rb, rc, s_x, acc);
        LM_STAMP(1);
        seq++;
        XWord *set = xrec + (size_t)(seq & 1) * kMaxLmBlocksBig * kRecWords;
        reduce_and_exchange<kT, kBlocks>(acc, s_acc, s_part, block_counters, counters_from, set, nb, seq, timeout_ticks,
                            s_tot, &s_failed, &state->error, cnt_pre,
                            (dbg_stamps && first_outer && ev == 1) ? dbg_stamps + 32 : nullptr);
        counters_from = 0;
        if (px.nranks > 1 && wave == 0 && !s_failed) {
            // ranks of one node: this GPU's totals become the totals over all ranks
            if (lane == 31) s_tot[31] = (double)n;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ten times the patience of the in-GPU waits: the peers are other processes
            global_exchange(px, s_tot, seq, timeout_ticks * 10, &s_failed, blockIdx.x == 0, lane);
        }
        LM_STAMP(3);
        // lom_debug_lm_trace: the point and the totals of every evaluation of this solve, as the
        // policy is about to see them ([ev][40]: x[7], pad, sums[32]; [200] = evaluations recorded)
        if (dbg_trace && blockIdx.x == 0 && wave == 0 && !s_failed && ev < 5) {
            if (lane < 7) dbg_trace[ev * 40 + lane] = s_x[lane];
            if (lane < 31) dbg_trace[ev * 40 + 8 + lane] = s_tot[lane];
            if (lane == 31) dbg_trace[ev * 40 + 8 + 31] = px.nranks > 1 ? s_tot[31] : (double)n;
            if (lane == 0) dbg_trace[200] = (double)(ev + 1);
        }
        LmWave W_keep = W;
        LmShared S_keep = r_lm;
        double x_keep = 0.0;
#pragma nounroll
        for (int rep = 0; rep < (kPolicyTwice ? 2 : 1); rep++)
        if (wave == 0 && !s_failed) {
            if constexpr (kPolicyTwice) {
                if (rep == 0) {
                    if (lane < 7) x_keep = s_x[lane];
                } else {
                    W = W_keep;
                    r_lm = S_keep;
                    if (lane < 7) s_x[lane] = x_keep;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    LM_STAMP(3);
                }
            }
            // the first wave holds the totals (s_tot) and runs the policy (lm_core.hpp's, lane-parallel and
            // register-resident: lm_wave.hpp)
            int a;
            if (ev == 0) {
                if (lane == 0) {
                    counters[0] = s_tot[28];
                    counters[1] = s_tot[29];
                    counters[2] = s_tot[30];
                    counters[3] = px.nranks > 1 ? s_tot[31] : (double)n;
                }
                a = kRegState ? lmw2_begin<true>(W, r_lm, s_tot, s_x, init.prior_b, lane)
                              : lmw2_begin<false>(W, s_lm, s_tot, s_x, init.prior_b, lane);
            } else {
                a = kRegState ? lmw2_feed<true>(W, r_lm, s_tot, s_x, init.prior_b, lane)
                              : lmw2_feed<false>(W, s_lm, s_tot, s_x, init.prior_b, lane);
            }
            // the point of the next evaluation lands in s_x
            if (a == LM_PROPOSE)
                a = kRegState ? lmw2_propose<true>(W, r_lm, s_x, lane) : lmw2_propose<false>(W, s_lm, s_x, lane);
            if (lane == 0) s_action = a;
        }
        LM_STAMP(4);
        __syncthreads();
        if (s_failed) {  // uniform over the workgroup
            if (tid == 0) {
                __hip_atomic_store(&state->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (px.nranks > 1) p2p_publish_abort(px);  // the peers leave their waits for this rank at once
                __hip_atomic_store(&report->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        action = s_action;
    }
#undef LM_STAMP
    if (blockIdx.x != 0 || tid != 0) return;
    // ---- end of the outer iteration (workgroup 0, one lane) ----
    const LmShared &S = kRegState ? r_lm : s_lm;
    prev.outer_done = s_prev_i[0];
    prev.lm_iterations = s_prev_i[1];
    prev.evaluations = s_prev_i[2];
    prev.valid_total = s_prev[0];
    prev.cand_total = s_prev[1];
    prev.occ_total = s_prev[2];
    prev.queries_total = s_prev[3];
    const int outer = prev.outer_done;
    float pq[4], pt[3];
    for (int a = 0; a < 4; a++) pq[a] = (float)S.x[a];      // :161-164
    for (int a = 0; a < 3; a++) pt[a] = (float)S.x[4 + a];  // :165-167
    const int finished = ((S.last_step_norm < 1e-4 && outer > 3) || outer + 1 >= 35) ? 1 : 0;  // :117, :169-172
    AlignState st;
    float R[9];
    rotation_matrix(pq, R);  // voxel_grid.h:212
    for (int i = 0; i < 9; i++) st.P.R[i] = (double)R[i];
    for (int i = 0; i < 3; i++) st.P.t[i] = (double)pt[i];
    st.P.max_sq = init.max_sq;
    for (int a = 0; a < 3; a++) st.pose_t[a] = pt[a];
    for (int a = 0; a < 4; a++) st.pose_q[a] = pq[a];
    st.finished = finished;
    st.error = 0;
    st.outer_done = outer + 1;
    st.lm_iterations = prev.lm_iterations + S.recorded;
    st.evaluations = prev.evaluations + S.evaluations;
    st.pad = 0;
    st.valid_last = counters[0];
    st.valid_total = prev.valid_total + counters[0];
    st.cand_total = prev.cand_total + counters[1];
    st.occ_total = prev.occ_total + counters[2];
    st.queries_total = prev.queries_total + counters[3];
    st.final_cost = S.cost;
    st.last_step_norm = S.last_step_norm;
    *state = st;
    // The host reads its first report after the fifth outer iteration (the stop rule cannot fire
    // earlier, and it enqueued five pairs at once): the reports of iterations 1-4 would only cost
    // this kernel a PCIe round trip each.
    if (st.outer_done < kPairsAhead) return;
    // report: payload as system-scope stores, drained, then the sequence word
    unsigned long long *dst_w = reinterpret_cast<unsigned long long *>(report);
    auto put = [&](size_t byte_off, unsigned long long v) {
        __hip_atomic_store(dst_w + byte_off / 8, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto two_i = [](int lo, int hi) { return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32); };
    auto two_f = [](float lo, float hi) {
        return (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32);
    };
    put(offsetof(AlignReport, finished), two_i(st.finished, 0));
    put(offsetof(AlignReport, outer_done), two_i(st.outer_done, st.lm_iterations));
    put(offsetof(AlignReport, evaluations), two_i(st.evaluations, 0));
    put(offsetof(AlignReport, pose_t), two_f(pt[0], pt[1]));
    put(offsetof(AlignReport, pose_t) + 8, two_f(pt[2], pq[0]));
    put(offsetof(AlignReport, pose_t) + 16, two_f(pq[1], pq[2]));
    put(offsetof(AlignReport, pose_t) + 24, two_f(pq[3], 0.f));
    put(offsetof(AlignReport, valid_last), (unsigned long long)__double_as_longlong(st.valid_last));
    put(offsetof(AlignReport, valid_total), (unsigned long long)__double_as_longlong(st.valid_total));
    put(offsetof(AlignReport, cand_total), (unsigned long long)__double_as_longlong(st.cand_total));
    put(offsetof(AlignReport, occ_total), (unsigned long long)__double_as_longlong(st.occ_total));
    put(offsetof(AlignReport, queries_total), (unsigned long long)__double_as_longlong(st.queries_total));
    put(offsetof(AlignReport, final_cost), (unsigned long long)__double_as_longlong(st.final_cost));
    put(offsetof(AlignReport, last_step_norm), (unsigned long long)__double_as_longlong(st.last_step_norm));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(dst_w, report_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// multi-GPU path: fold the records of one launch in workgroup order -> LOM_NSUMS doubles in HBM
__global__ __launch_bounds__(64) void k_sum_records(const double *__restrict__ rec, uint32_t n_rec,
                                                    uint32_t n_queries, double *__restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= LOM_NSUMS) return;
    double v = 0.0;
    if (k < 31) {
        uint32_t b = 0;
        for (; b + 4 <= n_rec; b += 4) {  // independent loads in flight, fixed summation order
            const double a0 = rec[(size_t)b * kRecWords + k], a1 = rec[(size_t)(b + 1) * kRecWords + k];
            const double a2 = rec[(size_t)(b + 2) * kRecWords + k], a3 = rec[(size_t)(b + 3) * kRecWords + k];
            v += a0;
            v += a1;
            v += a2;
            v += a3;
        }
        for (; b < n_rec; b++) v += rec[(size_t)b * kRecWords + k];
    } else {
        v = (double)n_queries;
    }
    out[k] = v;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// max_sq: the f32 threshold the f32 squared distances are compared with (strictly below).  findMatchingPairs forms it
// as max_dist * max_dist in f32 (voxel_grid.h:215); getCorrespondence takes a double (:164), see threshold_f32()
static void pose_args(const float t[3], const float q[4], float max_sq, PoseArgs &P)
{
    float R[9];
    rotation_matrix(q, R);  // voxel_grid.h:212 transform.rotationMatrix().cast<double>()
    for (int i = 0; i < 9; i++) P.R[i] = (double)R[i];
    for (int i = 0; i < 3; i++) P.t[i] = (double)t[i];
    P.max_sq = max_sq;
}

static inline float sq_f32(float max_dist) { return max_dist * max_dist; }  // voxel_grid.h:215

// voxel_grid.h:184-186 compares the f32 squared norm, widened to double, with a double threshold: (double)d2 < max_sq.
// For f32 d2 that is d2 < the smallest f32 that is >= max_sq (equal to max_sq where that is an f32 value itself, as
// findMatchingPairs' always is): the kernel's f32 compare with THAT threshold decides every case the same way.
static inline float threshold_f32(double max_sq)
{
    if (!(max_sq > 0.0)) return 0.f;                 // nothing is < 0 (NaN: every compare false)
    if (max_sq >= (double)FLT_MAX) return INFINITY;  // every finite d2 passes (an infinite d2 does only below an infinite threshold: not reproduced)
    float f = (float)max_sq;                         // round to nearest
    if ((double)f < max_sq) f = std::nextafterf(f, INFINITY);
    return f;
}

constexpr uint32_t kMaxMatchBlocks = 256u * (uint32_t)kMatchMinWaves;  // one resident round: kMatchMinWaves workgroups of 4 waves per CU
// (a context on a partition of the GPU: one resident round of ITS compute units)
static uint32_t match_grid(uint32_t n, uint32_t partition_cus = 0)
{
    const uint32_t per_block = (uint32_t)(kMatchThreads / kMatchG);
    const uint32_t need = (n + per_block - 1) / per_block;
    const uint32_t cap = partition_cus ? partition_cus * (uint32_t)kMatchMinWaves : kMaxMatchBlocks;
    return std::max(1u, std::min(need, cap));
}

constexpr uint32_t kMaxEvalBlocks = 64;  // records per launch (the host polls this many words)
static uint32_t eval_grid(uint32_t n)
{
    const uint32_t need = (n + kEvalThreads - 1) / kEvalThreads;
    return std::max(1u, std::min(need, kMaxEvalBlocks));
}

struct ScanCtx {
    lom_map *m;
    const char *d_src;
    size_t stride;
    uint32_t n;
    uint32_t match_blocks;
    // the records of a previous search of THIS scan against this map are at scan_on (outer iterations >= 2 of an align):
    // the next search may take its temporal pruning bound from them (k_match<..., kPrev>)
    bool have_prev = false;
    bool counted = false;  // the last launch produced the reference-algorithm counts
    int prof_used = 0;
    double launch_s = 0.0, wait_s = 0.0;  // host time inside launch calls / polling for results
};

static inline double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int scan_buffers(lom_map *m, uint32_t n, bool want_stats)
{
    int rc;
    const size_t nn = std::max<uint32_t>(n, 1);
    if ((rc = ensure(m, m->scan_idx, nn * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scan_on, nn * sizeof(MatchRec))) != LOM_OK) return rc;
    if (want_stats && (rc = ensure(m, m->scan_stats, nn * sizeof(QStat))) != LOM_OK) return rc;
    if ((rc = ensure(m, m->partials, (size_t)kMaxEvalBlocks * kRecWords * 8)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->results, 1024 + (size_t)kMaxMatchBlocks * 16)) != LOM_OK) return rc;
    return LOM_OK;
}

static uint32_t *d_block_counters(lom_map *m) { return (uint32_t *)((char *)m->results.p + 1024); }
static double *d_sums(lom_map *m) { return (double *)m->results.p; }

static void server_stop(lom_map *m);

// chained: the pose comes from the AlignState in HBM (t, q unused)
// count_mode: -1 = as the handle says (LOM_OPT_COUNT_CANDIDATES), 0 / 1 = without / with the reference-algorithm counts
static int launch_match(ScanCtx &c, const float t[3], const float q[4], float max_sq, bool stats,
                        bool chained = false, int count_mode = -1)
{
    lom_map *m = c.m;
    PoseArgs P;
    std::memset(&P, 0, sizeof P);
    if (!chained) pose_args(t, q, max_sq, P);
    c.match_blocks = c.n ? match_grid(c.n, m->stream == m->own_stream ? m->partition_cus : 0u) : 0;
    server_stop(m);  // the previous outer iteration's evaluation server leaves before the new search
    const double t_launch = now_s();
    if (c.n) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (m->profiling) {
            // three events per (k_match, k_lm) pair -- before, between, behind --, read back once at the end of the align
            while (m->prof_events.size() < (size_t)(c.prof_used + 1) * 3) {
                hipEvent_t e;
                LOM_HIP(m, hipEventCreate(&e));
                m->prof_events.push_back(e);
            }
            e0 = m->prof_events[(size_t)c.prof_used * 3];
            e1 = m->prof_events[(size_t)c.prof_used * 3 + 1];
            c.prof_used++;
            LOM_HIP(m, hipEventRecord(e0, m->stream));
        }
        // <lanes per query, candidates per lane and trip, min waves per SIMD>: measured on C2 / C3
        // (tools/ab_match.py): <16,1,8> 9.2 / 37.3 us, <16,2,1> 8.9 / 41.0, <16,4,1> 10.0 / 43.5,
        // <16,2,8> and <16,4,8> spill and lose; 8 lanes per query 12.3 / 44.1, 32 lanes 10.1 / 44.5.
        // Round 2 (query preparation split over the row's lanes, -15 % VALU instructions): 70 VGPRs, so 7 waves
        // per SIMD and a grid capped at one resident round of that; held to 64 VGPRs it spills 4 and loses
        // (C2 / C3 in the loop: 8.5 / 33.3 us at 7 waves, 9.3 / 36.5 at 8).  Four candidate loads in flight per lane
        // (<16,4,4>, 74 VGPRs) on C2 / C3: 9.3 / 39.3 us -- it pays only where few waves share a SIMD (C5's 8k-point
        // matching cloud: frame 0.329 -> 0.295 ms; eight in flight: the same).  With the candidate search at three
        // LDS round trips instead of six (C2 / C3 / C4 8.3 / 33.9 / 58.1 -> 7.7 / 31.0 / 54.3 us, 68 VGPRs) two loads
        // in flight fit the 7-wave budget (66 VGPRs): C2 the same, C3 31.6 -> 30.3 us.
        // A small cloud leaves the SIMDs with two or three waves each: nothing hides a round trip, so each lane keeps
        // four candidate loads in flight (<16,4,4>: 128-VGPR budget, one resident round up to 16384 queries).
        auto launch = [&](auto kernel, QStat *st, const AlignState *as) {
            hipLaunchKernelGGL(kernel, dim3(c.match_blocks), dim3(kMatchThreads), 0, m->stream, view_of(m), c.d_src, c.stride,
                               c.n, P, (int32_t *)m->scan_idx.p, (MatchRec *)m->scan_on.p, st, d_block_counters(m),
                               (unsigned long long *)nullptr, as);
        };
        QStat *st = (stats && !chained) ? (QStat *)m->scan_stats.p : (QStat *)nullptr;
        // (a chained launch always follows a search of the same scan: launch_pair's first pair is not chained)
        const bool prev = (chained || c.have_prev) && !m->opt_no_temporal;
        const bool count = count_mode < 0 ? m->opt_count : count_mode != 0;
        const AlignState *as = chained ? (const AlignState *)m->align_state.p : (const AlignState *)nullptr;
        constexpr int W = kMatchMinWaves;
        if (chained) {
            if (prev && count) launch(k_match<kMatchG, kMatchRows, W, false, true, true, true>, st, as);
            else if (prev) launch(k_match<kMatchG, kMatchRows, W, false, true, true, false>, st, as);
            else if (count) launch(k_match<kMatchG, kMatchRows, W, false, true, false, true>, st, as);
            else launch(k_match<kMatchG, kMatchRows, W, false, true, false, false>, st, as);
        } else {
            if (prev && count) launch(k_match<kMatchG, kMatchRows, W, false, false, true, true>, st, as);
            else if (prev) launch(k_match<kMatchG, kMatchRows, W, false, false, true, false>, st, as);
            else if (count) launch(k_match<kMatchG, kMatchRows, W, false, false, false, true>, st, as);
            else launch(k_match<kMatchG, kMatchRows, W, false, false, false, false>, st, as);
        }
        LOM_HIP(m, hipGetLastError());
        c.counted = count;
        c.have_prev = true;
        if (m->profiling) LOM_HIP(m, hipEventRecord(e1, m->stream));
    }
    c.launch_s += now_s() - t_launch;
    return LOM_OK;
}

constexpr size_t kEvalLdsBytes = (size_t)28 * kAccStride * sizeof(double);
// Every in-kernel wait is bounded by the handle's patience (lom_map::patience_ticks, 50 ms unless
// LOM_OPT_DEVICE_PATIENCE_TICKS changed it; tests shorten it to exercise the give-up paths).

static int eval_kernel_attrs(lom_map *m)
{
    if (m->eval_attr_set) return LOM_OK;
    LOM_HIP(m, hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kEvalLdsBytes));
    LOM_HIP(m, hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval_server),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEvalLdsBytes));
    m->eval_attr_set = true;
    return LOM_OK;
}

// tell a resident evaluation server to leave (it exits within one poll of the command word)
static void server_stop(lom_map *m)
{
    if (!m->server_alive) return;
    EvalCmd *cmd = reinterpret_cast<EvalCmd *>(m->h_cmd);
    cmd->op = kCmdStop;
    __atomic_store_n(&cmd->seq, ++m->mail_seq, __ATOMIC_RELEASE);
    m->server_alive = false;
}

// wait for the nb records of evaluation `seq` and add them in workgroup order.
// returns LOM_OK, a negative status, or 1 when the stream went idle without the records
// (the server timed out and left; the caller relaunches)
static int collect_records(lom_map *m, uint32_t nb, unsigned long long seq, double out[LOM_NSUMS])
{
    uint64_t spins = 0;
    for (uint32_t b = 0; b < nb; b++) {
        const double *rec = m->h_mail + (size_t)b * kRecWords;
        volatile const unsigned long long *flag = reinterpret_cast<volatile const unsigned long long *>(rec + 31);
        while (*flag != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFF) == 0) {
                const hipError_t e = hipStreamQuery(m->stream);
                if (e == hipSuccess) {
                    if (*flag == seq) break;
                    return 1;
                } else if (e != hipErrorNotReady) {
                    return set_error(m, LOM_ERR_HIP, "stream failed while waiting for an evaluation", e);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int k = 0; k < 31; k++) out[k] += rec[k];
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
    std::memset(out, 0, LOM_NSUMS * 8);
    int rc = eval_kernel_attrs(m);
    if (rc != LOM_OK) return rc;
    const double t_launch = now_s();
    if (!mailbox) {
        const unsigned long long seq = ++m->mail_seq;
        if (nb) {
            hipLaunchKernelGGL(k_eval, dim3(nb), dim3(kEvalThreads), kEvalLdsBytes, m->stream,
                               (const MatchRec *)m->scan_on.p, c.n, E, (const uint32_t *)d_block_counters(m),
                               fresh_match ? c.match_blocks : 0u, (double *)m->partials.p, seq);
        }
        hipLaunchKernelGGL(k_sum_records, dim3(1), dim3(64), 0, m->stream, (const double *)m->partials.p, nb, c.n,
                           d_sums(m));
        LOM_HIP(m, hipGetLastError());
        c.launch_s += now_s() - t_launch;
        const double t_wait = now_s();
        rc = ensure(m, m->gather, (size_t)m->nranks * LOM_NSUMS * 8);
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
        c.wait_s += now_s() - t_wait;
    } else if (nb) {
        for (int attempt = 0;; attempt++) {
            unsigned long long seq;
            const double t_l = now_s();
            if (fresh_match || !m->server_alive) {
                // (re)start the evaluation server of this outer iteration; its first evaluation is this one
                EvalCmd *cmd = reinterpret_cast<EvalCmd *>(m->h_cmd);
                seq = ++m->mail_seq;
                hipLaunchKernelGGL(k_eval_server, dim3(nb), dim3(kEvalThreads), kEvalLdsBytes, m->stream,
                                   (const MatchRec *)m->scan_on.p, c.n, E, (const uint32_t *)d_block_counters(m),
                                   fresh_match ? c.match_blocks : 0u, m->d_mail, seq,
                                   reinterpret_cast<const EvalCmd *>(m->d_cmd), (unsigned long long)cmd->seq,
                                   m->patience_ticks);
                LOM_HIP(m, hipGetLastError());
                m->server_alive = true;
            } else {
                EvalCmd *cmd = reinterpret_cast<EvalCmd *>(m->h_cmd);
                for (int a = 0; a < 4; a++) cmd->q[a] = q[a];
                for (int a = 0; a < 3; a++) cmd->t[a] = t[a];
                cmd->op = kCmdEval;
                seq = ++m->mail_seq;
                __atomic_store_n(&cmd->seq, seq, __ATOMIC_RELEASE);  // payload before the sequence word
            }
            const double t_w = now_s();
            c.launch_s += t_w - t_l;
            std::memset(out, 0, LOM_NSUMS * 8);
            rc = collect_records(m, nb, seq, out);
            c.wait_s += now_s() - t_w;
            if (m->opt_debug_timing)
                fprintf(stderr, "eval %s launch %.1f us wait %.1f us\n", fresh_match ? "fresh" : "fixed",
                        (t_w - t_l) * 1e6, (now_s() - t_w) * 1e6);
            if (rc == LOM_OK) break;
            if (rc < 0) return rc;
            m->server_alive = false;  // the server timed out and left (host was away > 50 ms): start another
            fresh_match = false;      // counters were already folded, or are folded again below
            if (attempt >= 3) return set_error(m, LOM_ERR_HIP, "evaluation server did not answer");
        }
        out[31] = (double)c.n;
    }
    // counters of the last k_match are summed on its first evaluation only
    if (fresh_match) {
        for (int k = 0; k < 4; k++) m->last_counters[k] = out[28 + k];
    } else {
        for (int k = 0; k < 3; k++) out[28 + k] = m->last_counters[k];
    }
    if (m->host_comm) {  // ranks of one node: the hosts exchange their 32 sums through shared memory
        double mine[LOM_NSUMS];
        std::memcpy(mine, out, sizeof mine);
        const double t_x = now_s();
        rc = host_exchange_sums(m, mine, out);
        c.wait_s += now_s() - t_x;
        if (rc != LOM_OK) return rc;
    }
    return LOM_OK;
}

static int hook_match_eval(void *user, const float pt[3], const float pq[4], const double q[4], const double t[3],
                           double out[LOM_NSUMS])
{
    ScanCtx &c = *(ScanCtx *)user;
    int rc = launch_match(c, pt, pq, sq_f32(0.3f), false);  // cloud_matcher.cpp:139
    if (rc != LOM_OK) return rc;
    return launch_eval(c, q, t, true, out);
}

static int hook_eval_fixed(void *user, const double q[4], const double t[3], double out[LOM_NSUMS])
{
    ScanCtx &c = *(ScanCtx *)user;
    return launch_eval(c, q, t, false, out);
}

static P2pArgs p2p_args(const lom_map *m)
{
    P2pArgs A;
    for (int r = 0; r < kP2pMaxRanks; r++) A.peer[r] = m->p2p ? (XWord *)m->p2p_peer[r] : nullptr;
    A.rank = m->p2p ? m->rank : 0;
    A.nranks = m->p2p ? m->nranks : 1;
    A.set_base = 0;
    A.epoch = ~0ull;
    return A;
}

void p2p_detach(lom_map *m)
{
    if (!m->p2p_local) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    for (int r = 0; r < kP2pMaxRanks; r++) {
        if (m->p2p_peer[r] && m->p2p_peer[r] != m->p2p_local) (void)hipIpcCloseMemHandle(m->p2p_peer[r]);
        m->p2p_peer[r] = nullptr;
    }
    (void)hipFree(m->p2p_local);
    m->p2p_local = nullptr;
    m->p2p = false;
}

// Single GPU, no exchange: the outer loop runs on the device.  One (k_match, k_lm) pair per outer
// iteration; the pose travels from pair to pair through AlignState in HBM, so the host enqueues
// pairs without waiting for results.  cloud_matcher.cpp:169-172 cannot stop before the fifth outer
// iteration (i > 3): five pairs go out at once, then one pair per report until `finished`.
// returned by align_chained when a workgroup of k_lm gave up waiting for the others (they are not all
// resident: a caller sharing the GPU, a CU mask) or for a peer rank: the caller redoes the align
// through the host-driven loop
constexpr int kDeviceLoopGaveUp = 100;

// k_lm's workgroups wait for each other inside the kernel, so all of them must be resident at once:
// the grid never exceeds what the occupancy query admits on this device.
// Workgroup size: 256 threads for clouds that 64 such workgroups cover with one point per lane (<= 16,384 points) or
// with TWO points per lane, both in registers for the whole solve (<= 32,768: the VLP16 scan of C2), else 512:
// the wave-level reduction is bound by the CU's f64 issue rate, and four waves -- one per SIMD -- are through it
// in half the time of eight; the final sum adds 8 partial sums instead of 16; two register points per lane are
// accumulated stage by stage so that their independent chains interleave (1.4k cycles for the two against 0.95k for
// one).  C2 (profiles/r03_*): k_lm 18.9 -> 17.8 us per launch against 512 threads with one point per lane; eight
// points per lane (C3 on 64 workgroups of 256) lose: 24.4-25.5 against 22.4 us.
// Clouds beyond what 64 workgroups of 512 cover with two points per lane (C3, C4 on one GPU) take up to 128 workgroups:
// the accumulation halves, the gather reads twice as many records (on C2-sized clouds that trade loses).
constexpr uint32_t kLmSmallThreads = 256;
enum LmShape { kLmSmall = 0, kLmMid = 1, kLmBig = 2, kLmSmall2 = 3 };
static LmShape lm_shape(uint32_t n)
{
    if (n <= kMaxLmBlocks * kLmSmallThreads) return kLmSmall;
    if (n <= 2u * kMaxLmBlocks * kLmSmallThreads) return kLmSmall2;  // 256 threads, two points per lane in registers
    // (for C3's 124k points 128 workgroups of 256 threads with four points per lane in registers -- no point re-read
    // per evaluation -- measured the same as 512 threads with one: align 0.2315-0.2324 against 0.2314-0.2335 ms on one
    // box; eight per lane for C4's 248k points lose, 0.448 against 0.343 ms: AGPR spills, eight points in a row)
    return n <= 2u * kMaxLmBlocks * (uint32_t)kEvalThreads ? kLmMid : kLmBig;
}

static int lm_block_limit(lom_map *m, LmShape shape, uint32_t *out)
{
    uint32_t &cached = m->lm_max_blocks[shape];
    if (!cached) {
        int per_cu = 0, cus = 0;
        if (shape == kLmSmall)
            LOM_HIP(m, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                           &per_cu, reinterpret_cast<const void *>(k_lm<(int)kLmSmallThreads>), (int)kLmSmallThreads, 0));
        else if (shape == kLmSmall2)
            LOM_HIP(m, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                           &per_cu, reinterpret_cast<const void *>(k_lm<(int)kLmSmallThreads, (int)kMaxLmBlocks, 2>),
                           (int)kLmSmallThreads, 0));
        else if (shape == kLmMid)
            LOM_HIP(m, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                           &per_cu, reinterpret_cast<const void *>(k_lm<kEvalThreads>), kEvalThreads, 0));
        else
            LOM_HIP(m, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                           &per_cu, reinterpret_cast<const void *>(k_lm<kEvalThreads, (int)kMaxLmBlocksBig>), kEvalThreads, 0));
        LOM_HIP(m, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
        if (m->partition_cus) cus = (int)m->partition_cus;  // this handle's stream only reaches its slice of the device
        cached = (uint32_t)std::max(1, per_cu * cus);
    }
    *out = cached;
    return LOM_OK;
}

static int align_chained(lom_map *m, const char *d_src, size_t n, size_t stride, const float guess_t[3],
                         const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats,
                         double *trace_out = nullptr, int trace_outer = 0)
{
    static_assert(offsetof(AlignReport, finished) == 8 && offsetof(AlignReport, outer_done) == 16 &&
                      offsetof(AlignReport, evaluations) == 24 && offsetof(AlignReport, pose_t) == 32 &&
                      offsetof(AlignReport, valid_last) == 64 && sizeof(AlignReport) <= 256,
                  "AlignReport is written as 64-bit words");
    int rc = scan_buffers(m, (uint32_t)n, false);
    if (rc != LOM_OK) return rc;
    if (!m->align_state.p) m->align_state_dirty = true;
    if ((rc = ensure(m, m->align_state, sizeof(AlignState))) != LOM_OK) return rc;
    if (m->align_state_dirty) {
        // a fresh allocation, or an align that ended in a give-up: its error flag must not be mistaken for this one's
        LOM_HIP(m, hipMemsetAsync(m->align_state.p, 0, sizeof(AlignState), m->stream));
        m->align_state_dirty = false;
    }
    if (!m->xrec.p) {
        const size_t bytes = (size_t)2 * kMaxLmBlocksBig * kRecWords * sizeof(XWord);
        if ((rc = ensure(m, m->xrec, bytes)) != LOM_OK) return rc;
        LOM_HIP(m, hipMemsetAsync(m->xrec.p, 0, bytes, m->stream));
    }
    if ((rc = eval_kernel_attrs(m)) != LOM_OK) return rc;
    ScanCtx c{m, d_src, stride, (uint32_t)n, 0};
    LmInit init;
    for (int a = 0; a < 3; a++) init.t[a] = guess_t[a];  // cloud_matcher.cpp:107
    for (int a = 0; a < 4; a++) init.q[a] = guess_q[a];
    for (int a = 0; a < 3; a++) init.prior_b[a] = (double)guess_t[a];  // :153
    init.max_sq = 0.3f * 0.3f;                                          // :139, voxel_grid.h:215
    uint32_t nb_limit = 0;
    // ranks of one node keep to 64 workgroups each: a shard is an eighth of the cloud, and ranks that share a GPU
    // (tests, rehearsals) must all be resident together
    LmShape shape = lm_shape(c.n);
    if (m->p2p && shape == kLmBig) shape = kLmMid;
    const uint32_t lm_threads = (shape == kLmSmall || shape == kLmSmall2) ? kLmSmallThreads : (uint32_t)kEvalThreads;
    if ((rc = lm_block_limit(m, shape, &nb_limit)) != LOM_OK) return rc;
    const uint32_t nb = std::min(std::min(std::max(1u, (c.n + lm_threads - 1) / lm_threads),
                                          shape == kLmBig ? kMaxLmBlocksBig : kMaxLmBlocks),
                                 nb_limit);
    double *d_trace = nullptr;  // lom_debug_lm_trace: k_lm of outer iteration `trace_outer` records its evaluations
    if (trace_out) {
        if ((rc = ensure(m, m->dbg_trace, 201 * 8)) != LOM_OK) return rc;
        d_trace = (double *)m->dbg_trace.p;
        LOM_HIP(m, hipMemsetAsync(d_trace, 0, 201 * 8, m->stream));
    }
    volatile AlignReport *rp = reinterpret_cast<volatile AlignReport *>(m->h_report);
    rp->error = 0;
    const unsigned long long seq0 = m->report_seq;
    int launched = 0;
    P2pArgs px = p2p_args(m);
    if (m->p2p) px.epoch = ++m->p2p_epoch;  // the same count on every rank: ranks issue the same sequence of aligns
    const int give_up_outer = m->test_give_up_outer;  // one shot
    m->test_give_up_outer = -1;
    unsigned long long *dbg = nullptr;  // LOM_OPT_DEBUG_LM_STAMPS: phase stamps of the last k_lm of the align
    if (m->opt_debug_lm) {
        if ((rc = ensure(m, m->dbg_stamps, 4096)) != LOM_OK) return rc;
        dbg = (unsigned long long *)m->dbg_stamps.p;
        LOM_HIP(m, hipMemsetAsync(dbg, 0, 40 * 8, m->stream));
    }
    int lm_events = 0;
    auto launch_pair = [&]() -> int {
        const int i = launched;
        int r = launch_match(c, guess_t, guess_q, sq_f32(0.3f), false, i > 0);
        if (r != LOM_OK) return r;
        const double t_l = now_s();
        m->lm_seq += 8;  // a solve spends at most 5 evaluations
        px.set_base = (int)((m->lm_launches++ & 1ull) * 2ull);  // same launch count on every rank
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(nb), dim3(lm_threads), 0, m->stream, (const MatchRec *)m->scan_on.p, c.n,
                               (AlignState *)m->align_state.p, init, i == 0 ? 1 : 0,
                               (const uint32_t *)d_block_counters(m), c.match_blocks, (XWord *)m->xrec.p, m->lm_seq,
                               reinterpret_cast<AlignReport *>(m->d_report), seq0 + (unsigned long long)i + 1,
                               m->patience_ticks, dbg, px,
                               (d_trace && i == trace_outer) ? d_trace : (double *)nullptr, i == give_up_outer ? 1 : 0);
        };
        if (shape == kLmSmall)
            launch(k_lm<(int)kLmSmallThreads>);
        else if (shape == kLmSmall2 && m->opt_debug_lm_twice)
            launch(k_lm<(int)kLmSmallThreads, (int)kMaxLmBlocks, 2, true>);
        else if (shape == kLmSmall2)
            launch(k_lm<(int)kLmSmallThreads, (int)kMaxLmBlocks, 2>);

        else if (shape == kLmMid)
            launch(k_lm<kEvalThreads>);
        else
            launch(k_lm<kEvalThreads, (int)kMaxLmBlocksBig>);
        LOM_HIP(m, hipGetLastError());
        if (m->profiling && c.prof_used) {
            LOM_HIP(m, hipEventRecord(m->prof_events[(size_t)(c.prof_used - 1) * 3 + 2], m->stream));
            lm_events++;
        }
        c.launch_s += now_s() - t_l;
        launched++;
        return LOM_OK;
    };
    for (int i = 0; i < kPairsAhead; i++)
        if ((rc = launch_pair()) != LOM_OK) return rc;
    // a caller that follows the align with radiusCleanup(result translation) (lidar_odometry.cpp:65-67) has said so: the
    // cleanup's scan goes out behind the pairs (an align that needs more than these finds it undone and scans later)
    if (m->spec_radius > 0.f && !m->p2p && !trace_out && !dbg) cleanup_scan_behind_align(m);
    m->spec_radius = 0.f;
    if (m->idle_hook) {  // the caller's own work for the ~0.1 ms this thread would only watch the report
        void (*fn)(void *) = m->idle_hook;
        m->idle_hook = nullptr;
        const double t_h = now_s();
        fn(m->idle_user);
        c.launch_s += now_s() - t_h;
    }
    for (;;) {
        const double t_w = now_s();
        const unsigned long long want = seq0 + (unsigned long long)launched;
        uint64_t spins = 0;
        while (rp->seq != want) {
            __builtin_ia32_pause();
            if (rp->error) break;
            if ((++spins & 0x3FFF) == 0) {
                const hipError_t e = hipStreamQuery(m->stream);
                if (e == hipSuccess) {
                    if (rp->seq == want) break;
                    m->report_seq = want;
                    return set_error(m, LOM_ERR_HIP, "device solve ended without a report");
                } else if (e != hipErrorNotReady) {
                    m->report_seq = want;
                    return set_error(m, LOM_ERR_HIP, "stream failed during the device solve", e);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        c.wait_s += now_s() - t_w;
        if (rp->error) {
            // the kernels still enqueued see the flag in AlignState and return at once
            (void)hipStreamSynchronize(m->stream);
            m->report_seq = want;
            m->align_state_dirty = true;
            set_error(m, LOM_ERR_HIP, "device solve: a workgroup timed out waiting for the others");
            return kDeviceLoopGaveUp;
        }
        if (rp->finished || launched >= 35) break;
        if ((rc = launch_pair()) != LOM_OK) return rc;
    }
    m->report_seq = seq0 + (unsigned long long)launched;
    lom_align_stats st;
    std::memset(&st, 0, sizeof st);
    st.outer_iterations = rp->outer_done;
    st.match_launches = rp->outer_done;
    st.lm_iterations = rp->lm_iterations;
    st.evaluations = rp->evaluations;
    st.valid_last = (int64_t)rp->valid_last;
    st.cand_total = (int64_t)rp->cand_total;
    st.occ_total = (int64_t)rp->occ_total;
    st.queries = (int64_t)rp->queries_total;
    // B(q) = 12 + 27*16 + 12*cand(q) + 12*valid(q) -- known only when the searches produced the counts
    st.algorithmic_bytes = c.counted ? 444.0 * rp->queries_total + 12.0 * rp->cand_total + 12.0 * rp->valid_total : 0.0;
    st.final_cost = rp->final_cost;
    st.last_step_norm = rp->last_step_norm;
    float pq[4] = {rp->pose_q[0], rp->pose_q[1], rp->pose_q[2], rp->pose_q[3]};
    {   // cloud_matcher.cpp:175 rotation.normalize(), f32
        const float n2 = (pq[0] * pq[0] + pq[1] * pq[1]) + (pq[2] * pq[2] + pq[3] * pq[3]);
        const float nn = std::sqrt(n2);
        for (int a = 0; a < 4; a++) pq[a] = pq[a] / nn;
    }
    for (int a = 0; a < 3; a++) out_t[a] = rp->pose_t[a];
    for (int a = 0; a < 4; a++) out_q[a] = pq[a];
    if (m->profiling && c.prof_used) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        // kernels enqueued beyond the end of the loop return at once: only the executed iterations count
        const int executed = std::min(c.prof_used, (int)rp->outer_done);
        for (int i = 0; i < executed; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->prof_events[(size_t)i * 3], m->prof_events[(size_t)i * 3 + 1]) == hipSuccess)
                st.match_kernel_ms += ms;
            if (i < lm_events &&
                hipEventElapsedTime(&ms, m->prof_events[(size_t)i * 3 + 1], m->prof_events[(size_t)i * 3 + 2]) == hipSuccess)
                st.lm_kernel_ms += ms;
        }
        st.profiled_launches = executed;
        st.lm_profiled_launches = std::min(executed, lm_events);
    }
    st.host_launch_ms = c.launch_s * 1e3;
    st.host_wait_ms = c.wait_s * 1e3;
    st.lm_workgroups = (int32_t)nb;
    if (stats) *stats = st;
    if (trace_out) {
        LOM_HIP(m, hipMemcpyAsync(trace_out, d_trace, 201 * 8, hipMemcpyDeviceToHost, m->stream));
        LOM_HIP(m, hipStreamSynchronize(m->stream));
    }
    if (dbg) {
        unsigned long long h[40];
        LOM_HIP(m, hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, m->stream));
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        fprintf(stderr, "k_lm eval 1 reduce+exchange: LDS rows %llu, publish %llu, gather %llu, final sum %llu cycles\n",
                h[33] - h[32], h[34] - h[33], h[35] - h[34], h[36] - h[35]);
        for (int ev = 0; ev < 5 && h[ev * 5]; ev++)
            fprintf(stderr, "k_lm eval %d: at %llu: accumulate %llu reduce+exchange %llu policy %llu cycles\n", ev,
                    h[ev * 5] - h[0], h[ev * 5 + 1] - h[ev * 5], h[ev * 5 + 3] - h[ev * 5 + 1],
                    h[ev * 5 + 4] - h[ev * 5 + 3]);
    }
    return LOM_OK;
}

static int align_device_paths(lom_map *m, const char *d_src, size_t n, size_t stride, const float guess_t[3],
                              const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats);

static int align_device(lom_map *m, const char *d_src, size_t n, size_t stride, const float guess_t[3],
                        const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    const int rc = align_device_paths(m, d_src, n, stride, guess_t, guess_q, out_t, out_q, stats);
    // lom_map_radius_cleanup_after_align and lom_map_set_align_idle_hook arm ONE align, whichever path it took and however it ended
    m->spec_radius = 0.f;
    m->idle_hook = nullptr;
    return rc;
}

static int align_device_paths(lom_map *m, const char *d_src, size_t n, size_t stride, const float guess_t[3],
                              const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    if (n >= 0x7FFFFFFFull) return set_error(m, LOM_ERR_ARG, "too many source points");
    {   // an insert nobody has looked at since (no lom_map_status): the search must see its points
        const int rcp = resolve_pending(m);
        if (rcp != LOM_OK) return rcp;
    }
    m->profiling = m->profile_period > 0 && (m->align_count++ % (unsigned)m->profile_period) == 0;
    bool fell_back = false;
    if (!m->comm && (!m->host_comm || m->p2p) && !m->opt_host_lm) {
        server_stop(m);
        int rc = align_chained(m, d_src, n, stride, guess_t, guess_q, out_t, out_q, stats);
        if (m->p2p) {
            // The ranks agree on the outcome of EVERY align, whatever happened on this one: a time-out that lands
            // on the last exchange of an align lets the peers that already hold all words finish with LOM_OK, and
            // a rank that gave up -- or failed for good -- must neither redo the align alone nor leave its peers
            // waiting (the host exchange pairs operations by its own counter only).  Two counts through the host
            // exchange: ranks that gave up (recoverable: everybody redoes the align over the host exchange) and
            // ranks that failed for good (nobody continues).  The deadline outlasts the device side: a rank can
            // be late by its kernels' patience for a peer rank, once per pair still enqueued at worst (the abort
            // words normally cut that to one patience), and an exchange nobody completes is ABANDONED, which
            // every late rank sees (comm.cpp) -- round 2's failure was a fixed 60 s here against 10 x 10 s there.
            const bool gave_up = rc == kDeviceLoopGaveUp, hard = rc != LOM_OK && !gave_up;
            double verdict[2] = {gave_up ? 1.0 : 0.0, hard ? 1.0 : 0.0};
            const double cross_s = (double)m->patience_ticks * 10.0 * 1e-8;
            const double deadline_s = 30.0 + 2.0 * (kPairsAhead + 1) * cross_s;
            if (host_comm_allreduce_deadline(m->host_comm, verdict, 2, deadline_s) != LOM_OK) {
                m->p2p = false;
                const std::string why = std::string("agreement after a device-to-device align failed: ") + host_comm_error(m->host_comm);
                return set_error(m, LOM_ERR_COMM, why.c_str());
            }
            if (verdict[1] != 0.0) {  // some rank cannot continue: the same for all
                m->p2p = false;
                (void)hipStreamSynchronize(m->stream);
                if (hard) return rc;
                return set_error(m, LOM_ERR_COMM, "a peer rank failed during a device-to-device align");
            }
            if (verdict[0] == 0.0) return LOM_OK;
            fprintf(stderr, "lidar_odometry_amd: device-to-device exchange given up on %d rank(s) (%s); rank %d redoes the align over the host exchange\n",
                    (int)verdict[0], gave_up ? m->last_error.c_str() : "a peer gave up", m->rank);
            (void)hipStreamSynchronize(m->stream);
            m->p2p = false;
        } else if (rc != kDeviceLoopGaveUp) {
            return rc;
        }
        // single GPU: k_lm's workgroups were not all resident within their patience (another process or
        // handle on the GPU, a CU mask): same align again through the host-driven loop, whose
        // workgroups never wait for each other
        m->last_error.clear();
        fell_back = true;
    }
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
    server_stop(m);
    if (!c.counted) st.algorithmic_bytes = 0.0;  // (SURVEY.md 8d's bytes need the counts: LOM_OPT_COUNT_CANDIDATES)
    if (rc != LOM_OK) {
        if (m->last_error.empty()) set_error(m, rc, "align failed");
        // ranks of one node: a rank that leaves the loop tells its peers (they would wait for its sums otherwise)
        if (m->host_comm) (void)lom_host_comm_abort((lom_host_comm *)m->host_comm);
        return rc == LOM_ERR_HOOK ? LOM_ERR_HIP : rc;
    }
    if (m->profiling && c.prof_used) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        for (int i = 0; i < c.prof_used; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->prof_events[(size_t)i * 3], m->prof_events[(size_t)i * 3 + 1]) == hipSuccess)
                st.match_kernel_ms += ms;
        }
        st.profiled_launches = c.prof_used;
    }
    st.host_launch_ms = c.launch_s * 1e3;
    st.host_wait_ms = c.wait_s * 1e3;
    st.host_fallback = fell_back ? 1 : 0;
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

// one search (or two: first at the pose (t0, q0), then at (t, q) with the first one's records as the temporal bound)
static int64_t find_pairs_core(lom_map *m, const float *src, size_t n, size_t stride, const float *t0, const float *q0,
                               const float t[3], const float q[4], float max_sq, lom_correspondence *out)
{
    if (!m || (n && (!src || !out)) || !t || !q || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    if (n == 0) return 0;
    LOM_HIP(m, hipSetDevice(m->device));
    const char *d_src = nullptr;
    int rc = resolve_pending(m);
    if (rc != LOM_OK) return rc;
    if ((rc = stage_scan(m, src, n, stride, &d_src)) != LOM_OK) return rc;
    if ((rc = scan_buffers(m, (uint32_t)n, true)) != LOM_OK) return rc;
    ScanCtx c{m, d_src, stride, (uint32_t)n, 0};
    if (t0 && q0 && (rc = launch_match(c, t0, q0, max_sq, true)) != LOM_OK) return rc;
    if ((rc = launch_match(c, t, q, max_sq, true)) != LOM_OK) return rc;
    std::vector<int32_t> idx(n);
    std::vector<MatchRec> on(n);
    std::vector<QStat> st(n);
    LOM_HIP(m, hipMemcpyAsync(idx.data(), m->scan_idx.p, n * 4, hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipMemcpyAsync(on.data(), m->scan_on.p, n * sizeof(MatchRec), hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipMemcpyAsync(st.data(), m->scan_stats.p, n * sizeof(QStat), hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    // index = voxel_creation_index * max_points + in_voxel_index, the creation index counting the voxels the map HOLDS: where
    // a radius cleanup has left erased voxels' slabs in place (k_cleanup_mark) the slab number runs ahead of it
    std::vector<uint32_t> dense;
    const lom_map *mp = m->parent ? m->parent : m;
    const uint32_t n_dead = mp->n_dead, dead_below = mp->dead_below;
    if (n_dead) {
        std::vector<uint32_t> cnt(dead_below);
        LOM_HIP(m, hipMemcpy(cnt.data(), mp->d_slab_count, (size_t)dead_below * 4, hipMemcpyDeviceToHost));
        dense.resize(dead_below);
        uint32_t live = 0;
        for (uint32_t s = 0; s < dead_below; s++) {
            dense[s] = live;
            live += cnt[s] != 0u;
        }
    }
    const uint32_t K = mp->K;
    int64_t valid = 0;
    for (size_t i = 0; i < n; i++) {
        lom_correspondence &o = out[i];
        o.index = idx[i];
        if (n_dead && idx[i] >= 0) {
            const uint32_t slab = (uint32_t)idx[i] / K, j = (uint32_t)idx[i] % K;
            o.index = (int64_t)(slab < dead_below ? dense[slab] : slab - n_dead) * K + j;
        }
        o.origin[0] = on[i].ox;
        o.origin[1] = on[i].oy;
        o.origin[2] = on[i].oz;
        o.normal[0] = on[i].nx;
        o.normal[1] = on[i].ny;
        o.normal[2] = on[i].nz;
        o.sq_dist = st[i].sq_dist;
        o.n_cand = c.counted ? st[i].n_cand : 0u;  // the reference-algorithm counts, or nothing (LOM_OPT_COUNT_CANDIDATES)
        o.n_occ = c.counted ? st[i].n_occ : 0u;
        valid += idx[i] >= 0;
    }
    return valid;
}

int64_t lom_match_find_pairs(lom_map *m, const float *src, size_t n, size_t stride, const float t[3],
                             const float q[4], float max_dist, lom_correspondence *out)
{
    return find_pairs_core(m, src, n, stride, nullptr, nullptr, t, q, sq_f32(max_dist), out);
}

int64_t lom_match_find_pairs_sq(lom_map *m, const float *src, size_t n, size_t stride, const float t[3],
                                const float q[4], double max_dist_sq, lom_correspondence *out)
{
    return find_pairs_core(m, src, n, stride, nullptr, nullptr, t, q, threshold_f32(max_dist_sq), out);
}

int64_t lom_debug_find_pairs_after(lom_map *m, const float *src, size_t n, size_t stride, const float t_prev[3],
                                   const float q_prev[4], const float t[3], const float q[4], float max_dist,
                                   lom_correspondence *out)
{
    if (!t_prev || !q_prev) return LOM_ERR_ARG;
    return find_pairs_core(m, src, n, stride, t_prev, q_prev, t, q, sq_f32(max_dist), out);
}

int lom_comm_attach_p2p(lom_map *m, lom_host_comm *hc)
{
    if (!m || !hc) return LOM_ERR_ARG;
    if (m->comm) return set_error(m, LOM_ERR_STATE, "an RCCL communicator is already attached");
    LOM_HIP(m, hipSetDevice(m->device));
    int rank = 0, nranks = 1;
    if (host_comm_rank(hc, &rank, &nranks) != LOM_OK) return LOM_ERR_ARG;
    if (nranks > kP2pMaxRanks) return set_error(m, LOM_ERR_ARG, "device-to-device exchange: at most 8 ranks");
    p2p_detach(m);
    server_stop(m);
    // From here on every step is collective: a rank that fails locally still takes part in the
    // exchanges below, so that all ranks reach the same verdict.
    const size_t bytes = kP2pBufferBytes;  // four exchange sets + one abort word per rank
    m->p2p_epoch = 0;
    m->lm_launches = 0;  // the set pairs alternate with this count: the same on every rank from here on
    int ok = 1;
    struct Blob {
        hipIpcMemHandle_t handle;
        int ok;
    } mine, all[kP2pMaxRanks];
    std::memset(&mine, 0, sizeof mine);
    static_assert(sizeof(Blob) <= 256, "fits one host-exchange slot");
    // fine-grained (coherent across agents) device memory; plain device memory if that is refused --
    // the self-test below decides whether the exchange works on this machine
    if (hipExtMallocWithFlags(&m->p2p_local, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&m->p2p_local, bytes) != hipSuccess) {
            m->p2p_local = nullptr;
            ok = 0;
        }
    }
    if (ok && (hipMemsetAsync(m->p2p_local, 0, bytes, m->stream) != hipSuccess ||
               hipStreamSynchronize(m->stream) != hipSuccess))
        ok = 0;
    if (ok && nranks > 1 && hipIpcGetMemHandle(&mine.handle, m->p2p_local) != hipSuccess) ok = 0;
    mine.ok = ok;
    if (lom_host_comm_allgather(hc, &mine, sizeof mine, all) != LOM_OK) {
        p2p_detach(m);
        return set_error(m, LOM_ERR_COMM, "device-to-device exchange: handle exchange failed");
    }
    for (int r = 0; r < nranks; r++) ok = ok && all[r].ok;
    if (ok) {
        for (int r = 0; r < nranks && ok; r++) {
            if (r == rank) {
                m->p2p_peer[r] = m->p2p_local;
            } else if (hipIpcOpenMemHandle(&m->p2p_peer[r], all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                m->p2p_peer[r] = nullptr;
                ok = 0;
            }
        }
    }
    // barrier: every rank has its mappings before anybody stores through them
    double sync[1] = {0.0};
    if (lom_host_comm_allreduce(hc, sync, 1) != LOM_OK) ok = 0;
    // sequence numbers: one epoch per attach, identical on all ranks (the host exchange has done the
    // same number of operations on every rank) and above every number this handle has used
    unsigned long long hseq = 0;
    (void)host_comm_rank(hc, &rank, &nranks, &hseq);
    m->lm_seq = std::max(m->lm_seq, hseq << 32);
    m->rank = rank;
    m->nranks = nranks;
    // self-test: 200 exchanges of known values through the mappings
    uint32_t res[2] = {1u, 1u};
    if (ok && scan_buffers(m, 1, false) != LOM_OK) ok = 0;
    if (ok) {
        m->p2p = true;
        const P2pArgs A = p2p_args(m);
        m->p2p = false;
        hipLaunchKernelGGL(k_p2p_selftest, dim3(1), dim3(64), 0, m->stream, A, m->lm_seq, 200,
                           m->patience_ticks, (uint32_t *)m->results.p);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(res, (uint32_t *)m->results.p, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            ok = 0;
        else if (res[0] != 0 || res[1] != 0)
            ok = 0;
    }
    m->lm_seq += 256;
    double verdict[1] = {ok ? 0.0 : 1.0};
    if (lom_host_comm_allreduce(hc, verdict, 1) != LOM_OK) verdict[0] = 1.0;
    if (verdict[0] != 0.0) {
        p2p_detach(m);
        m->rank = 0;
        m->nranks = 1;
        return set_error(m, LOM_ERR_COMM, "device-to-device exchange failed its self-test on some rank");
    }
    m->host_comm = hc;  // rank / nranks bookkeeping as with the host exchange; the caller keeps ownership
    m->p2p = true;
    return LOM_OK;
}

int lom_profile_match(lom_map *m, const float *d_src, size_t n, size_t stride, const float t[3], const float q[4],
                      float max_dist, int reps, double *avg_us_out, double *bytes_out, double *requested_bytes_out,
                      double *pair_avg_us_out)
{
    if (!m || !d_src || !n || !t || !q || reps < 1 || !avg_us_out || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    int rc = scan_buffers(m, (uint32_t)n, false);
    if (rc != LOM_OK) return rc;
    ScanCtx c{m, (const char *)d_src, stride, (uint32_t)n, 0};
    const bool was = m->profiling;
    m->profiling = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    LOM_HIP(m, hipEventCreate(&e0));
    LOM_HIP(m, hipEventCreate(&e1));
    // warm-up; it also leaves the records the launches behind it take their temporal bound from: the train is what the
    // searches of outer iterations >= 2 of an align run (LOM_OPT_NO_TEMPORAL_BOUND: what the first one runs)
    rc = launch_match(c, t, q, sq_f32(max_dist), false);
    if (rc == LOM_OK) rc = hipEventRecord(e0, m->stream) == hipSuccess ? LOM_OK : LOM_ERR_HIP;
    for (int i = 0; rc == LOM_OK && i < reps; i++) rc = launch_match(c, t, q, sq_f32(max_dist), false);
    if (rc == LOM_OK) rc = hipEventRecord(e1, m->stream) == hipSuccess ? LOM_OK : LOM_ERR_HIP;
    // the same launches with one event pair EACH (what the sampled in-loop measurement of lom_match_align* does):
    // the difference to the train above is what an event pair adds to a single short kernel
    double pair_ms = 0.0;
    if (rc == LOM_OK && pair_avg_us_out) {
        m->profiling = true;
        c.prof_used = 0;
        const int pr = std::min(reps, 128);
        for (int i = 0; rc == LOM_OK && i < pr; i++) rc = launch_match(c, t, q, sq_f32(max_dist), false);
        m->profiling = false;
        if (rc == LOM_OK && hipStreamSynchronize(m->stream) != hipSuccess) rc = LOM_ERR_HIP;
        for (int i = 0; rc == LOM_OK && i < c.prof_used; i++) {
            float ms1 = 0.f;
            if (hipEventElapsedTime(&ms1, m->prof_events[(size_t)i * 3], m->prof_events[(size_t)i * 3 + 1]) == hipSuccess)
                pair_ms += ms1;
        }
        *pair_avg_us_out = c.prof_used ? pair_ms * 1e3 / c.prof_used : 0.0;
        c.prof_used = 0;
    }
    // what the timed launches asked the memory system for: one more launch of the same form that tallies, per query, the
    // rows it read and the slots it looked up (with the counts every query looks up 27 and the rows are the per-workgroup
    // tally of the launch)
    double scanned = 0.0, probed = 0.0;
    const bool train_counted = c.counted;
    if (rc == LOM_OK && requested_bytes_out) {
        if (train_counted) {
            std::vector<uint32_t> bc((size_t)c.match_blocks * 4);
            if (hipMemcpyAsync(bc.data(), d_block_counters(m), bc.size() * 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
                hipStreamSynchronize(m->stream) != hipSuccess)
                rc = LOM_ERR_HIP;
            for (uint32_t b = 0; rc == LOM_OK && b < c.match_blocks; b++) scanned += bc[(size_t)b * 4 + 3];
        } else {
            std::vector<QStat> qs(n);
            if ((rc = scan_buffers(m, (uint32_t)n, true)) == LOM_OK) rc = launch_match(c, t, q, sq_f32(max_dist), true);
            if (rc == LOM_OK && (hipMemcpyAsync(qs.data(), m->scan_stats.p, n * sizeof(QStat), hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
                                 hipStreamSynchronize(m->stream) != hipSuccess))
                rc = LOM_ERR_HIP;
            for (size_t i = 0; rc == LOM_OK && i < n; i++) {
                scanned += qs[i].n_cand;
                probed += qs[i].n_occ;
            }
        }
    }
    // the algorithmic bytes need the reference-algorithm counts: one more launch that produces them
    double sums[LOM_NSUMS];
    const double qd[4] = {q[0], q[1], q[2], q[3]}, td[3] = {t[0], t[1], t[2]};
    if (rc == LOM_OK) rc = launch_match(c, t, q, sq_f32(max_dist), false, false, 1);
    if (rc == LOM_OK) rc = launch_eval(c, qd, td, true, sums);  // folds the counters of that launch
    server_stop(m);
    if (rc == LOM_OK && hipStreamSynchronize(m->stream) != hipSuccess) rc = LOM_ERR_HIP;
    float ms = 0.f;
    if (rc == LOM_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = LOM_ERR_HIP;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    m->profiling = was;
    if (rc != LOM_OK) return set_error(m, rc, "lom_profile_match failed");
    *avg_us_out = (double)ms * 1e3 / reps;
    if (bytes_out) *bytes_out = 444.0 * sums[31] + 12.0 * sums[29] + 12.0 * sums[28];
    if (requested_bytes_out) {
        // source point + slots looked up + rows of the scanned voxels + winner's normal + 52 B of output per query (+ 16 B
        // of the previous record where the temporal bound is in use)
        const double nq = sums[31];
        const double slots = train_counted ? 27.0 * nq : probed;
        const double prev_b = (c.have_prev && !m->opt_no_temporal) ? 16.0 * nq : 0.0;
        *requested_bytes_out = 12.0 * nq + 16.0 * slots + 12.0 * scanned + 12.0 * sums[28] + 52.0 * nq + prev_b;
    }
    return LOM_OK;
}

int lom_map_set_align_idle_hook(lom_map *m, void (*fn)(void *user), void *user)
{
    if (!m) return LOM_ERR_ARG;
    m->idle_hook = fn;
    m->idle_user = user;
    return LOM_OK;
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

// parity entry: one search at the f32 pose, then ONE evaluation of the reduced normal equations at (q, t)
// through the host-driven path's kernels (k_match + k_eval_server: accumulate_point, LDS reduction,
// record per workgroup, workgroup-ordered host sum)
int lom_debug_eval_sums(lom_map *m, const float *src, size_t n, size_t stride, const float pose_t[3],
                        const float pose_q[4], const double q[4], const double t[3], double out[LOM_NSUMS])
{
    if (!m || (n && !src) || !pose_t || !pose_q || !q || !t || !out || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    if (m->comm || m->host_comm) return set_error(m, LOM_ERR_STATE, "not with an attached exchange");
    LOM_HIP(m, hipSetDevice(m->device));
    m->last_error.clear();
    const char *d_src = nullptr;
    int rc = stage_scan(m, src, n, stride, &d_src);
    if (rc != LOM_OK) return rc;
    if ((rc = scan_buffers(m, (uint32_t)n, false)) != LOM_OK) return rc;
    ScanCtx c{m, d_src, stride, (uint32_t)n, 0};
    const bool was = m->profiling;
    m->profiling = false;
    rc = launch_match(c, pose_t, pose_q, sq_f32(0.3f), false);  // cloud_matcher.cpp:139
    if (rc == LOM_OK) rc = launch_eval(c, q, t, true, out);
    server_stop(m);
    m->profiling = was;
    if (rc == LOM_OK && hipStreamSynchronize(m->stream) != hipSuccess) rc = LOM_ERR_HIP;
    return rc;
}

// parity entry: a whole align on the device-resident path (k_match / k_lm chain) that also returns what
// k_lm's policy saw in outer iteration `outer_index`: for every evaluation of that solve the point
// x = [q, t] it was made at and the 32 totals after the in-kernel reduction and exchange
int lom_debug_lm_trace(lom_map *m, const float *src, size_t n, size_t stride, const float guess_t[3],
                       const float guess_q[4], int outer_index, double *trace_out, int *n_evals_out, float out_t[3],
                       float out_q[4], lom_align_stats *stats)
{
    if (!m || (n && !src) || !guess_t || !guess_q || !trace_out || !n_evals_out || !out_t || !out_q || stride < 12 ||
        (stride & 3) || outer_index < 0 || outer_index >= 35)
        return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    if (m->comm || m->host_comm) return set_error(m, LOM_ERR_STATE, "not with an attached exchange");
    LOM_HIP(m, hipSetDevice(m->device));
    m->last_error.clear();
    const char *d_src = nullptr;
    int rc = stage_scan(m, src, n, stride, &d_src);
    if (rc != LOM_OK) return rc;
    server_stop(m);
    m->profiling = false;
    double raw[201];
    rc = align_chained(m, d_src, n, stride, guess_t, guess_q, out_t, out_q, stats, raw, outer_index);
    if (rc == kDeviceLoopGaveUp) return LOM_ERR_HIP;
    if (rc != LOM_OK) return rc;
    const int ne = (int)raw[200];
    *n_evals_out = ne;
    for (int e = 0; e < ne && e < 5; e++) std::memcpy(trace_out + (size_t)e * 40, raw + (size_t)e * 40, 40 * sizeof(double));
    return LOM_OK;
}

// diagnostic: per-workgroup phase stamps of one correspondence launch (shader clock ticks)
int lom_debug_match_stamps(lom_map *m, const float *d_src, size_t n, size_t stride, const float t[3],
                           const float q[4], float max_dist, unsigned long long *stamps_out, size_t cap_blocks,
                           uint32_t *n_blocks_out)
{
    if (!m || !d_src || !n || !t || !q || !stamps_out || !n_blocks_out) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    int rc = scan_buffers(m, (uint32_t)n, false);
    if (rc != LOM_OK) return rc;
    server_stop(m);
    const uint32_t nb = match_grid((uint32_t)n);
    if (nb > cap_blocks) return LOM_ERR_ARG;
    unsigned long long *d_st = nullptr;
    LOM_HIP(m, hipMalloc(&d_st, (size_t)nb * 64));
    PoseArgs P;
    pose_args(t, q, sq_f32(max_dist), P);
    for (int rep = 0; rep < 3; rep++)  // the last launch's stamps are kept (warm caches, like an align)
        hipLaunchKernelGGL((k_match<kMatchG, kMatchRows, kMatchMinWaves, true>), dim3(nb), dim3(kMatchThreads), 0, m->stream, view_of(m),
                           (const char *)d_src, stride, (uint32_t)n, P, (int32_t *)m->scan_idx.p,
                           (MatchRec *)m->scan_on.p, (QStat *)nullptr, d_block_counters(m), d_st);
    hipError_t e = hipMemcpyAsync(stamps_out, d_st, (size_t)nb * 64, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    (void)hipFree(d_st);
    if (e != hipSuccess) return set_error(m, LOM_ERR_HIP, "stamp readback", e);
    *n_blocks_out = nb;
    return LOM_OK;
}

int lom_match_align_repeat(lom_map *m, const float *d_src, size_t n, size_t stride, const float guess_t[3],
                           const float guess_q[4], int reps, float out_t[3], float out_q[4],
                           lom_align_stats *total)
{
    if (!m || (n && !d_src) || !guess_t || !guess_q || !out_t || !out_q || reps < 1 || stride < 12 || (stride & 3))
        return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    m->last_error.clear();
    lom_align_stats acc;
    std::memset(&acc, 0, sizeof acc);
    for (int r = 0; r < reps; r++) {
        lom_align_stats st;
        const int rc = align_device(m, (const char *)d_src, n, stride, guess_t, guess_q, out_t, out_q, &st);
        if (rc != LOM_OK) return rc;
        acc.outer_iterations += st.outer_iterations;
        acc.lm_iterations += st.lm_iterations;
        acc.evaluations += st.evaluations;
        acc.match_launches += st.match_launches;
        acc.queries += st.queries;
        acc.valid_last = st.valid_last;
        acc.cand_total += st.cand_total;
        acc.occ_total += st.occ_total;
        acc.final_cost = st.final_cost;
        acc.last_step_norm = st.last_step_norm;
        acc.match_kernel_ms += st.match_kernel_ms;
        acc.algorithmic_bytes += st.algorithmic_bytes;
        acc.host_launch_ms += st.host_launch_ms;
        acc.host_wait_ms += st.host_wait_ms;
        acc.profiled_launches += st.profiled_launches;
        acc.host_fallback += st.host_fallback;
        acc.lm_kernel_ms += st.lm_kernel_ms;
        acc.lm_profiled_launches += st.lm_profiled_launches;
        acc.lm_workgroups = st.lm_workgroups;
    }
    if (total) *total = acc;
    return LOM_OK;
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
