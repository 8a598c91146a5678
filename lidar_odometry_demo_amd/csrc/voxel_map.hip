// Device-resident voxel map: the MI355X counterpart of the reference's
// VoxelGrid container (src/voxel_grid.h:17-257) and VoxelWithPlanes payload
// (src/voxel_with_planes.h:10-36).
//
//   addCloud / addCloudWithoutNormals (:77-110)  -> lom_map_add_points[_device]
//   radiusCleanup (:236-246)                     -> lom_map_radius_cleanup
//   getCloud / getCloudWithoutNormals /
//   getSparseCloudWithoutNormals (:112-162)      -> lom_map_export
//   setVoxelSize / setMaxPoints / size (:56-66, :248-251)
//
// The reference inserts serially; a voxel keeps the first max_points points in
// call order and that order is the nearest-neighbour tie-break order.  The
// insert below is data-parallel but a pure function of the input order:
// hash slots are claimed with a 64-bit CAS (which voxel gets which slot does
// not matter), creation order comes from a prefix scan over "first point of a
// new voxel" flags, and a point's position inside its voxel is its rank among
// the batch's points of that voxel by input index -- no result depends on the
// order in which atomics land.
//
// Built with -ffp-contract=off: the f32 index and distance expressions must
// round like the reference's (plain -O3 x86-64 build, no FMA contraction).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "grid_scan.hpp"
#include "lom_internal.hpp"
#include "pose_math.hpp"

namespace lom {

// ---------------------------------------------------------------------------
// error handling / buffers
// ---------------------------------------------------------------------------
static thread_local std::string g_create_error;

int set_error(lom_map *m, int code, const char *what, hipError_t e)
{
    std::string s = what ? what : "";
    if (e != hipSuccess) {
        s += ": ";
        s += hipGetErrorString(e);
    }
    if (m)
        m->last_error = s;
    else
        g_create_error = s;
    return code;
}

int ensure(lom_map *m, DeviceBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return LOM_OK;
    size_t nb = std::max(bytes, b.bytes + b.bytes / 2);
    nb = (nb + 255) & ~size_t(255);
    if (b.p) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        LOM_HIP(m, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, nb);
    if (e != hipSuccess) {
        b.p = nullptr;
        return set_error(m, LOM_ERR_OOM, "hipMalloc", e);
    }
    b.bytes = nb;
    return LOM_OK;
}

MapView view_of(const lom_map *self)
{
    const lom_map *m = self->parent ? self->parent : self;  // a scan context reads its keyframe's table and slabs
    MapView v;
    v.table = m->d_table;
    v.mask = m->cap - 1;
    v.shift = 64 - (uint32_t)__builtin_ctz(m->cap);
    v.pts = m->d_pts;
    v.nrm = m->d_nrm;
    v.K = m->K;
    v.voxel_size = m->voxel_size;
    int e = 0;
    const float inv = 1.0f / m->voxel_size;
    // power of two with a normal reciprocal: scaling by inv is exact
    v.inv_voxel_size = (std::frexp(m->voxel_size, &e) == 0.5f && std::isnormal(inv)) ? inv : 0.f;
    v.prune_slack = 1e-4f * m->voxel_size;
    return v;
}

enum {
    S_IN_XYZ = 0,
    S_IN_NRM,
    S_PT_SLOT,
    S_PT_POS,
    S_FLAG,
    S_RANK,
    S_BKT_CNT,
    S_BKT_HEAD,
    S_BKT_OFF,
    S_BKT_OLD,
    S_PT_OFF,
    S_PT_M,
    S_ITEMS,
    S_SCAN,
    S_DS_HEAD,
    S_MISC,
    S_ENT_ROW,  // bulk insert: row of every survivor
    S_HIST,     // bulk insert: [partition][block] counts
    S_PART      // bulk insert: partition sizes and starts
};


// ---------------------------------------------------------------------------
// exclusive prefix scan of uint32 (tile = 256 threads x 8 items)
// ---------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;

template <typename T>
__global__ __launch_bounds__(kThreads) void k_scan_tile(const T *__restrict__ in, T *__restrict__ out,
                                                        T *__restrict__ tile_sums, uint32_t n)
{
    __shared__ T s_wave[kThreads / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    T v[kScanItems];
    T sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        v[k] = (base + k < n) ? in[base + k] : T(0);
        sum += v[k];
    }
    // inclusive scan of per-thread sums inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    T wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        if (w < wave) wave_off += s_wave[w];
        total += s_wave[w];
    }
    T run = wave_off + inc - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

template <typename T>
__global__ void k_scan_add(T *__restrict__ out, const T *__restrict__ tile_prefix, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_prefix[i / kScanTile];
}

// out[i] = sum in[0..i), *d_total = sum of all.  tmp must hold scan_tmp_words(n) elements of T.
// T = uint64 scans two packed uint32 quantities at once (no carry while the low sum < 2^32).
template <typename T>
static int scan_exclusive(lom_map *m, const T *in, T *out, uint32_t n, T *d_total, T *tmp)
{
    const uint32_t nt = (n + kScanTile - 1) / kScanTile;
    if (nt <= 1) {
        hipLaunchKernelGGL(k_scan_tile<T>, dim3(1), dim3(kThreads), 0, m->stream, in, out, d_total, n);
        LOM_HIP(m, hipGetLastError());
        return LOM_OK;
    }
    T *sums = tmp, *prefix = tmp + nt;
    hipLaunchKernelGGL(k_scan_tile<T>, dim3(nt), dim3(kThreads), 0, m->stream, in, out, sums, n);
    LOM_HIP(m, hipGetLastError());
    int rc = scan_exclusive<T>(m, sums, prefix, nt, d_total, tmp + 2 * (size_t)nt);
    if (rc != LOM_OK) return rc;
    hipLaunchKernelGGL(k_scan_add<T>, dim3(blocks_for(n)), dim3(kThreads), 0, m->stream, out, prefix, n);
    LOM_HIP(m, hipGetLastError());
    return LOM_OK;
}

static size_t scan_tmp_words(uint32_t n)
{
    size_t w = 0;
    while (n > (uint32_t)kScanTile) {
        n = (n + kScanTile - 1) / kScanTile;
        w += 2 * (size_t)n;
    }
    return w + 16;
}

// ---------------------------------------------------------------------------
// table kernels
// ---------------------------------------------------------------------------
__global__ void k_table_init(Slot *table, uint32_t cap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        Slot s;
        s.key = kEmptyKey;
        s.count = 0;
        s.slab = kNoSlab;
        table[i] = s;
    }
}

__device__ inline uint32_t claim_slot(Slot *table, uint32_t mask, uint32_t shift, unsigned long long key)
{
    uint32_t h = hash_key(key, shift) & mask;
    for (;;) {
        // Look before the CAS: a slot that already shows this key needs no atomic (most points of a frame fall
        // into voxels the map already has, and an atomic is a round trip to the memory side).  A stale view
        // -- the slot still looks empty, or shows another key that is itself final -- only costs the CAS
        // (keys never change once set) or moves on to the next slot exactly as the CAS would.
        const unsigned long long seen = __hip_atomic_load(&table[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == key) return h;
        if (seen == kEmptyKey) {
            const unsigned long long prev = atomicCAS(&table[h].key, kEmptyKey, key);
            if (prev == kEmptyKey || prev == key) return h;
        }
        h = (h + 1) & mask;
    }
}

// rebuild the table from the slab arrays (after rehash / cleanup)
__global__ void k_rebuild(Slot *table, uint32_t mask, uint32_t shift, const unsigned long long *slab_key,
                          const uint32_t *slab_count, uint32_t n_vox)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_vox) return;
    const uint32_t c = slab_count[s];
    if (!c) return;  // a slab whose voxel a radius cleanup erased (k_cleanup_mark): the key is free again
    const uint32_t h = claim_slot(table, mask, shift, slab_key[s]);
    table[h].count = c;
    table[h].slab = s;
}

// ---------------------------------------------------------------------------
// insert kernels
// ---------------------------------------------------------------------------
__device__ inline const float *point_at(const char *base, size_t i, size_t stride)
{
    return reinterpret_cast<const float *>(base + i * stride);
}

// one 12-byte point as ONE memory instruction: a packed struct tells the compiler that the three floats are
// contiguous and 4-byte aligned, and it issues global_load / global_store_dwordx3 -- on the scattered slab
// writes that is one partial-line transaction per point instead of three
struct __attribute__((packed, aligned(4))) Point3 {
    float x, y, z;
};
__device__ __forceinline__ Point3 load3(const float *p) { return *reinterpret_cast<const Point3 *>(p); }
__device__ __forceinline__ void store3(float *p, Point3 v) { *reinterpret_cast<Point3 *>(p) = v; }

// per point: low word = 1 if it is the first point of a voxel seen for the first time (creation
// order = order of first appearance, voxel_grid.h:83-87), high word = size of the voxel's bucket
// if the point is the bucket's head.  One 64-bit exclusive scan then yields the new voxel's slab
// rank and the bucket's offset in the scratch list -- no same-address atomics.
__global__ void k_ins_heads(const Slot *table, uint32_t n, const uint32_t *pt_slot, const uint32_t *bkt_cnt,
                            const uint32_t *bkt_head, unsigned long long *flag64, uint32_t seq, const uint32_t *words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = pt_slot[i];
    unsigned long long f = 0;
    if (words[5] != seq && h != 0xFFFFFFFFu && bkt_head[h] == i) {
        f = (unsigned long long)bkt_cnt[h] << 32;
        if (table[h].slab == kNoSlab) f |= 1ull;  // voxel_grid.h:83 it == end()
    }
    flag64[i] = f;
}

__global__ void k_ins_assign(Slot *table, uint32_t n, const uint32_t *pt_slot, const uint32_t *bkt_head,
                             const unsigned long long *flag64, const unsigned long long *scan64,
                             const uint32_t *n_vox_dev, unsigned long long *slab_key, uint32_t *bkt_off,
                             uint32_t *bkt_old, uint32_t seq, const uint32_t *words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || words[5] == seq) return;
    const uint32_t h = pt_slot[i];
    if (h == 0xFFFFFFFFu || bkt_head[h] != i) return;
    const uint32_t n_vox_before = *n_vox_dev;  // device-side voxel counter (bumped by k_ins_place2)
    const unsigned long long sc = scan64[i];
    bkt_off[h] = (uint32_t)(sc >> 32);
    bkt_old[h] = (flag64[i] & 1ull) ? 0u : table[h].count;
    if (flag64[i] & 1ull) {
        const uint32_t slab = n_vox_before + (uint32_t)sc;
        table[h].slab = slab;
        slab_key[slab] = table[h].key;
    }
}

__global__ void k_set_word(uint32_t *w, uint32_t v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *w = v;
}

// ---------------------------------------------------------------------------
// fused down-sampler: VoxelGrid(voxel, 1).addCloud(cloud) followed by getCloud() /
// getCloudWithoutNormals() (the reference's idiom, lidar_odometry.cpp:37-38,42,46-47,50) keeps
// the FIRST point of every voxel in input order and returns them in order of first appearance.
// That is: claim a slot per voxel, take the minimum input index per slot, keep the points whose
// index is that minimum, compact them by a scan over the input.  No payload slabs are touched.
// ---------------------------------------------------------------------------
// n_dev (optional): the number of input points when only the device knows it (n is then its upper bound)
__global__ void k_ds_claim(Slot *table, uint32_t mask, uint32_t shift, const char *xyz, size_t stride, uint32_t n,
                           const uint32_t *n_dev, float vs, uint32_t *pt_slot, uint32_t *head, uint32_t seq, uint32_t *bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    const float *p = point_at(xyz, i, stride);
    int ix = 0, iy = 0, iz = 0;
    if (!voxel_index(p[0], vs, ix) || !voxel_index(p[1], vs, iy) || !voxel_index(p[2], vs, iz)) {
        __hip_atomic_store(bad, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // out of range / not finite: the call fails
        pt_slot[i] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t h = claim_slot(table, mask, shift, pack_key(ix, iy, iz));
    pt_slot[i] = h;
    atomicMin(&head[h], i);
}

__global__ void k_ds_flag(uint32_t n, const uint32_t *pt_slot, const uint32_t *head, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t h = pt_slot[i];
        flag[i] = (h != 0xFFFFFFFFu && head[h] == i) ? 1u : 0u;
    }
}

// CloudTransformer::transform / transformWithNormals (utils/cloud_transform.h:43-97) on the device:
// the same f32 expressions as lom_transform_points, R and t prepared on the host
struct RigidArgs {
    float R[9], t[3];
};
__global__ void k_transform(const char *xyz, const char *nrm, size_t stride, uint32_t n, RigidArgs A, float *out_xyz,
                            float *out_nrm)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = point_at(xyz, i, stride);
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    float *o = out_xyz + (size_t)i * 3;
    o[0] = (A.R[0] * p0 + (A.R[1] * p1 + A.R[2] * p2)) + A.t[0];
    o[1] = (A.R[3] * p0 + (A.R[4] * p1 + A.R[5] * p2)) + A.t[1];
    o[2] = (A.R[6] * p0 + (A.R[7] * p1 + A.R[8] * p2)) + A.t[2];
    if (nrm && out_nrm) {
        const float *q = point_at(nrm, i, stride);
        const float n0 = q[0], n1 = q[1], n2 = q[2];
        float *no = out_nrm + (size_t)i * 3;
        no[0] = A.R[0] * n0 + (A.R[1] * n1 + A.R[2] * n2);
        no[1] = A.R[3] * n0 + (A.R[4] * n1 + A.R[5] * n2);
        no[2] = A.R[6] * n0 + (A.R[7] * n1 + A.R[8] * n2);
    }
}

__global__ void k_ds_write(uint32_t n, const uint32_t *flag, const uint32_t *rank, const char *xyz, const char *nrm,
                           size_t stride, float *out_xyz, float *out_nrm, Slot *table, const uint32_t *pt_slot,
                           uint32_t *head)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    {  // the workspace goes back to rest: the kept point frees its voxel's slot and head word
        const uint32_t h = pt_slot[i];
        Slot e;
        e.key = kEmptyKey;
        e.count = 0;
        e.slab = kNoSlab;
        table[h] = e;
        head[h] = 0xFFFFFFFFu;
    }
    const size_t d = (size_t)rank[i] * 3;
    const float *p = point_at(xyz, i, stride);
    out_xyz[d] = p[0];
    out_xyz[d + 1] = p[1];
    out_xyz[d + 2] = p[2];
    if (out_nrm) {
        if (nrm) {
            const float *q = point_at(nrm, i, stride);
            out_nrm[d] = q[0];
            out_nrm[d + 1] = q[1];
            out_nrm[d + 2] = q[2];
        } else {
            out_nrm[d] = out_nrm[d + 1] = out_nrm[d + 2] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------
// Single-pass variants for per-frame sizes (n <= kOnePassMax): one element per thread, at most 256
// workgroups of 256, all resident at once.  What used to be {flag kernel, 1-3 scan launches, consumer
// kernel} is one kernel: block-local scan, then every workgroup publishes its total as a tagged 8-byte
// word {call sequence number, value} (one store; no reset between calls, the sequence number tells
// fresh from stale) and adds up the totals of the workgroups before it -- <= 255 words, one per
// thread, fixed order, so the prefix is deterministic.  Every wait is bounded (s_memrealtime); a
// workgroup that gives up writes the call's sequence number into the error word and carries on
// with a zero prefix: the host sees the error at its next read-back.
//
// Per-slot scratch (batch count, earliest input index, down-sampler head) is kept "at rest" between
// calls -- zero / 0xFFFFFFFF everywhere -- by the one thread per voxel that consumed it, so no call
// pays a memset proportional to the table capacity.
// ---------------------------------------------------------------------------
// ---- down-sampler, two kernels ------------------------------------------------------------------
// k_ds_claim (above) leaves pt_slot[] and head[]; this kernel keeps the first point of every voxel in
// order of first appearance and puts the workspace back to rest: the head point of a voxel frees its
// table slot and its head word, so neither a table re-initialisation nor a memset follows.
template <int kItems>  // consecutive points per thread: 1 up to 65536 points, 4 up to 262144
__global__ __launch_bounds__(kThreads) void k_ds_emit(Slot *table, uint32_t n, const uint32_t *n_dev,
                                                      const uint32_t *__restrict__ pt_slot, uint32_t *head, const char *xyz,
                                                      const char *nrm, size_t stride, float *out_xyz, float *out_nrm,
                                                      Granule *agg, uint32_t seq, uint32_t *words, uint32_t test_fail_from)
{
    __shared__ unsigned long long s_w[8];
    const uint32_t base = (blockIdx.x * kThreads + threadIdx.x) * kItems;
    if (n_dev) n = min(n, *n_dev);
    uint32_t h[kItems];
    bool keep[kItems];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t i = base + k;
        h[k] = kInvalidSlot;
        keep[k] = false;
        if (i < n) {
            h[k] = pt_slot[i];
            keep[k] = h[k] != kInvalidSlot && head[h[k]] == i;
        }
        mine += keep[k] ? 1u : 0u;
    }
    unsigned long long total;
    const unsigned long long excl = block_scan64(mine, s_w, total);
    bool gave_up;
    const unsigned long long before = grid_prefix64(total, agg, seq, words + 7, s_w, gave_up, test_fail_from);
    uint32_t at = (uint32_t)(before + excl);
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        if (!keep[k]) continue;
        const uint32_t i = base + k;
        const size_t d = (size_t)at * 3;
        at++;
        if (!gave_up) {  // without a prefix there is no place to write to; the workspace still goes back to rest below
            const float *p = point_at(xyz, i, stride);
            out_xyz[d] = p[0];
            out_xyz[d + 1] = p[1];
            out_xyz[d + 2] = p[2];
            if (out_nrm) {
                if (nrm) {
                    const float *q = point_at(nrm, i, stride);
                    out_nrm[d] = q[0];
                    out_nrm[d + 1] = q[1];
                    out_nrm[d + 2] = q[2];
                } else {
                    out_nrm[d] = out_nrm[d + 1] = out_nrm[d + 2] = 0.f;
                }
            }
        }
        Slot e;
        e.key = kEmptyKey;
        e.count = 0;
        e.slab = kNoSlab;
        table[h[k]] = e;
        head[h[k]] = 0xFFFFFFFFu;
    }
    // voxels kept; a grid that gave up reports none (whoever consumes the count on the device finds an empty cloud)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) words[4] = gave_up ? 0u : (uint32_t)(before + total);
}

// ---- insert, four kernels ----------------------------------------------------------------------
// 1. k_ins_claim2   slot per point (CAS), arrival position in the voxel's bucket, earliest input index;
//                   range check folded in (a call with a bad point inserts nothing: the later kernels
//                   see the call's sequence number in the error word and only put the scratch to rest)
// 2. k_ins_assign2  one 64-bit scan: creation order of the new voxels (low word) and bucket offsets
//                   (high word); the head point of a voxel assigns slab, offset and the old count
// 3. k_ins_scatter2 bucket lists; every point takes a private copy of its bucket's size and offset
// 4. k_ins_place2   rank by input index inside the voxel (= insertion order, voxel_grid.h:86-90), store
//                   the first K - count; the head point publishes the new count and resets the scratch
__global__ __launch_bounds__(kThreads) void k_ins_claim2(Slot *table, uint32_t mask, uint32_t shift, const char *xyz,
                                                         size_t stride, uint32_t n, float vs, uint32_t *pt_slot,
                                                         uint32_t *pt_pos, uint32_t *bkt_cnt, uint32_t *bkt_head,
                                                         uint32_t seq, uint32_t *words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = point_at(xyz, i, stride);
    int ix = 0, iy = 0, iz = 0;
    if (!voxel_index(p[0], vs, ix) || !voxel_index(p[1], vs, iy) || !voxel_index(p[2], vs, iz)) {
        pt_slot[i] = kInvalidSlot;
        __hip_atomic_store(words + 5, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // LOM_ERR_RANGE for this call
        return;
    }
    const uint32_t h = claim_slot(table, mask, shift, pack_key(ix, iy, iz));
    pt_slot[i] = h;
    pt_pos[i] = atomicAdd(&bkt_cnt[h], 1u);  // arbitrary order; fixed up by rank in k_ins_place2
    atomicMin(&bkt_head[h], i);              // earliest input index touching the voxel
}

__global__ __launch_bounds__(kThreads) void k_ins_assign2(Slot *table, uint32_t n, const uint32_t *__restrict__ pt_slot,
                                                          const uint32_t *__restrict__ bkt_cnt,
                                                          const uint32_t *__restrict__ bkt_head, uint32_t *bkt_off,
                                                          uint32_t *bkt_old, const uint32_t *n_vox_dev,
                                                          unsigned long long *slab_key, Granule *agg, uint32_t seq,
                                                          uint32_t *words, uint32_t test_fail_from)
{
    __shared__ unsigned long long s_w[8];
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    const bool failed = words[5] == seq;  // a point of this call was out of range: nothing is inserted
    uint32_t h = kInvalidSlot;
    bool is_head = false, is_new = false;
    uint32_t old_count = 0, m = 0;
    if (i < n && !failed) {
        h = pt_slot[i];
        is_head = bkt_head[h] == i;
        if (is_head) {
            const Slot s = table[h];
            is_new = s.slab == kNoSlab;  // voxel_grid.h:83 it == end()
            old_count = is_new ? 0u : s.count;
            m = bkt_cnt[h];
        }
    }
    unsigned long long total;
    const unsigned long long v = ((unsigned long long)m << 32) | (is_new ? 1ull : 0ull);
    const unsigned long long excl = block_scan64(v, s_w, total);
    bool gave_up;
    const unsigned long long before = grid_prefix64(total, agg, seq, words + 7, s_w, gave_up, test_fail_from);
    // A workgroup without a prefix assigns nothing; what the others assigned before the give-up is taken back by
    // k_ins_place2 (slab ids at or beyond the voxel counter, which such a call does not advance).
    if (is_head && !gave_up) {
        const unsigned long long at = before + excl;
        bkt_off[h] = (uint32_t)(at >> 32);
        bkt_old[h] = old_count;
        if (is_new) {
            const uint32_t slab = *n_vox_dev + (uint32_t)at;  // creation order = order of first appearance
            table[h].slab = slab;
            slab_key[slab] = table[h].key;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && !gave_up) words[0] = (uint32_t)(before + total);  // new voxels of this call
}

__global__ __launch_bounds__(kThreads) void k_ins_scatter2(uint32_t n, const uint32_t *__restrict__ pt_slot,
                                                           const uint32_t *__restrict__ pt_pos,
                                                           const uint32_t *__restrict__ bkt_off,
                                                           const uint32_t *__restrict__ bkt_cnt, uint32_t *items,
                                                           uint32_t *pt_off, uint32_t *pt_m, uint32_t seq,
                                                           const uint32_t *words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || words[5] == seq || words[7] == seq) return;  // a call that failed (range / grid give-up) inserts nothing
    const uint32_t h = pt_slot[i];
    const uint32_t off = bkt_off[h];
    items[off + pt_pos[i]] = i;
    pt_off[i] = off;
    pt_m[i] = bkt_cnt[h];
}

__global__ __launch_bounds__(kThreads) void k_ins_place2(Slot *table, uint32_t n, const uint32_t *__restrict__ pt_slot,
                                                         const uint32_t *__restrict__ pt_off,
                                                         const uint32_t *__restrict__ pt_m, uint32_t *bkt_cnt,
                                                         uint32_t *bkt_head, const uint32_t *__restrict__ bkt_old,
                                                         const uint32_t *__restrict__ items, const char *xyz,
                                                         const char *nrm, size_t stride, uint32_t K, uint32_t cap_points,
                                                         float *pts, float *nrm_out, uint32_t *slab_count,
                                                         uint32_t *n_vox_dev, uint32_t seq, const uint32_t *words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = pt_slot[i];
    if (h == kInvalidSlot) return;
    const bool is_head = bkt_head[h] == i;  // only this thread resets the word, and only after this read
    if (words[7] == seq) {
        // the scan of k_ins_assign2 gave up part-way: take back the slab ids the workgroups before the give-up
        // handed to NEW voxels (at or beyond the voxel counter, which this call does not advance), so that the
        // table is what it was before the call -- apart from claimed keys without a voxel, as after a range error
        if (is_head) {
            const uint32_t slab = table[h].slab;
            if (slab != kNoSlab && slab >= *n_vox_dev) table[h].slab = kNoSlab;
        }
    } else if (words[5] != seq) {
        const uint32_t old = bkt_old[h];
        const uint32_t slab = table[h].slab;
        const uint32_t m = pt_m[i];
        // voxel_grid.h:86,89-90: a voxel takes points while size() < max_points_ (cap_points; the row stride K is at least
        // that, and a voxel filled under a larger max_points_ keeps what it holds)
        const uint32_t room = cap_points > old ? cap_points - old : 0u;
        if (room) {
            const uint32_t *it = items + pt_off[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m && rank < room; j++) rank += it[j] < i;
            if (rank < room) {  // voxel_grid.h:86,89-90: append while size() < max_points_, in input order
                const size_t dst = ((size_t)slab * K + old + rank) * 3;
                const Point3 pv = load3(point_at(xyz, i, stride));
                Point3 nv = {0.f, 0.f, 0.f};  // voxel_grid.h:103,107: no normals -> (0, 0, 0)
                if (nrm) nv = load3(point_at(nrm, i, stride));
                store3(pts + dst, pv);
                store3(nrm_out + dst, nv);
            }
        }
        if (is_head) {
            const uint32_t nc = old + (m < room ? m : room);
            table[h].count = nc;
            slab_count[slab] = nc;
        }
        if (i == 0) *n_vox_dev += words[0];  // point 0 is always the head of its voxel's bucket... and exists once
    }
    if (is_head) {  // scratch back to rest
        bkt_cnt[h] = 0u;
        bkt_head[h] = 0xFFFFFFFFu;
    }
}

// ---- bulk insert (batches above kOnePassMax points): partitions of the table, grouped in LDS -------------
// The four kernels above pay three scattered device-scope atomics per point (slot CAS, bucket count, earliest
// index) and write every stored point as a lone 12-byte transaction.  A bulk batch -- the map build -- goes
// through a partition pass instead, so that all points of one voxel meet in ONE workgroup and everything per
// voxel happens in LDS:
//   1. k_bi_claim    slot per point (look, CAS only where the slot is still empty; the looks of a thread's
//                    points are in flight together); LDS histogram of the points over the partitions
//                    (partition = a contiguous segment of the table: the top bits of the slot), one column of
//                    the [partition][block] count matrix per workgroup
//   2. k_bi_colscan  one wave per partition: prefix of its counts over the blocks (blocks are in input order,
//                    so a partition's list is ordered by block); partition sizes; verdict (a partition above
//                    the LDS budget of step 4 sends the whole call to the four-kernel path: nothing written)
//   3. k_bi_scatter  (slot, input index) pairs to their partition's list (LDS cursor per partition)
//   4. k_bi_group    one workgroup per partition, all in LDS: points -> voxels (hash on the slot), bucket
//                    sizes, room left in each voxel (voxel_grid.h:86), rank of every point inside its voxel
//                    by input index (= insertion order), the first `room` survive; the survivors of a voxel
//                    are written out side by side, in rank order; a NEW voxel's first point sets its bit in a
//                    bitmap over the input indices
//   5. k_bi_flagscan one workgroup: prefix of the bitmap's population counts -- creation order = order of
//                    first appearance (voxel_grid.h:83-87)
//   6. k_bi_place    one thread per survivor: consecutive threads write consecutive rows of a slab --
//                    coalesced slab writes, nothing but the stored rows is written
// No kernel waits for another workgroup, so there is no give-up path; a range error (step 1) or an oversized
// partition (step 2) is known before anything is written except claimed keys, which the table tolerates
// (slab == kNoSlab, as after a range error of the four-kernel path).
constexpr int kBiThreads = 1024;         // k_bi_claim, k_bi_scatter, k_bi_flagscan
constexpr uint32_t kBiPartMax = 1024;    // points per partition k_bi_group can hold in LDS (its larger shape)
constexpr uint32_t kBiMaxParts = 16384;  // LDS histogram of k_bi_claim: 64 KB
constexpr uint32_t kBiMaxPoints = 4u << 20;
constexpr uint32_t kBiDropped = 0xFFFFFFFFu;
constexpr uint32_t kBiNewBit = 0x80000000u;

// what k_bi_claim learns about a point's voxel on the way: the points it holds (kBiNewBit: the map does not have the
// voxel yet -- no slab; voxel_grid.h:83 it == end()).  Nothing in this kernel changes counts or slabs, so a look taken
// at any time during it holds for the whole call.
template <int kPpt>
__global__ __launch_bounds__(kBiThreads) void k_bi_claim(Slot *table, uint32_t mask, uint32_t shift, const char *xyz,
                                                         size_t stride, uint32_t n, float vs, uint2 *pt_info,
                                                         uint32_t *flag_bits, uint32_t *hist, uint32_t n_parts,
                                                         uint32_t part_shift, const uint32_t *n_vox_dev, uint32_t seq,
                                                         uint32_t *words)
{
    extern __shared__ uint32_t s_hist[];
    for (uint32_t b = threadIdx.x; b < n_parts; b += kBiThreads) s_hist[b] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) words[9] = *n_vox_dev;  // voxel count before this call (k_bi_place)
    const uint32_t first = blockIdx.x * kPpt * kBiThreads;
    if (threadIdx.x < kPpt * kBiThreads / 32) {  // this block's words of the bitmap of first appearances
        const uint32_t w = first / 32 + threadIdx.x;
        if (w < (n + 31) / 32) flag_bits[w] = 0u;
    }
    __syncthreads();
    // the chain per point is point -> slot look -> (CAS) -> (next slot): all looks of this thread's points are issued
    // before the first is consumed (the table is far larger than the caches while it is being built).  A look is the
    // whole 16-byte slot: key, count and slab together.
    unsigned long long key[kPpt];
    typedef uint32_t SlotWords __attribute__((ext_vector_type(4)));
    SlotWords seen[kPpt];
    uint32_t h[kPpt];
    bool ok[kPpt];
#pragma unroll
    for (int k = 0; k < kPpt; k++) {
        const uint32_t i = first + k * kBiThreads + threadIdx.x;
        ok[k] = false;
        key[k] = 0;
        h[k] = 0;
        if (i < n) {
            const Point3 p = load3(point_at(xyz, i, stride));
            int ix = 0, iy = 0, iz = 0;
            if (voxel_index(p.x, vs, ix) && voxel_index(p.y, vs, iy) && voxel_index(p.z, vs, iz)) {
                ok[k] = true;
                key[k] = pack_key(ix, iy, iz);
                h[k] = hash_key(key[k], shift) & mask;
            } else {
                pt_info[i] = make_uint2(kInvalidSlot, 0u);
                __hip_atomic_store(words + 5, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // LOM_ERR_RANGE for this call
            }
        }
    }
    const SlotWords *slots = reinterpret_cast<const SlotWords *>(table);
#pragma unroll
    for (int k = 0; k < kPpt; k++) seen[k] = ok[k] ? slots[h[k]] : SlotWords{0u, 0u, 0u, 0u};
    // first round of the claims, all of this thread's compare-and-swaps in flight together: a slot that looked empty
    unsigned long long prev[kPpt];
#pragma unroll
    for (int k = 0; k < kPpt; k++) {
        const unsigned long long sk = ((unsigned long long)seen[k].y << 32) | seen[k].x;
        prev[k] = (ok[k] && sk == kEmptyKey) ? atomicCAS(&table[h[k]].key, kEmptyKey, key[k]) : sk;
    }
#pragma unroll
    for (int k = 0; k < kPpt; k++) {
        if (!ok[k]) continue;
        uint32_t s = h[k];
        SlotWords sn = seen[k];
        const unsigned long long sk0 = ((unsigned long long)sn.y << 32) | sn.x;
        uint32_t oldw;
        if (sk0 == kEmptyKey && (prev[k] == kEmptyKey || prev[k] == key[k])) {
            oldw = kBiNewBit;  // this call's own claim, now or a moment ago: no voxel yet
        } else {
            if (sk0 == kEmptyKey) {  // somebody else's key arrived in between: on to the next slot
                s = (s + 1) & mask;
                sn = slots[s];
            }
            for (;;) {  // claim_slot from here on
                const unsigned long long sk = ((unsigned long long)sn.y << 32) | sn.x;
                if (sk == key[k]) {
                    oldw = sn.w == kNoSlab ? kBiNewBit : sn.z;
                    break;
                }
                if (sk == kEmptyKey) {
                    const unsigned long long pv = atomicCAS(&table[s].key, kEmptyKey, key[k]);
                    if (pv == kEmptyKey || pv == key[k]) {
                        oldw = kBiNewBit;
                        break;
                    }
                }
                s = (s + 1) & mask;
                sn = slots[s];
            }
        }
        pt_info[first + k * kBiThreads + threadIdx.x] = make_uint2(s, oldw);
        atomicAdd(&s_hist[s >> part_shift], 1u);
    }
    __syncthreads();
    uint32_t *row = hist + (size_t)blockIdx.x * n_parts;  // [block][partition]
    for (uint32_t b = threadIdx.x; b < n_parts; b += kBiThreads) row[b] = s_hist[b];
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// 64 partitions per workgroup (one 256-byte piece of every block's row), the blocks in sixteen slices, one per wave:
// column sums of the slices, then the running prefixes written over the counts
__global__ __launch_bounds__(kBiThreads) void k_bi_colscan(uint32_t *hist, uint32_t n_parts, uint32_t n_blk,
                                                           uint32_t *part_total, uint32_t part_max, uint32_t seq,
                                                           uint32_t *words)
{
    constexpr uint32_t kSlices = kBiThreads / 64;
    __shared__ uint32_t s_slice[kSlices][64];
    if (words[5] == seq) return;
    const uint32_t lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const uint32_t part = blockIdx.x * 64 + lane;
    const bool live = part < n_parts;
    const uint32_t per = (n_blk + kSlices - 1) / kSlices;
    const uint32_t b0 = min(slice * per, n_blk), b1 = min(b0 + per, n_blk);
    uint32_t *col = hist + part;
    uint32_t sum = 0;
    if (live) {
#pragma unroll 8
        for (uint32_t b = b0; b < b1; b++) sum += col[(size_t)b * n_parts];
    }
    s_slice[slice][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSlices; k++) {
        if (k < slice) run += s_slice[k][lane];
        total += s_slice[k][lane];
    }
    if (live) {
#pragma unroll 8
        for (uint32_t b = b0; b < b1; b++) {
            const uint32_t v = col[(size_t)b * n_parts];
            col[(size_t)b * n_parts] = run;
            run += v;
        }
    }
    if (slice == 0 && live) {
        part_total[part] = total;
        if (total > part_max) __hip_atomic_store(words + 8, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int kPpt>
__global__ __launch_bounds__(kBiThreads) void k_bi_scatter(uint32_t n, const uint2 *__restrict__ pt_info,
                                                           const uint32_t *__restrict__ hist, uint32_t n_parts,
                                                           uint32_t part_shift, const uint32_t *__restrict__ part_total,
                                                           uint32_t *part_start, uint4 *part_rec, uint32_t seq,
                                                           const uint32_t *words)
{
    extern __shared__ uint32_t s_cur[];  // [n_parts] write cursor of this block in every partition's list
    __shared__ uint32_t s_w[kBiThreads / 64];
    if (words[5] == seq || words[8] == seq) return;
    const uint32_t first = blockIdx.x * kPpt * kBiThreads;
    uint2 info[kPpt];
#pragma unroll
    for (int k = 0; k < kPpt; k++) {
        const uint32_t i = first + k * kBiThreads + threadIdx.x;
        info[k] = i < n ? pt_info[i] : make_uint2(0u, 0u);
    }
    // start of every partition's list = exclusive scan of the partition sizes (every block redoes it: <= 16384 values)
    const uint32_t per = (n_parts + kBiThreads - 1) / kBiThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b0 + per && b < n_parts; b++) sum += part_total[b];
    const uint32_t inc = wave_inclusive_scan(sum);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wave; w++) run += s_w[w];
    const uint32_t *row = hist + (size_t)blockIdx.x * n_parts;  // this block's offsets inside the partitions' lists
    for (uint32_t b = b0; b < b0 + per && b < n_parts; b++) {
        s_cur[b] = run + row[b];
        if (blockIdx.x == 0) part_start[b] = run;
        run += part_total[b];
    }
    if (blockIdx.x == 0 && threadIdx.x == kBiThreads - 1) part_start[n_parts] = run;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPpt; k++) {
        const uint32_t i = first + k * kBiThreads + threadIdx.x;
        if (i >= n) break;
        const uint32_t pos = atomicAdd(&s_cur[info[k].x >> part_shift], 1u);
        part_rec[pos] = make_uint4(info[k].x, i, info[k].y, 0u);  // slot, input index, what the voxel held
    }
}

// One workgroup per partition.  What it leaves behind, at the partition's positions of three N-sized arrays:
// the survivors (points that are stored) first, grouped by voxel and in rank order inside a voxel --
// ent_idx = input index, ent_row = row inside the slab, ent_w = the slot of a voxel that exists (| kBiFirstBit for the
// first survivor, which settles the slab's count), or for a voxel this call creates kBiNewBit | its slot (first
// survivor) / kBiNewBit | the input index of its first point (the others) -- and ent_row = kBiDropped for the rest.
// table[h].count is final here; slab ids of new voxels follow from the bitmap of first appearances (k_bi_flagscan,
// k_bi_place).  Nothing is read but the partition's own records: what the voxels held came along from k_bi_claim.
constexpr uint32_t kBiFirstBit = 0x40000000u;
template <int kPartMax>  // points the partition may hold: 512 (20 KB of LDS, eight workgroups per CU) or 1024
__global__ __launch_bounds__(kThreads) void k_bi_group(Slot *table, const uint32_t *__restrict__ part_start,
                                                       const uint4 *__restrict__ part_rec, uint32_t cap_points,
                                                       uint32_t *flag_bits, uint32_t *ent_idx, uint32_t *ent_w,
                                                       uint32_t *ent_row, uint32_t seq, const uint32_t *words)
{
    constexpr uint32_t kEntries = 2 * kPartMax;
    __shared__ uint32_t s_key[kEntries];   // slot of the voxel (0xFFFFFFFF: free)
    __shared__ uint32_t s_cnt[kEntries];   // points of this call in the voxel
    __shared__ uint32_t s_old[kEntries];   // stored points before this call | kBiNewBit
    __shared__ uint16_t s_start[kEntries]; // first position of the voxel's bucket in s_grp
    __shared__ uint16_t s_sst[kEntries];   // first position of the voxel's survivors in the output
    __shared__ uint32_t s_grp[kPartMax];
    __shared__ uint32_t s_w[kThreads / 64];
    if (words[5] == seq || words[8] == seq) return;
    const uint32_t base = part_start[blockIdx.x];
    const uint32_t P = part_start[blockIdx.x + 1] - base;
    if (P == 0) return;
    constexpr int kItems = kPartMax / kThreads;
    uint4 pr[kItems];
#pragma unroll
    for (int k = 0; k < kItems; k++) {  // the partition's records, on their way while the tables are cleared
        const uint32_t j = k * kThreads + threadIdx.x;
        pr[k] = j < P ? part_rec[base + j] : make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t E = 256;  // entries: a power of two >= 2 P
    while (E < 2 * P) E <<= 1;
    const uint32_t ebits = (uint32_t)__builtin_ctz(E);
    for (uint32_t e = threadIdx.x; e < E; e += kThreads) {
        s_key[e] = 0xFFFFFFFFu;
        s_cnt[e] = 0u;
    }
    __syncthreads();
    uint32_t my_e[kItems], my_a[kItems];
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t j = k * kThreads + threadIdx.x;
        my_e[k] = 0;
        my_a[k] = 0;
        if (j < P) {
            const uint32_t h = pr[k].x;
            uint32_t e = (h * 0x9E3779B1u) >> (32 - ebits);
            for (;;) {
                const uint32_t prev = atomicCAS(&s_key[e], 0xFFFFFFFFu, h);
                if (prev == 0xFFFFFFFFu || prev == h) break;
                e = (e + 1) & (E - 1);
            }
            my_e[k] = e;
            my_a[k] = atomicAdd(&s_cnt[e], 1u);
            if (my_a[k] == 0) s_old[e] = pr[k].z;  // the same for all points of the voxel
        }
    }
    __syncthreads();
    // per voxel: how much room is left (voxel_grid.h:86: while size() < max_points_), the new count.
    // Thread t takes the entries [t * ept, (t + 1) * ept), so that the scan below runs over entries in order.
    const uint32_t ept = E / kThreads;
    uint32_t packed = 0;  // bucket sizes << 16 | survivors of this thread's entries
    for (uint32_t e = threadIdx.x * ept; e < (threadIdx.x + 1) * ept; e++) {
        const uint32_t h = s_key[e];
        if (h == 0xFFFFFFFFu) continue;
        const uint32_t old = s_old[e] & ~kBiNewBit;
        const uint32_t room = cap_points > old ? cap_points - old : 0u;
        const uint32_t c = s_cnt[e];
        const uint32_t st = c < room ? c : room;
        if (st) table[h].count = old + st;
        packed += (c << 16) | st;
    }
    const uint32_t inc = wave_inclusive_scan(packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t run = inc - packed, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        if (w < wave) run += s_w[w];
        total += s_w[w];
    }
    for (uint32_t e = threadIdx.x * ept; e < (threadIdx.x + 1) * ept; e++) {
        if (s_key[e] == 0xFFFFFFFFu) continue;
        const uint32_t old = s_old[e] & ~kBiNewBit;
        const uint32_t room = cap_points > old ? cap_points - old : 0u;
        const uint32_t c = s_cnt[e];
        s_start[e] = (uint16_t)(run >> 16);
        s_sst[e] = (uint16_t)(run & 0xFFFFu);
        run += (c << 16) | (c < room ? c : room);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t j = k * kThreads + threadIdx.x;
        if (j < P) s_grp[s_start[my_e[k]] + my_a[k]] = pr[k].y;
    }
    __syncthreads();
    const uint32_t survivors = total & 0xFFFFu;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t j = k * kThreads + threadIdx.x;
        if (j >= P) continue;
        const uint32_t e = my_e[k];
        const uint32_t i = pr[k].y;
        const uint32_t ow = pr[k].z;
        const uint32_t old = ow & ~kBiNewBit;
        const uint32_t room = cap_points > old ? cap_points - old : 0u;
        const uint32_t *g = s_grp + s_start[e];
        const uint32_t m = s_cnt[e];
        uint32_t rank = 0, head = i;  // voxel_grid.h:86,89-90: append in input order while size() < max_points_
        for (uint32_t q = 0; q < m && rank < room; q++) {
            const uint32_t o = g[q];
            rank += o < i;
            head = o < head ? o : head;
        }
        if (rank < room) {
            const uint32_t o = base + s_sst[e] + rank;
            ent_idx[o] = i;
            ent_row[o] = old + rank;
            if (ow & kBiNewBit) {
                // (a survivor has seen its whole bucket -- the loop ends early only once `room` smaller indices were
                // counted, and then the point is no survivor -- so `head` is the bucket's minimum)
                ent_w[o] = kBiNewBit | (rank == 0 ? pr[k].x : head);
                if (rank == 0) atomicOr(&flag_bits[i >> 5], 1u << (i & 31));  // first appearance of a voxel the map does not have yet
            } else {
                ent_w[o] = pr[k].x | (rank == 0 ? kBiFirstBit : 0u);
            }
        }
    }
    for (uint32_t j = survivors + threadIdx.x; j < P; j += kThreads) ent_row[base + j] = kBiDropped;
}

// Prefix of the population counts of the bitmap's words, one launch: a workgroup per tile of 1024 words (32,768 points)
// leaves the prefix inside its tile and the tile's total; the workgroup that finishes LAST (a counter tells it; nobody
// waits for anybody) turns the <= 128 totals into the tiles' own prefix.
// rank of a new voxel = tile_prefix[i / 32768] + word_prefix[i / 32] + the bits below bit i % 32 of its word
constexpr uint32_t kBiTileWords = 1024;
__global__ __launch_bounds__(kThreads) void k_bi_flagscan(const uint32_t *__restrict__ flag_bits, uint32_t n_words,
                                                          uint32_t *word_prefix, uint32_t *tile_total,
                                                          uint32_t *tile_prefix, uint32_t *done, uint32_t *total_out,
                                                          uint32_t seq, const uint32_t *words)
{
    __shared__ uint32_t s_w[kThreads / 64];
    __shared__ uint32_t s_last;
    if (words[5] == seq || words[8] == seq) return;
    const uint32_t w0 = blockIdx.x * kBiTileWords + threadIdx.x * 4;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        c[k] = w0 + k < n_words ? (uint32_t)__popc(flag_bits[w0 + k]) : 0u;
        sum += c[k];
    }
    const uint32_t inc = wave_inclusive_scan(sum);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        if (w < wave) run += s_w[w];
        total += s_w[w];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (w0 + k < n_words) word_prefix[w0 + k] = run;
        run += c[k];
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(tile_total + blockIdx.x, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < gridDim.x; t0 += 64) {  // one wave, 64 tiles at a time
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < gridDim.x ? __hip_atomic_load(tile_total + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t in2 = wave_inclusive_scan(v);
        if (t < gridDim.x) tile_prefix[t] = carry + in2 - v;
        carry += __shfl(in2, 63, 64);
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // at rest for the next call
    }
}

__global__ __launch_bounds__(kThreads) void k_bi_place(Slot *table, uint32_t n, const uint32_t *__restrict__ ent_idx,
                                                       const uint32_t *__restrict__ ent_w,
                                                       const uint32_t *__restrict__ ent_row,
                                                       const uint32_t *__restrict__ flag_bits,
                                                       const uint32_t *__restrict__ word_prefix,
                                                       const uint32_t *__restrict__ tile_prefix,
                                                       const uint32_t *__restrict__ new_total, const char *xyz,
                                                       const char *nrm, size_t stride, uint32_t K, float *pts,
                                                       float *nrm_out, unsigned long long *slab_key, uint32_t *slab_count,
                                                       uint32_t *n_vox_dev, uint32_t seq, const uint32_t *words)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || words[5] == seq || words[8] == seq) return;
    const uint32_t n_vox_before = words[9];
    if (j == 0) *n_vox_dev = n_vox_before + *new_total;
    const uint32_t row = ent_row[j];
    if (row == kBiDropped) return;
    const uint32_t i = ent_idx[j];
    const uint32_t w = ent_w[j];
    const Point3 pv = load3(point_at(xyz, i, stride));
    Point3 nv = {0.f, 0.f, 0.f};  // voxel_grid.h:103,107: no normals -> (0, 0, 0)
    if (nrm) nv = load3(point_at(nrm, i, stride));
    uint32_t slab;
    if (w & kBiNewBit) {
        const uint32_t head = row == 0 ? i : (w & ~kBiNewBit);  // a new voxel's rows start at 0: row 0 is its first point
        // creation order = order of first appearance: the new voxels' first points before this one
        slab = n_vox_before + tile_prefix[head / (32 * kBiTileWords)] + word_prefix[head >> 5] +
               (uint32_t)__popc(flag_bits[head >> 5] & ((1u << (head & 31)) - 1u));
        if (row == 0) {
            const uint32_t h = w & ~kBiNewBit;
            const Slot s = table[h];
            table[h].slab = slab;
            slab_key[slab] = s.key;
            slab_count[slab] = s.count;
        }
    } else {
        const Slot s = table[w & ~kBiFirstBit];
        slab = s.slab;
        if (w & kBiFirstBit) slab_count[slab] = s.count;  // the voxel's first survivor settles the slab's count
    }
    const size_t dst = ((size_t)slab * K + row) * 3;
    store3(pts + dst, pv);
    store3(nrm_out + dst, nv);
}

// ---------------------------------------------------------------------------
// cleanup / export kernels
// ---------------------------------------------------------------------------
// voxel_grid.h:238-241: erase iff (getOrigin() - point).squaredNorm() > radius_sq (f32, strict)
// (a slab with no points is a voxel an earlier cleanup erased -- k_cleanup_mark --: not kept, not counted)
__global__ void k_cleanup_flag(const float *pts, const uint32_t *slab_count, uint32_t K, uint32_t n_vox, float cx, float cy,
                               float cz, float r2, uint32_t *keep)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_vox) return;
    const float *o = pts + (size_t)s * K * 3;  // voxel_with_planes.h:32-35 front()
    const float dx = o[0] - cx, dy = o[1] - cy, dz = o[2] - cz;
    const float d2 = dx * dx + (dy * dy + dz * dz);
    keep[s] = (slab_count[s] == 0u || d2 > r2) ? 0u : 1u;
}

// voxel_grid.h:240 erase(it), without moving anybody: the erased voxel's slab keeps its place in the creation order with no
// points in it, and its key stays in the table as a claimed slot without a voxel (slab == kNoSlab, count 0 -- what a
// range error leaves behind, too): a search finds no candidates there, an insert finds "it == end()" (voxel_grid.h:83)
// and creates the voxel anew at the end of the creation order, exactly as after an erase.  Exports skip empty slabs.
// The holes are closed (k_compact, table rebuilt) once they are a quarter of the slabs.
__global__ void k_cleanup_mark(Slot *table, uint32_t mask, uint32_t shift, const uint32_t *__restrict__ keep, uint32_t n_vox,
                               const unsigned long long *__restrict__ slab_key, uint32_t *slab_count)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_vox || keep[s] || slab_count[s] == 0u) return;
    const unsigned long long key = slab_key[s];
    uint32_t h = hash_key(key, shift) & mask;
    for (uint32_t probe = 0; probe <= mask; probe++) {  // (the key is there: its voxel was live)
        const unsigned long long seen = table[h].key;
        if (seen == key) {
            table[h].count = 0u;
            table[h].slab = kNoSlab;
            break;
        }
        if (seen == kEmptyKey) break;
        h = (h + 1) & mask;
    }
    slab_count[s] = 0u;
}

// the same flags and their exclusive scan in one kernel (kItems consecutive voxels per thread, <= 256
// workgroups): keep[], newid[] and the number of voxels kept (words[4])
// `from`: the scan was enqueued behind an align on the same stream (lom_map_radius_cleanup_after_align) and takes its
// centre from the pose that align ended with -- lidar_odometry.cpp:65-67: current_transform_ = result, then
// radiusCleanup(current_transform_.translation, ...).  An align that has not ended there (more outer iterations to
// come, a give-up) leaves words[12] = 0 and the scan undone; otherwise words[12] = seq and words[13..15] = the bits of
// the centre used: the host takes the result only for exactly the centre it would have passed.  keep[] / newid[] are
// scratch either way.
template <int kItems>
__global__ __launch_bounds__(kThreads) void k_cleanup_scan(const float *pts, const uint32_t *slab_count, uint32_t K,
                                                           uint32_t n_vox, float cx, float cy,
                                                           float cz, float r2, uint32_t *keep, uint32_t *newid,
                                                           Granule *agg, uint32_t seq, uint32_t *words, uint32_t test_fail_from,
                                                           const AlignState *from = nullptr)
{
    __shared__ unsigned long long s_w[8];
    if (from) {  // (uniform over the grid: the align's kernels are through)
        typedef const __attribute__((address_space(4))) AlignState *ConstState;
        ConstState cs = (ConstState)(from);
        const int usable = cs->finished && !cs->error;
        cx = cs->pose_t[0];
        cy = cs->pose_t[1];
        cz = cs->pose_t[2];
        if (!usable) {
            if (blockIdx.x == 0 && threadIdx.x == 0) words[12] = 0u;
            return;
        }
    }
    const uint32_t base = (blockIdx.x * kThreads + threadIdx.x) * kItems;
    uint32_t f[kItems], mine = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t s = base + k;
        f[k] = 0;
        if (s < n_vox) {
            const float *o = pts + (size_t)s * K * 3;  // voxel_with_planes.h:32-35 front()
            const float dx = o[0] - cx, dy = o[1] - cy, dz = o[2] - cz;
            const float d2 = dx * dx + (dy * dy + dz * dz);
            f[k] = (slab_count[s] == 0u || d2 > r2) ? 0u : 1u;  // voxel_grid.h:238-241 (an empty slab: erased before)
        }
        mine += f[k];
    }
    unsigned long long total;
    const unsigned long long excl = block_scan64(mine, s_w, total);
    bool gave_up;
    const unsigned long long before = grid_prefix64(total, agg, seq, words + 7, s_w, gave_up, test_fail_from);
    uint32_t run = (uint32_t)(before + excl);
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        const uint32_t s = base + k;
        if (s < n_vox && !gave_up) {  // keep[] / newid[] are scratch: the host redoes a scan that gave up
            keep[s] = f[k];
            newid[s] = run;
        }
        run += f[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        words[4] = (uint32_t)(before + total);
        if (from) {
            words[13] = __float_as_uint(cx);
            words[14] = __float_as_uint(cy);
            words[15] = __float_as_uint(cz);
            words[12] = seq;
        }
    }
}

__global__ void k_compact(const uint32_t *keep, const uint32_t *newid, uint32_t n_vox, uint32_t K,
                          const unsigned long long *key_in, const uint32_t *cnt_in, const float *pts_in,
                          const float *nrm_in, unsigned long long *key_out, uint32_t *cnt_out, float *pts_out,
                          float *nrm_out, uint32_t *n_vox_dev, uint32_t n_keep)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) *n_vox_dev = n_keep;  // the device-side voxel counter follows the compaction
    if (idx >= (size_t)n_vox * K) return;
    const uint32_t s = (uint32_t)(idx / K), j = (uint32_t)(idx % K);
    if (!keep[s]) return;
    const uint32_t d = newid[s];
    const uint32_t c = cnt_in[s];
    if (j == 0) {
        key_out[d] = key_in[s];
        cnt_out[d] = c;
    }
    if (j < c) {
        const size_t a = ((size_t)s * K + j) * 3, b = ((size_t)d * K + j) * 3;
        pts_out[b] = pts_in[a];
        pts_out[b + 1] = pts_in[a + 1];
        pts_out[b + 2] = pts_in[a + 2];
        nrm_out[b] = nrm_in[a];
        nrm_out[b + 1] = nrm_in[a + 1];
        nrm_out[b + 2] = nrm_in[a + 2];
    }
}

__global__ void k_export_counts(const uint32_t *slab_count, uint32_t n_vox, int mode, uint32_t *out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_vox) out[s] = (mode == LOM_EXPORT_FIRST_PER_VOXEL) ? (slab_count[s] ? 1u : 0u) : slab_count[s];  // (empty slab: erased voxel)
}

__global__ void k_export_write(const uint32_t *off, const uint32_t *slab_count, uint32_t n_vox, uint32_t K,
                               int mode, const float *pts, const float *nrm, float *out_xyz, float *out_nrm)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_vox * K) return;
    const uint32_t s = (uint32_t)(idx / K), j = (uint32_t)(idx % K);
    const uint32_t c = (mode == LOM_EXPORT_FIRST_PER_VOXEL) ? (slab_count[s] ? 1u : 0u) : slab_count[s];
    if (j >= c) return;
    const size_t a = ((size_t)s * K + j) * 3, b = ((size_t)off[s] + j) * 3;
    out_xyz[b] = pts[a];
    out_xyz[b + 1] = pts[a + 1];
    out_xyz[b + 2] = pts[a + 2];
    if (out_nrm) {
        out_nrm[b] = nrm[a];
        out_nrm[b + 1] = nrm[a + 1];
        out_nrm[b + 2] = nrm[a + 2];
    }
}

// ---------------------------------------------------------------------------
// host-side management
// ---------------------------------------------------------------------------
static uint32_t next_pow2(uint64_t v)
{
    uint64_t p = 1024;
    while (p < v) p <<= 1;
    return (uint32_t)std::min<uint64_t>(p, 1ull << 31);
}

static int table_alloc(lom_map *m, uint32_t cap, Slot **out)
{
    Slot *t = nullptr;
    hipError_t e = hipMalloc(&t, (size_t)cap * sizeof(Slot));
    if (e != hipSuccess) return set_error(m, LOM_ERR_OOM, "hipMalloc(table)", e);
    hipLaunchKernelGGL(k_table_init, dim3(blocks_for(cap)), dim3(kThreads), 0, m->stream, t, cap);
    LOM_HIP(m, hipGetLastError());
    *out = t;
    return LOM_OK;
}

// scratch words: [0..1] u64 scan total, [2] flag, [4] u32 scan total, [6] device-side voxel counter
static uint32_t *d_nvox(lom_map *m) { return (uint32_t *)m->scr[S_MISC].p + 6; }
static uint32_t *d_word(lom_map *m, int i) { return (uint32_t *)m->scr[S_MISC].p + i; }

int read_words(lom_map *m, int first, int n);

// exact voxel count on the host (waits for pending inserts of this handle)
int resolve_pending(lom_map *m);

static int refresh_nvox(lom_map *m)
{
    int rc = resolve_pending(m);
    if (rc != LOM_OK) return rc;
    if (!m->n_vox_stale) return LOM_OK;
    rc = read_words(m, 6, 1);
    if (rc != LOM_OK) return rc;
    m->n_vox = m->h_flags[0];
    m->n_vox_ub = m->n_vox;
    m->n_vox_stale = false;
    return LOM_OK;
}

static int rehash(lom_map *m, uint32_t new_cap)
{
    m->mutations++;
    Slot *t = nullptr;
    int rc = table_alloc(m, new_cap, &t);
    if (rc != LOM_OK) return rc;
    Slot *old = m->d_table;
    m->d_table = t;
    m->cap = new_cap;
    if (m->n_vox) {
        const MapView v = view_of(m);
        hipLaunchKernelGGL(k_rebuild, dim3(blocks_for(m->n_vox)), dim3(kThreads), 0, m->stream, m->d_table,
                           v.mask, v.shift, m->d_slab_key, m->d_slab_count, m->n_vox);
        LOM_HIP(m, hipGetLastError());
    }
    if (old) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        LOM_HIP(m, hipFree(old));
    }
    return LOM_OK;
}

struct Slabs {
    unsigned long long *key = nullptr;
    uint32_t *count = nullptr;
    float *pts = nullptr, *nrm = nullptr;
};

static void slabs_free(Slabs &s)
{
    if (s.key) (void)hipFree(s.key);
    if (s.count) (void)hipFree(s.count);
    if (s.pts) (void)hipFree(s.pts);
    if (s.nrm) (void)hipFree(s.nrm);
    s = Slabs();
}

// k_match reads a chunk of four consecutive 12-byte rows from any live row on: the rows behind the last slab exist
constexpr size_t kRowPadBytes = 64;
static int slabs_alloc(lom_map *m, uint32_t cap, Slabs &s)
{
    const size_t pb = (size_t)cap * m->K * 3 * sizeof(float);
    if (hipMalloc(&s.key, (size_t)cap * 8) != hipSuccess || hipMalloc(&s.count, (size_t)cap * 4) != hipSuccess ||
        hipMalloc(&s.pts, pb + kRowPadBytes) != hipSuccess || hipMalloc(&s.nrm, pb + kRowPadBytes) != hipSuccess) {
        (void)hipGetLastError();
        slabs_free(s);
        return set_error(m, LOM_ERR_OOM, "hipMalloc(slabs)");
    }
    return LOM_OK;
}

static int ensure_slabs(lom_map *m, uint64_t want)
{
    if (want <= m->slab_cap) return LOM_OK;
    if (want > 0x7FFFFFFFull / std::max<uint32_t>(1, m->K)) return set_error(m, LOM_ERR_OOM, "map too large");
    uint32_t nc = std::max<uint32_t>(4096, m->slab_cap);
    while (nc < want) nc *= 2;
    Slabs s;
    int rc = slabs_alloc(m, nc, s);
    if (rc != LOM_OK) return rc;
    const uint32_t live = std::min(std::max(m->n_vox, m->n_vox_ub), m->slab_cap);  // upper bound of the slabs in use
    if (live) {
        const size_t pb = (size_t)live * m->K * 3 * sizeof(float);
        LOM_HIP(m, hipMemcpyAsync(s.key, m->d_slab_key, (size_t)live * 8, hipMemcpyDeviceToDevice, m->stream));
        LOM_HIP(m, hipMemcpyAsync(s.count, m->d_slab_count, (size_t)live * 4, hipMemcpyDeviceToDevice, m->stream));
        LOM_HIP(m, hipMemcpyAsync(s.pts, m->d_pts, pb, hipMemcpyDeviceToDevice, m->stream));
        LOM_HIP(m, hipMemcpyAsync(s.nrm, m->d_nrm, pb, hipMemcpyDeviceToDevice, m->stream));
    }
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    Slabs old{m->d_slab_key, m->d_slab_count, m->d_pts, m->d_nrm};
    slabs_free(old);
    m->d_slab_key = s.key;
    m->d_slab_count = s.count;
    m->d_pts = s.pts;
    m->d_nrm = s.nrm;
    m->slab_cap = nc;
    return LOM_OK;
}

// Device words -> host in ONE launch and no copy engine: a single wave stores {word, call tag} pairs as
// 64-bit system-scope words into the handle's coherent pinned block; the host watches the tags.  (A
// hipMemcpyAsync per word is a 4 us blit kernel each plus its enqueue: ten of them per frame of the streaming
// path were 15 % of its kernel time.)  The stream is in order, so the words arriving also says that
// everything enqueued before them is through.
struct WordPtrs {
    const uint32_t *p[32];
};

__global__ __launch_bounds__(64) void k_gather_words(WordPtrs w, int n, unsigned long long *host_out, uint32_t tag)
{
    const int i = (int)threadIdx.x;
    if (i >= n) return;
    const uint32_t v = __hip_atomic_load(w.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host_out + i, (unsigned long long)v | ((unsigned long long)tag << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr size_t kWordsOffset = 512;  // of h_report / d_report: [0, 256) AlignReport, [512, 768) these words

int gather_words_begin(lom_map *m, const uint32_t *const *ptrs, int n)
{
    m->words_pending = 0;
    if (n <= 0) return LOM_OK;
    WordPtrs w;
    for (int i = 0; i < 32; i++) w.p[i] = i < n ? ptrs[i] : nullptr;
    if (++m->words_tag == 0) m->words_tag = 1;  // the block starts zeroed: 0 is "nothing yet"
    hipLaunchKernelGGL(k_gather_words, dim3(1), dim3(64), 0, m->stream, w, n,
                       reinterpret_cast<unsigned long long *>((char *)m->d_report + kWordsOffset), m->words_tag);
    LOM_HIP(m, hipGetLastError());
    m->words_pending = n;
    return LOM_OK;
}

int gather_words_end(lom_map *m, uint32_t *out)
{
    const int n = m->words_pending;
    const uint32_t tag = m->words_tag;
    m->words_pending = 0;
    volatile unsigned long long *hw = reinterpret_cast<volatile unsigned long long *>((char *)m->h_report + kWordsOffset);
    uint64_t spins = 0;
    for (int i = 0; i < n; i++) {
        while ((uint32_t)(hw[i] >> 32) != tag) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFF) != 0) continue;
            const hipError_t e = hipStreamQuery(m->stream);
            if (e == hipSuccess) {
                if ((uint32_t)(hw[i] >> 32) == tag) break;
                return set_error(m, LOM_ERR_HIP, "device words did not arrive");
            }
            if (e != hipErrorNotReady) return set_error(m, LOM_ERR_HIP, "stream failed while reading device words", e);
        }
        out[i] = (uint32_t)hw[i];
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return LOM_OK;
}

int gather_words(lom_map *m, const uint32_t *const *ptrs, int n, uint32_t *out)
{
    const int rc = gather_words_begin(m, ptrs, n);
    return rc != LOM_OK ? rc : gather_words_end(m, out);
}

int read_words(lom_map *m, int first, int n)
{
    const uint32_t *ptrs[32];
    for (int i = 0; i < n; i++) ptrs[i] = d_word(m, first + i);
    return gather_words(m, ptrs, n, m->h_flags);
}

// grow-only scratch that is kept at rest (every byte == fill) between calls: a fresh allocation is
// filled once; whoever dirties a word puts it back
static int ensure_rest(lom_map *m, DeviceBuf &b, size_t bytes, int fill)
{
    if (bytes <= b.bytes) return LOM_OK;
    const int rc = ensure(m, b, bytes);
    if (rc != LOM_OK) return rc;
    LOM_HIP(m, hipMemsetAsync(b.p, fill, b.bytes, m->stream));
    return LOM_OK;
}

static Granule *d_agg(lom_map *m) { return (Granule *)((char *)m->scr[S_MISC].p + 256); }

static int add_points_device(lom_map *m, const char *d_xyz, const char *d_nrm, size_t n, size_t stride,
                             bool validated_on_host, bool sync_status, bool allow_shrink = true, bool multi_launch = false);

// one-shot test hook (LOM_OPT_TEST_GRID_GIVE_UP): the first workgroup that gives up in the next in-kernel scan
static uint32_t take_test_fail_from(lom_map *m)
{
    const int v = m->test_grid_give_up;
    if (v >= 65536) {  // + 65536 per in-kernel scan to let pass first
        m->test_grid_give_up = v - 65536;
        return 0xFFFFFFFFu;
    }
    m->test_grid_give_up = -1;
    return v < 0 ? 0xFFFFFFFFu : (uint32_t)v;
}

// deferred verdict of the calls enqueued since the last check (waits for them): a point out of range
// (such a call inserted / returned nothing) or a workgroup that gave up waiting inside a single-pass kernel.
// A call that gave up has changed nothing (grid_scan.hpp); if it is the handle's last insert -- whose input
// buffers the caller keeps valid until this check -- it is redone here with the multi-launch scan, whose
// kernels do not wait for each other.
static int map_status(lom_map *m)
{
    // [0] range flag, [1] voxel counter, [2] grid error, [3] bulk insert sent back: sequence numbers of failed calls
    int rc = read_words(m, 5, 4);
    if (rc != LOM_OK) return rc;
    const uint32_t checked = m->status_seq;
    m->status_seq = m->call_seq;
    if (m->n_vox_stale) {
        m->n_vox = m->h_flags[1];
        m->n_vox_ub = m->n_vox;
        m->n_vox_stale = false;
    }
    const uint32_t range_seq = m->h_flags[0];
    const size_t pending = m->pending_n;
    m->pending_n = 0;  // the last single-pass / bulk insert is through, one way or the other
    // a bulk insert with a partition beyond k_bi_group's LDS wrote nothing: the four-kernel path redoes it, like a
    // single-pass insert whose scan gave up
    const uint32_t grid_seq = (pending && m->h_flags[3] == m->pending_seq) ? m->h_flags[3] : m->h_flags[2];
    if (grid_seq > checked && grid_seq != m->grid_resolved_seq) {
        if (grid_seq == m->pending_seq && pending) {
            m->grid_redos++;
            m->grid_resolved_seq = grid_seq;
            const int rc2 = add_points_device(m, m->pending_xyz, m->pending_nrm, pending, m->pending_stride, false, true, true, true);
            if (rc2 != LOM_OK) return rc2;
            if (range_seq > checked && range_seq != grid_seq)
                return set_error(m, LOM_ERR_RANGE, "coordinate / voxel_size out of range or not finite");
            return LOM_OK;
        }
        return set_error(m, LOM_ERR_HIP,
                         "a workgroup timed out waiting for the others of its grid; that call changed nothing, repeat it");
    }
    if (range_seq > checked) return set_error(m, LOM_ERR_RANGE, "coordinate / voxel_size out of range or not finite");
    return LOM_OK;
}

// The handle's last single-pass insert has not been looked at yet (a caller that never asks for lom_map_status):
// before anything consumes or changes the map, see whether its in-kernel scan gave up, and redo it if so.
// Callers that check lom_map_status() themselves (the streaming path) never pay this read-back.
static int settle_pending_locked(lom_map *m);
int resolve_pending(lom_map *m)
{
    if (m->parent) {
        // A scan context.  The reference's calls are synchronous: an align that follows an addCloud sees its points.
        // Here the map's maintenance kernels run on the MAP's stream and a context has a stream of its own, so the
        // first call of a context after the map changed (a) settles an insert nobody has looked at yet and (b) orders
        // the context's stream behind the map's.  (Nobody changes the map while contexts are in use; several contexts
        // may arrive here together after a change, hence the lock.)
        lom_map *p = m->parent;
        if (!p->pending_n.load(std::memory_order_acquire) && m->seen_mutations == p->mutations.load(std::memory_order_acquire))
            return LOM_OK;
        std::lock_guard<std::mutex> lock(p->settle_mutex);
        if (p->pending_n) {
            const int rc = settle_pending_locked(p);
            if (rc != LOM_OK) return set_error(m, rc, p->last_error.c_str());
        }
        const uint64_t now = p->mutations;
        if (m->seen_mutations != now) {
            if (!m->parent_ev) LOM_HIP(m, hipEventCreateWithFlags(&m->parent_ev, hipEventDisableTiming));
            LOM_HIP(m, hipEventRecord(m->parent_ev, p->stream));
            LOM_HIP(m, hipStreamWaitEvent(m->stream, m->parent_ev, 0));
            m->seen_mutations = now;
        }
        return LOM_OK;
    }
    if (!m->pending_n.load(std::memory_order_acquire)) return LOM_OK;
    // the map's own caller: contexts of this map may be settling the same insert right now
    std::lock_guard<std::mutex> lock(m->settle_mutex);
    return settle_pending_locked(m);
}

// the caller holds m->settle_mutex (m is a map, not a context)
static int settle_pending_locked(lom_map *m)
{
    if (!m->pending_n) return LOM_OK;  // somebody else settled it while this thread waited for the lock
    int rc = read_words(m, 7, 2);  // scan gave up / bulk insert sent back
    if (rc != LOM_OK) return rc;
    const size_t n = m->pending_n;
    m->pending_n = 0;
    if (m->h_flags[0] != m->pending_seq && m->h_flags[1] != m->pending_seq) return LOM_OK;
    m->grid_redos++;
    m->grid_resolved_seq = m->pending_seq;
    return add_points_device(m, m->pending_xyz, m->pending_nrm, n, m->pending_stride, false, true, true, true);
}

// Table size after a bulk insert: 16..32 slots per voxel (LOM_TABLE_SLOTS_PER_VOXEL at create).  The search's probe
// phase pays for every collision with a dependent round trip, and a longer table costs nothing but memory: k_match on
// C2 / C3 / C4 with 4 slots per voxel (rounds 1-2) 7.5 / 25.1 / 43.2 us, 8: 7.5 / 24.0 / 41.1, 16: 7.2 / 23.9 / 40.7,
// 32: 7.2 / 23.8 / 40.6, 64: 7.2 / 23.4 / 40.8 (same box, tools/ab_match.py).
static int shrink_after_bulk(lom_map *m)
{
    if ((uint64_t)m->cap < 16ull * std::max<uint32_t>(m->min_cap, 1u)) return LOM_OK;
    int rc = refresh_nvox(m);
    if (rc != LOM_OK) return rc;
    const uint32_t target = std::max(m->min_cap, next_pow2((uint64_t)m->table_slots_per_voxel * m->n_vox));
    return m->cap > target ? rehash(m, target) : LOM_OK;
}

struct BulkShape {
    uint32_t n_parts, part_shift, part_max, ppt, n_blk, n_words;
};

static BulkShape bulk_shape(uint32_t N, uint32_t cap, uint32_t ppt_override = 0)
{
    BulkShape b;
    // ~128 points per partition (the 512-point shape of k_bi_group: 20 KB of LDS, eight workgroups per CU) while the
    // partitions number at most kBiMaxParts; beyond 2 M points the partitions grow and the 1024-point shape takes over
    uint32_t np = 64;
    while (np < kBiMaxParts && (uint64_t)np * 128 < N) np <<= 1;
    b.n_parts = std::min(np, cap);
    b.part_shift = (uint32_t)__builtin_ctz(cap) - (uint32_t)__builtin_ctz(b.n_parts);
    b.part_max = (uint64_t)b.n_parts * 128 >= N ? 512u : kBiPartMax;
    b.ppt = N <= (1u << 20) ? 4u : 8u;
    if (ppt_override == 2 || ppt_override == 4 || ppt_override == 8) b.ppt = ppt_override;
    b.n_blk = (N + b.ppt * kBiThreads - 1) / (b.ppt * kBiThreads);
    b.n_words = (N + 31) / 32;
    return b;
}

static int bulk_scratch(lom_map *m, uint32_t N, const BulkShape &b)
{
    int rc;
    for (int s : {S_PT_OFF, S_PT_M, S_ENT_ROW})
        if ((rc = ensure(m, m->scr[s], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_PT_SLOT], (size_t)N * 8)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_PT_POS], (size_t)N * 16)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_FLAG], (size_t)b.n_words * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_RANK], (size_t)b.n_words * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_HIST], (size_t)b.n_parts * b.n_blk * 4)) != LOM_OK) return rc;
    // partition sizes and starts, then the bitmap's tiles: totals, prefixes, and the counter of finished tiles (at rest: 0)
    const size_t tiles = (b.n_words + kBiTileWords - 1) / kBiTileWords;
    const bool fresh = m->scr[S_PART].bytes < (size_t)(2 * kBiMaxParts + 1 + 2 * 256 + 1) * 4;
    if ((rc = ensure(m, m->scr[S_PART], (size_t)(2 * kBiMaxParts + 1 + 2 * 256 + 1) * 4)) != LOM_OK) return rc;
    if (fresh) LOM_HIP(m, hipMemsetAsync(m->scr[S_PART].p, 0, m->scr[S_PART].bytes, m->stream));
    return tiles <= 256 ? LOM_OK : set_error(m, LOM_ERR_ARG, "bulk insert: too many points");
}

// batches above kOnePassMax points (see the kernels): everything is enqueued, nothing waits; the verdict -- range error,
// or a partition beyond k_bi_group's LDS, which sends the call to the four-kernel path -- is read with the call's status
static int add_points_bulk(lom_map *m, const char *d_xyz, const char *d_nrm, uint32_t N, size_t stride, uint64_t worst,
                           bool validated_on_host, bool sync_status, bool allow_shrink)
{
    int rc;
    const BulkShape b = bulk_shape(N, m->cap, m->bulk_ppt);
    if ((rc = bulk_scratch(m, N, b)) != LOM_OK) return rc;
    if (worst > m->slab_cap && (rc = ensure_slabs(m, worst + worst / 2)) != LOM_OK) return rc;
    uint2 *pt_info = (uint2 *)m->scr[S_PT_SLOT].p;
    uint32_t *flag_bits = (uint32_t *)m->scr[S_FLAG].p, *word_prefix = (uint32_t *)m->scr[S_RANK].p;
    uint4 *part_rec = (uint4 *)m->scr[S_PT_POS].p;
    uint32_t *ent_idx = (uint32_t *)m->scr[S_PT_OFF].p, *ent_w = (uint32_t *)m->scr[S_PT_M].p;
    uint32_t *ent_row = (uint32_t *)m->scr[S_ENT_ROW].p;
    uint32_t *hist = (uint32_t *)m->scr[S_HIST].p;
    uint32_t *part_total = (uint32_t *)m->scr[S_PART].p, *part_start = part_total + kBiMaxParts;
    uint32_t *tile_total = part_start + kBiMaxParts + 1, *tile_prefix = tile_total + 256, *tiles_done = tile_prefix + 256;
    const uint32_t n_tiles = (b.n_words + kBiTileWords - 1) / kBiTileWords;
    uint32_t *words = d_word(m, 0);
    const uint32_t seq = ++m->call_seq;
    m->mutations++;
    const MapView v = view_of(m);
    const uint32_t part_max = m->test_bulk_part_max ? std::min<uint32_t>(m->test_bulk_part_max, b.part_max) : b.part_max;
    const size_t lds = (size_t)b.n_parts * 4;
    if (lds + 256 > 65536) {  // the histograms of 16,384 partitions fill the 64 KB a launch gets without asking
        static std::once_flag once;
        std::call_once(once, [] {
            const int want = (int)kBiMaxParts * 4;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_claim<2>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_scatter<2>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_claim<4>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_claim<8>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_scatter<4>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bi_scatter<8>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
        });
    }
    const dim3 gb(b.n_blk), tb(kBiThreads);
    if (b.ppt == 2)
        hipLaunchKernelGGL(k_bi_claim<2>, gb, tb, lds, m->stream, m->d_table, v.mask, v.shift, d_xyz, stride, N, m->voxel_size,
                           pt_info, flag_bits, hist, b.n_parts, b.part_shift, d_nvox(m), seq, words);
    else if (b.ppt == 4)
        hipLaunchKernelGGL(k_bi_claim<4>, gb, tb, lds, m->stream, m->d_table, v.mask, v.shift, d_xyz, stride, N, m->voxel_size,
                           pt_info, flag_bits, hist, b.n_parts, b.part_shift, d_nvox(m), seq, words);
    else
        hipLaunchKernelGGL(k_bi_claim<8>, gb, tb, lds, m->stream, m->d_table, v.mask, v.shift, d_xyz, stride, N, m->voxel_size,
                           pt_info, flag_bits, hist, b.n_parts, b.part_shift, d_nvox(m), seq, words);
    hipLaunchKernelGGL(k_bi_colscan, dim3((b.n_parts + 63) / 64), dim3(kBiThreads), 0, m->stream, hist, b.n_parts, b.n_blk,
                       part_total, part_max, seq, words);
    if (b.ppt == 2)
        hipLaunchKernelGGL(k_bi_scatter<2>, gb, tb, lds, m->stream, N, pt_info, hist, b.n_parts, b.part_shift, part_total,
                           part_start, part_rec, seq, words);
    else if (b.ppt == 4)
        hipLaunchKernelGGL(k_bi_scatter<4>, gb, tb, lds, m->stream, N, pt_info, hist, b.n_parts, b.part_shift, part_total,
                           part_start, part_rec, seq, words);
    else
        hipLaunchKernelGGL(k_bi_scatter<8>, gb, tb, lds, m->stream, N, pt_info, hist, b.n_parts, b.part_shift, part_total,
                           part_start, part_rec, seq, words);
    if (b.part_max == 512)
        hipLaunchKernelGGL(k_bi_group<512>, dim3(b.n_parts), dim3(kThreads), 0, m->stream, m->d_table, part_start, part_rec,
                           m->max_points, flag_bits, ent_idx, ent_w, ent_row, seq, words);
    else
        hipLaunchKernelGGL(k_bi_group<kBiPartMax>, dim3(b.n_parts), dim3(kThreads), 0, m->stream, m->d_table, part_start,
                           part_rec, m->max_points, flag_bits, ent_idx, ent_w, ent_row, seq, words);
    hipLaunchKernelGGL(k_bi_flagscan, dim3(n_tiles), dim3(kThreads), 0, m->stream, flag_bits, b.n_words, word_prefix,
                       tile_total, tile_prefix, tiles_done, words + 10, seq, words);
    hipLaunchKernelGGL(k_bi_place, dim3(blocks_for(N)), dim3(kThreads), 0, m->stream, m->d_table, N, ent_idx, ent_w, ent_row,
                       flag_bits, word_prefix, tile_prefix, words + 10, d_xyz, d_nrm, stride, m->K, m->d_pts, m->d_nrm,
                       m->d_slab_key, m->d_slab_count, d_nvox(m), seq, words);
    LOM_HIP(m, hipGetLastError());
    // should a partition have been too large, lom_map_status() / whoever consumes the map next redoes this insert with
    // the four-kernel path: the caller keeps the input valid until then (as for a single-pass insert)
    m->pending_xyz = d_xyz;
    m->pending_nrm = d_nrm;
    m->pending_stride = stride;
    m->pending_seq = seq;
    m->pending_n = N;
    m->table_clean = false;
    m->n_vox_ub = (uint32_t)std::min<uint64_t>(worst, 0xFFFFFFFFull);
    m->n_vox_stale = true;
    if (sync_status && (rc = map_status(m)) != LOM_OK) return rc;
    return allow_shrink ? shrink_after_bulk(m) : LOM_OK;
}

// sync_status: wait for the insert's verdict (LOM_ERR_RANGE when a point's index is out of range; such a
// call inserts nothing).  Without it the call only enqueues; lom_map_status() reports later.
static int add_points_device(lom_map *m, const char *d_xyz, const char *d_nrm, size_t n, size_t stride,
                             bool validated_on_host, bool sync_status, bool allow_shrink, bool multi_launch)
{
    if (m->parent) return set_error(m, LOM_ERR_STATE, "a scan context has no map of its own");
    if (n == 0) return LOM_OK;
    if (n >= 0x7FFFFFFFull) return set_error(m, LOM_ERR_ARG, "too many points in one call");
    const uint32_t N = (uint32_t)n;
    int rc;
    if ((rc = resolve_pending(m)) != LOM_OK) return rc;  // inserts apply in call order
    // 1. table capacity for the worst case (every point a new voxel); shrunk afterwards
    uint64_t worst = (uint64_t)m->n_vox_ub + N;
    if ((uint64_t)m->cap < 2 * worst) {
        if ((rc = refresh_nvox(m)) != LOM_OK) return rc;
        worst = (uint64_t)m->n_vox + N;
        if ((uint64_t)m->cap < 2 * worst && (rc = rehash(m, next_pow2(4 * worst))) != LOM_OK) return rc;
    }
    // 2. scratch
    const bool one_pass = N <= kOnePassMax && !multi_launch;
    if (N > kOnePassMax && N <= kBiMaxPoints && !multi_launch && !m->opt_no_bulk)
        return add_points_bulk(m, d_xyz, d_nrm, N, stride, worst, validated_on_host, sync_status, allow_shrink);
    if ((rc = ensure(m, m->scr[S_PT_SLOT], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_PT_POS], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_PT_OFF], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_PT_M], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_ITEMS], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure_rest(m, m->scr[S_BKT_CNT], (size_t)m->cap * 4, 0)) != LOM_OK) return rc;
    if ((rc = ensure_rest(m, m->scr[S_BKT_HEAD], (size_t)m->cap * 4, 0xFF)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_BKT_OFF], (size_t)m->cap * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_BKT_OLD], (size_t)m->cap * 4)) != LOM_OK) return rc;
    if (!one_pass) {
        if ((rc = ensure(m, m->scr[S_FLAG], (size_t)N * 8)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_RANK], (size_t)N * 8)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_SCAN], scan_tmp_words(N) * 8)) != LOM_OK) return rc;
    }
    uint32_t *pt_slot = (uint32_t *)m->scr[S_PT_SLOT].p, *pt_pos = (uint32_t *)m->scr[S_PT_POS].p;
    uint32_t *pt_off = (uint32_t *)m->scr[S_PT_OFF].p, *pt_m = (uint32_t *)m->scr[S_PT_M].p;
    uint32_t *items = (uint32_t *)m->scr[S_ITEMS].p;
    uint32_t *bcnt = (uint32_t *)m->scr[S_BKT_CNT].p, *bhead = (uint32_t *)m->scr[S_BKT_HEAD].p;
    uint32_t *boff = (uint32_t *)m->scr[S_BKT_OFF].p, *bold = (uint32_t *)m->scr[S_BKT_OLD].p;
    uint32_t *words = d_word(m, 0);
    const uint32_t seq = ++m->call_seq;
    m->mutations++;
    const MapView v = view_of(m);
    const dim3 g(blocks_for(N)), b(kThreads);
    hipLaunchKernelGGL(k_ins_claim2, g, b, 0, m->stream, m->d_table, v.mask, v.shift, d_xyz, stride, N, m->voxel_size,
                       pt_slot, pt_pos, bcnt, bhead, seq, words);
    LOM_HIP(m, hipGetLastError());
    // Nothing below depends on a host decision: the slabs are grown for the worst case up front (memory is
    // not the constraint on a 288 GB device; the voxel counter lives on the device), so the call only
    // enqueues.  The exact voxel count is read back by whoever needs it (refresh_nvox).
    if (worst > m->slab_cap && (rc = ensure_slabs(m, worst + worst / 2)) != LOM_OK) return rc;
    if (!one_pass) {
        // large batches: flags, one 64-bit scan (1-3 launches), assignment
        if ((rc = ensure(m, m->scr[S_FLAG], (size_t)N * 8)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_RANK], (size_t)N * 8)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_SCAN], scan_tmp_words(N) * 8)) != LOM_OK) return rc;
        unsigned long long *flag64 = (unsigned long long *)m->scr[S_FLAG].p;
        unsigned long long *scan64 = (unsigned long long *)m->scr[S_RANK].p;
        hipLaunchKernelGGL(k_ins_heads, g, b, 0, m->stream, m->d_table, N, pt_slot, bcnt, bhead, flag64, seq, words);
        LOM_HIP(m, hipGetLastError());
        unsigned long long *d_total64 = (unsigned long long *)m->scr[S_MISC].p;
        if ((rc = scan_exclusive<unsigned long long>(m, flag64, scan64, N, d_total64,
                                                     (unsigned long long *)m->scr[S_SCAN].p)) != LOM_OK)
            return rc;
        hipLaunchKernelGGL(k_ins_assign, g, b, 0, m->stream, m->d_table, N, pt_slot, bhead, flag64, scan64, d_nvox(m),
                           m->d_slab_key, boff, bold, seq, words);
    } else {
        hipLaunchKernelGGL(k_ins_assign2, g, b, 0, m->stream, m->d_table, N, pt_slot, bcnt, bhead, boff, bold, d_nvox(m),
                           m->d_slab_key, d_agg(m), seq, words, take_test_fail_from(m));
        // should its scan give up, lom_map_status() / the wait below redoes this insert: the caller keeps the
        // input valid until then
        m->pending_xyz = d_xyz;
        m->pending_nrm = d_nrm;
        m->pending_n = n;
        m->pending_stride = stride;
        m->pending_seq = seq;
    }
    hipLaunchKernelGGL(k_ins_scatter2, g, b, 0, m->stream, N, pt_slot, pt_pos, boff, bcnt, items, pt_off, pt_m, seq, words);
    hipLaunchKernelGGL(k_ins_place2, g, b, 0, m->stream, m->d_table, N, pt_slot, pt_off, pt_m, bcnt, bhead, bold, items,
                       d_xyz, d_nrm, stride, m->K, m->max_points, m->d_pts, m->d_nrm, m->d_slab_count, d_nvox(m), seq, words);
    LOM_HIP(m, hipGetLastError());
    m->table_clean = false;
    m->n_vox_ub = (uint32_t)std::min<uint64_t>(worst, 0xFFFFFFFFull);
    m->n_vox_stale = true;
    if (sync_status && (!validated_on_host || multi_launch)) {
        if ((rc = map_status(m)) != LOM_OK) return rc;
    }
    if (allow_shrink && N > kOnePassMax && (rc = shrink_after_bulk(m)) != LOM_OK) return rc;
    return LOM_OK;
}

// pinned bounce buffer: the caller's (pageable) memory is copied once on the CPU, the H2D copy is
// then truly asynchronous and the call can return before the GPU has consumed it
static int stage_pinned(lom_map *m, size_t bytes, char **out)
{
    if (m->stage_ev) LOM_HIP(m, hipEventSynchronize(m->stage_ev));  // the previous H2D has read the buffer
    if (bytes > m->h_stage_bytes) {
        if (m->h_stage) LOM_HIP(m, hipHostFree(m->h_stage));
        m->h_stage = nullptr;
        m->h_stage_bytes = 0;
        const size_t nb = std::max(bytes + bytes / 2, (size_t)1 << 20);
        hipError_t e = hipHostMalloc(&m->h_stage, nb, hipHostMallocDefault);
        if (e != hipSuccess) return set_error(m, LOM_ERR_OOM, "hipHostMalloc(stage)", e);
        m->h_stage_bytes = nb;
    }
    if (!m->stage_ev) LOM_HIP(m, hipEventCreateWithFlags(&m->stage_ev, hipEventDisableTiming));
    *out = (char *)m->h_stage;
    return LOM_OK;
}

static int stage_host_points(lom_map *m, const float *xyz, const float *nrm, size_t n, size_t stride,
                             const char **d_xyz, const char **d_nrm)
{
    // Copy the caller's (possibly interleaved) records as they are; the kernels
    // read them with the caller's stride, so PCL structs need no repacking.
    int rc;
    const size_t bytes = (n - 1) * stride + 12;
    const char *hx = (const char *)xyz, *hn = (const char *)nrm;
    char *pin = nullptr;
    *d_nrm = nullptr;
    if (nrm && hn >= hx && (size_t)(hn - hx) + 12 <= stride) {  // normals inside the same record
        const size_t all = (n - 1) * stride + (size_t)(hn - hx) + 12;
        if ((rc = ensure(m, m->scr[S_IN_XYZ], all)) != LOM_OK) return rc;
        if ((rc = stage_pinned(m, all, &pin)) != LOM_OK) return rc;
        std::memcpy(pin, hx, all);
        LOM_HIP(m, hipMemcpyAsync(m->scr[S_IN_XYZ].p, pin, all, hipMemcpyHostToDevice, m->stream));
        LOM_HIP(m, hipEventRecord(m->stage_ev, m->stream));
        *d_xyz = (const char *)m->scr[S_IN_XYZ].p;
        *d_nrm = *d_xyz + (hn - hx);
        return LOM_OK;
    }
    const size_t padded = (bytes + 255) & ~size_t(255);
    if ((rc = ensure(m, m->scr[S_IN_XYZ], bytes)) != LOM_OK) return rc;
    if (nrm && (rc = ensure(m, m->scr[S_IN_NRM], bytes)) != LOM_OK) return rc;
    if ((rc = stage_pinned(m, nrm ? 2 * padded : padded, &pin)) != LOM_OK) return rc;
    std::memcpy(pin, hx, bytes);
    LOM_HIP(m, hipMemcpyAsync(m->scr[S_IN_XYZ].p, pin, bytes, hipMemcpyHostToDevice, m->stream));
    *d_xyz = (const char *)m->scr[S_IN_XYZ].p;
    if (nrm) {
        std::memcpy(pin + padded, hn, bytes);
        LOM_HIP(m, hipMemcpyAsync(m->scr[S_IN_NRM].p, pin + padded, bytes, hipMemcpyHostToDevice, m->stream));
        *d_nrm = (const char *)m->scr[S_IN_NRM].p;
    }
    LOM_HIP(m, hipEventRecord(m->stage_ev, m->stream));
    return LOM_OK;
}

}  // namespace lom

using namespace lom;

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int lom_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int lom_device_local_cpus(int device, char *out, size_t cap)
{
    // CPUs of the NUMA node the GPU hangs off: /sys/bus/pci/devices/<bdf>/local_cpulist.  The align
    // is a chain of host<->device round trips over PCIe; a caller running on the far socket pays
    // for it (measured: 0.47 ms instead of 0.33 ms per C2 frame).  The caller decides what to do
    // with the list (bench.py pins itself to it).
    if (!out || cap < 2) return LOM_ERR_ARG;
    out[0] = 0;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return LOM_ERR_NO_DEVICE;
    }
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return LOM_ERR_STATE;
    const bool ok = std::fgets(out, (int)cap, f) != nullptr;
    std::fclose(f);
    if (!ok) return LOM_ERR_STATE;
    for (char *c = out; *c; c++)
        if (*c == '\n') *c = 0;
    return LOM_OK;
}

const char *lom_last_error(const lom_map *m) { return m ? m->last_error.c_str() : g_create_error.c_str(); }

// what every handle owns besides a map: a stream and the pinned blocks the align talks to the host through
// (part >= 0: the stream runs on partition `part` of `nparts` equal slices of the device's compute units)
static hipError_t create_stream(lom_map *m, int part, int nparts)
{
    if (part < 0) return hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
    int cus = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device);
    if (e != hipSuccess) return e;
    // Bit i of the mask is the device's i-th compute unit in the driver's enumeration, which deals consecutive bits
    // round-robin over the XCDs: a contiguous range of bits is the same number of CUs on every XCD.
    const uint32_t lo = (uint32_t)((uint64_t)cus * (uint32_t)part / (uint32_t)nparts);
    const uint32_t hi = (uint32_t)((uint64_t)cus * ((uint32_t)part + 1u) / (uint32_t)nparts);
    // What the slice holds AT ONCE of a grid whose workgroups wait for each other (k_lm): the dispatcher deals workgroups
    // round-robin over the 8 XCDs and, inside an XCD, over its 4 shader engines, whatever the mask says; a slice of c CUs
    // has at least floor(c / 32) of them in each of the 32 (XCD, engine) pairs, so that many workgroups per pair always find
    // a CU.  2, 4, 8 slices: 128, 64, 32 (all of the slice); 3 slices: 64 of 85; 5, 6, 7 slices: 32 of 51, 42, 36 -- with
    // the whole 42 counted, a solve of 42 workgroups found one pair short and waited out its patience on every align.
    constexpr uint32_t kDispatchPairs = 32;
    uint32_t usable = hi - lo;
    if ((uint32_t)cus % kDispatchPairs == 0u && usable >= kDispatchPairs) usable = usable / kDispatchPairs * kDispatchPairs;
    std::vector<uint32_t> mask(((size_t)cus + 31) / 32, 0u);
    for (uint32_t c = lo; c < hi; c++) mask[c >> 5] |= 1u << (c & 31);
    e = hipExtStreamCreateWithCUMask(&m->own_stream, (uint32_t)mask.size(), mask.data());
    if (e == hipSuccess) m->partition_cus = std::max(usable, 1u);
    return e;
}

static int handle_setup(lom_map *m, int part = -1, int nparts = 1)
{
    hipError_t e;
    if ((e = hipSetDevice(m->device)) != hipSuccess ||
        (e = create_stream(m, part, nparts)) != hipSuccess ||
        (e = hipHostMalloc((void **)&m->h_results, 1024 * sizeof(double), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void **)&m->h_flags, 64 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void **)&m->h_mail, 64 * 32 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) !=
            hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&m->d_mail, m->h_mail, 0)) != hipSuccess ||
        (e = hipHostMalloc(&m->h_cmd, 256, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer(&m->d_cmd, m->h_cmd, 0)) != hipSuccess ||
        (e = hipHostMalloc(&m->h_report, 1024, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer(&m->d_report, m->h_report, 0)) != hipSuccess)
        return set_error(nullptr, LOM_ERR_HIP, "handle setup", e);
    m->stream = m->own_stream;
    std::memset(m->h_mail, 0, 64 * 32 * sizeof(double));
    std::memset(m->h_cmd, 0, 256);
    std::memset(m->h_report, 0, 1024);
    return LOM_OK;
}

// ---- scan contexts ---------------------------------------------------------------------------------------------
// The reference's search and align take the grid by const reference (voxel_grid.h:164,206; cloud_matcher.h:15-16):
// any number of callers may align against one keyframe at a time.  A context is a handle without a map of its own
// -- stream, per-scan buffers, solve state, report block -- whose kernels read the keyframe's table and slabs.
int lom_scan_create(lom_map *map, lom_scan **out) { return lom_scan_create_on_partition(map, -1, 1, out); }

int lom_scan_create_on_partition(lom_map *map, int part, int nparts, lom_scan **out)
{
    if (!map || !out) return LOM_ERR_ARG;
    *out = nullptr;
    if (part >= 0 && (nparts < 1 || nparts > 8 || part >= nparts))
        return set_error(map, LOM_ERR_ARG, "partition index / count: 0 <= part < nparts <= 8");
    if (map->parent) return set_error(map, LOM_ERR_ARG, "a scan context cannot be the keyframe of another");
    LOM_HIP(map, hipSetDevice(map->device));
    int rc;
    {   // settles a pending insert (nothing mutates the keyframe while contexts read it).  Under the map's settle lock:
        // the C++ mirror's worker threads create their contexts -- and other contexts make their first call -- together.
        std::lock_guard<std::mutex> lock(map->settle_mutex);
        rc = settle_pending_locked(map);
        if (rc == LOM_OK) rc = refresh_nvox(map);
    }
    if (rc != LOM_OK) return rc;
    lom_map *c = new (std::nothrow) lom_map();
    if (!c) return set_error(map, LOM_ERR_OOM, "host allocation");
    c->device = map->device;
    c->parent = map;
    c->opt_host_lm = map->opt_host_lm;
    c->opt_debug_lm = map->opt_debug_lm;
    c->opt_debug_timing = map->opt_debug_timing;
    c->opt_no_temporal = map->opt_no_temporal;
    c->opt_count = map->opt_count;
    c->patience_ticks = map->patience_ticks;
    if (handle_setup(c, part, nparts) != LOM_OK) {
        map->last_error = g_create_error;
        lom_map_destroy(c);
        return LOM_ERR_HIP;
    }
    *out = reinterpret_cast<lom_scan *>(c);
    return LOM_OK;
}

void lom_scan_destroy(lom_scan *s) { lom_map_destroy(reinterpret_cast<lom_map *>(s)); }
const char *lom_scan_last_error(const lom_scan *s) { return s ? reinterpret_cast<const lom_map *>(s)->last_error.c_str() : ""; }
int lom_scan_set_option(lom_scan *s, int option, int64_t value) { return lom_map_set_option(reinterpret_cast<lom_map *>(s), option, value); }
int lom_scan_set_stream(lom_scan *s, void *hip_stream) { return lom_map_set_stream(reinterpret_cast<lom_map *>(s), hip_stream); }
void *lom_scan_get_stream(lom_scan *s) { return lom_map_get_stream(reinterpret_cast<lom_map *>(s)); }
int lom_scan_align(lom_scan *s, const float *src, size_t n, size_t stride, const float guess_t[3], const float guess_q[4],
                   float out_t[3], float out_q[4], lom_align_stats *stats)
{
    return lom_match_align(reinterpret_cast<lom_map *>(s), src, n, stride, guess_t, guess_q, out_t, out_q, stats);
}
int lom_scan_align_device(lom_scan *s, const float *d_src, size_t n, size_t stride, const float guess_t[3],
                          const float guess_q[4], float out_t[3], float out_q[4], lom_align_stats *stats)
{
    return lom_match_align_device(reinterpret_cast<lom_map *>(s), d_src, n, stride, guess_t, guess_q, out_t, out_q, stats);
}
int lom_scan_align_repeat(lom_scan *s, const float *d_src, size_t n, size_t stride, const float guess_t[3],
                          const float guess_q[4], int reps, float out_t[3], float out_q[4], lom_align_stats *total)
{
    return lom_match_align_repeat(reinterpret_cast<lom_map *>(s), d_src, n, stride, guess_t, guess_q, reps, out_t, out_q, total);
}
int64_t lom_scan_find_pairs(lom_scan *s, const float *src, size_t n, size_t stride, const float t[3], const float q[4],
                            float max_dist, lom_correspondence *out)
{
    return lom_match_find_pairs(reinterpret_cast<lom_map *>(s), src, n, stride, t, q, max_dist, out);
}
int64_t lom_scan_find_pairs_sq(lom_scan *s, const float *src, size_t n, size_t stride, const float t[3], const float q[4],
                               double max_dist_sq, lom_correspondence *out)
{
    return lom_match_find_pairs_sq(reinterpret_cast<lom_map *>(s), src, n, stride, t, q, max_dist_sq, out);
}

int lom_map_create(float voxel_size, size_t max_points, size_t capacity_hint, int device, lom_map **out)
{
    if (!out) return LOM_ERR_ARG;
    *out = nullptr;
    if (!(voxel_size > 0.f) || max_points == 0 || max_points > 65535)
        return set_error(nullptr, LOM_ERR_ARG, "voxel_size must be > 0 and 1 <= max_points <= 65535");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return set_error(nullptr, LOM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    }
    if (device < 0 || device >= ndev) return set_error(nullptr, LOM_ERR_ARG, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return set_error(nullptr, LOM_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string s = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return set_error(nullptr, LOM_ERR_NO_DEVICE, s.c_str());
    }
    lom_map *m = new (std::nothrow) lom_map();
    if (!m) return set_error(nullptr, LOM_ERR_OOM, "host allocation");
    m->device = device;
    m->voxel_size = voxel_size;
    m->K = (uint32_t)max_points;
    m->max_points = m->K;
    // the environment is looked at here and nowhere on the align path (lom_map_set_option changes the switches later)
    m->opt_host_lm = getenv("LOM_HOST_LM") != nullptr;
    if (const char *e = getenv("LOM_TABLE_SLOTS_PER_VOXEL")) m->table_slots_per_voxel = (uint32_t)std::min(256, std::max(2, atoi(e)));
    m->opt_debug_lm = getenv("LOM_DEBUG_LM") != nullptr;
    m->opt_debug_lm_twice = getenv("LOM_DEBUG_LM_TWICE") != nullptr;
    m->opt_debug_timing = getenv("LOM_DEBUG_TIMING") != nullptr;
    m->opt_no_temporal = getenv("LOM_NO_TEMPORAL") != nullptr;
    if (const char *e = getenv("LOM_COUNT_CANDIDATES")) m->opt_count = atoi(e) != 0;
    m->opt_no_bulk = getenv("LOM_NO_BULK_INSERT") != nullptr;
    m->opt_dense_cleanup = getenv("LOM_DENSE_CLEANUP") != nullptr;
    if (const char *e = getenv("LOM_BULK_PPT")) m->bulk_ppt = (uint32_t)atoi(e);  // development: points per thread of k_bi_claim
    if (handle_setup(m) != LOM_OK) {
        lom_map_destroy(m);
        return LOM_ERR_HIP;
    }
    m->min_cap = next_pow2(4ull * std::max<size_t>(capacity_hint, 256));
    // status / counter words (256 bytes) + the block aggregates of the single-pass kernels (256 x 2 granules)
    int rc = ensure(m, m->scr[S_MISC], 256 + 256 * 2 * sizeof(Granule));
    if (rc == LOM_OK) {
        if (hipMemsetAsync(m->scr[S_MISC].p, 0, m->scr[S_MISC].bytes, m->stream) != hipSuccess) rc = LOM_ERR_HIP;
    }
    if (rc == LOM_OK) rc = table_alloc(m, m->min_cap, &m->d_table);
    if (rc == LOM_OK) m->cap = m->min_cap;
    if (rc == LOM_OK && hipStreamSynchronize(m->stream) != hipSuccess) rc = LOM_ERR_HIP;
    if (rc != LOM_OK) {
        g_create_error = m->last_error.empty() ? "map setup failed" : m->last_error;
        lom_map_destroy(m);
        return rc;
    }
    *out = m;
    return LOM_OK;
}

void lom_map_destroy(lom_map *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->server_alive && m->h_cmd) {  // a resident evaluation server leaves on op = 2 (stop)
        unsigned long long *w = reinterpret_cast<unsigned long long *>(m->h_cmd);
        reinterpret_cast<unsigned int *>(w + 1)[0] = 2u;
        __atomic_store_n(w, ++m->mail_seq, __ATOMIC_RELEASE);
        m->server_alive = false;
    }
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm || m->host_comm) lom_comm_finalize(m);
    if (m->d_table) (void)hipFree(m->d_table);
    Slabs s{m->d_slab_key, m->d_slab_count, m->d_pts, m->d_nrm};
    slabs_free(s);
    Slabs alt{m->alt_key, m->alt_count, m->alt_pts, m->alt_nrm};
    slabs_free(alt);
    for (auto &b : m->scr)
        if (b.p) (void)hipFree(b.p);
    for (DeviceBuf *b : {&m->scan_src, &m->scan_idx, &m->scan_on, &m->scan_stats, &m->partials, &m->results, &m->gather,
                         &m->align_state, &m->xrec, &m->dbg_trace, &m->dbg_stamps})
        if (b->p) (void)hipFree(b->p);
    if (m->h_results) (void)hipHostFree(m->h_results);
    if (m->h_flags) (void)hipHostFree(m->h_flags);
    if (m->h_mail) (void)hipHostFree(m->h_mail);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->stage_ev) (void)hipEventDestroy(m->stage_ev);
    if (m->parent_ev) (void)hipEventDestroy(m->parent_ev);
    if (m->h_cmd) (void)hipHostFree(m->h_cmd);
    if (m->h_report) (void)hipHostFree(m->h_report);
    for (auto &e : m->prof_events)
        if (e) (void)hipEventDestroy(e);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

int lom_map_set_stream(lom_map *m, void *hip_stream)
{
    if (!m) return LOM_ERR_ARG;
    (void)hipSetDevice(m->device);
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    m->stream = hip_stream ? (hipStream_t)hip_stream : m->own_stream;
    return LOM_OK;
}

int lom_map_set_profiling(lom_map *m, int period)
{
    if (!m || period < 0) return LOM_ERR_ARG;
    m->profile_period = period;
    m->profiling = false;
    m->align_count = 0;
    return LOM_OK;
}

int lom_map_set_option(lom_map *m, int option, int64_t value)
{
    if (!m) return LOM_ERR_ARG;
    switch (option) {
    case LOM_OPT_HOST_LM: m->opt_host_lm = value != 0; return LOM_OK;
    case LOM_OPT_DEVICE_PATIENCE_TICKS:
        if (value < 1) return LOM_ERR_ARG;
        m->patience_ticks = (unsigned long long)value;
        return LOM_OK;
    case LOM_OPT_DEBUG_LM_STAMPS: m->opt_debug_lm = value != 0; return LOM_OK;
    case LOM_OPT_DEBUG_TIMING: m->opt_debug_timing = value != 0; return LOM_OK;
    case LOM_OPT_NO_TEMPORAL_BOUND: m->opt_no_temporal = value != 0; return LOM_OK;
    case LOM_OPT_COUNT_CANDIDATES: m->opt_count = value != 0; return LOM_OK;
    case LOM_OPT_NO_BULK_INSERT: m->opt_no_bulk = value != 0; return LOM_OK;
    case LOM_OPT_TEST_BULK_PARTITION_MAX:
        if (value < 0 || value > (int64_t)kBiPartMax) return LOM_ERR_ARG;
        m->test_bulk_part_max = (uint32_t)value;
        return LOM_OK;
    case LOM_OPT_TEST_GIVE_UP_AT_OUTER:
        if (value < -1 || value >= 35) return LOM_ERR_ARG;
        m->test_give_up_outer = (int)value;
        return LOM_OK;
    case LOM_OPT_TEST_GRID_GIVE_UP:
        if (value < -1 || value >= (1 << 20)) return LOM_ERR_ARG;
        m->test_grid_give_up = (int)value;
        return LOM_OK;
    default: return set_error(m, LOM_ERR_ARG, "unknown option");
    }
}

int64_t lom_map_debug_counter(const lom_map *m, int which)
{
    if (!m) return LOM_ERR_ARG;
    if (which == LOM_COUNTER_GRID_REDOS) return (int64_t)m->grid_redos;
    if (which == LOM_COUNTER_CLEANUPS_BEHIND_ALIGN) return (int64_t)m->cleanups_taken;
    if (which == LOM_COUNTER_EMPTY_SLABS) return (int64_t)m->n_dead;
    return LOM_ERR_ARG;
}

int lom_map_clear(lom_map *m, float voxel_size)
{
    if (!m || !(voxel_size > 0.f)) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    m->voxel_size = voxel_size;
    m->n_vox = 0;
    m->n_dead = 0;
    m->n_vox_ub = 0;
    m->n_vox_stale = false;
    m->n_points = 0;
    m->pending_n = 0;
    m->mutations++;
    if (!m->table_clean) {
        hipLaunchKernelGGL(k_table_init, dim3(blocks_for(m->cap)), dim3(kThreads), 0, m->stream, m->d_table, m->cap);
        LOM_HIP(m, hipMemsetAsync(d_nvox(m), 0, 4, m->stream));
        LOM_HIP(m, hipGetLastError());
        m->table_clean = true;
    }
    return LOM_OK;
}

// rows of every live slab from stride K0 to stride K1 > K0 (setMaxPoints raised on a map that holds voxels)
__global__ void k_restride(const float *pts0, const float *nrm0, const uint32_t *slab_count, uint32_t n_vox, uint32_t K0,
                           uint32_t K1, float *pts1, float *nrm1)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slab = (uint32_t)(i / K0), row = (uint32_t)(i % K0);
    if (slab >= n_vox || row >= slab_count[slab]) return;
    const size_t src = ((size_t)slab * K0 + row) * 3, dst = ((size_t)slab * K1 + row) * 3;
    store3(pts1 + dst, load3(pts0 + src));
    store3(nrm1 + dst, load3(nrm0 + src));
}

int lom_map_set_max_points(lom_map *m, size_t max_points)
{
    if (!m || max_points == 0 || max_points > 65535) return LOM_ERR_ARG;
    if (m->parent) return set_error(m, LOM_ERR_ARG, "a scan context cannot change its keyframe");
    LOM_HIP(m, hipSetDevice(m->device));
    {
        const int rcn = refresh_nvox(m);  // also resolves a pending insert: it was made under the old value
        if (rcn != LOM_OK) return rcn;
    }
    // voxel_grid.h:56-59: max_points_ = max_points, nothing else -- stored voxels keep what they hold, and :86-90
    // appends to a voxel only while size() < max_points_.  The row stride K follows the largest value seen while
    // voxels exist (a raise re-strides the slabs); an empty map starts over with stride = max_points.
    if (m->n_vox == 0 && max_points != m->K) {
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        Slabs s{m->d_slab_key, m->d_slab_count, m->d_pts, m->d_nrm};
        slabs_free(s);
        Slabs alt{m->alt_key, m->alt_count, m->alt_pts, m->alt_nrm};
        slabs_free(alt);
        m->alt_key = nullptr;
        m->alt_count = nullptr;
        m->alt_pts = m->alt_nrm = nullptr;
        m->alt_cap = 0;
        m->d_slab_key = nullptr;
        m->d_slab_count = nullptr;
        m->d_pts = m->d_nrm = nullptr;
        m->slab_cap = 0;
        m->K = (uint32_t)max_points;
    } else if (max_points > m->K) {
        if ((uint64_t)m->slab_cap > 0x7FFFFFFFull / max_points) return set_error(m, LOM_ERR_OOM, "map too large");
        m->mutations++;
        const uint32_t K0 = m->K, K1 = (uint32_t)max_points;
        const size_t pb = (size_t)m->slab_cap * K1 * 3 * sizeof(float);
        float *p1 = nullptr, *n1 = nullptr;
        if (hipMalloc(&p1, pb + kRowPadBytes) != hipSuccess || hipMalloc(&n1, pb + kRowPadBytes) != hipSuccess) {
            (void)hipGetLastError();
            if (p1) (void)hipFree(p1);
            return set_error(m, LOM_ERR_OOM, "hipMalloc(slabs)");
        }
        const size_t work = (size_t)m->n_vox * K0;
        hipLaunchKernelGGL(k_restride, dim3(blocks_for(work)), dim3(kThreads), 0, m->stream, m->d_pts, m->d_nrm,
                           m->d_slab_count, m->n_vox, K0, K1, p1, n1);
        LOM_HIP(m, hipGetLastError());
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        (void)hipFree(m->d_pts);
        (void)hipFree(m->d_nrm);
        m->d_pts = p1;
        m->d_nrm = n1;
        // the cleanup's second set of slabs has the old stride: it is allocated again when needed
        Slabs alt{m->alt_key, m->alt_count, m->alt_pts, m->alt_nrm};
        slabs_free(alt);
        m->alt_key = nullptr;
        m->alt_count = nullptr;
        m->alt_pts = m->alt_nrm = nullptr;
        m->alt_cap = 0;
        m->K = K1;
    }
    m->max_points = (uint32_t)max_points;
    return LOM_OK;
}

int lom_map_add_points_device(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n, size_t stride)
{
    if (!m || (n && !d_xyz) || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    return add_points_device(m, (const char *)d_xyz, (const char *)d_nrm, n, stride, false, true);
}

int lom_map_add_points_device_nowait(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n, size_t stride)
{
    if (!m || (n && !d_xyz) || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    return add_points_device(m, (const char *)d_xyz, (const char *)d_nrm, n, stride, false, false);
}

int lom_profile_insert(lom_map *m, const float *d_xyz, const float *d_nrm, size_t n, size_t stride, double *total_us_out)
{
    if (!m || !d_xyz || !n || !total_us_out || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    // capacity and scratch first, so that the bracket holds the insert's kernels only
    int rc = refresh_nvox(m);
    if (rc != LOM_OK) return rc;
    const uint64_t worst = (uint64_t)m->n_vox + n;
    if ((uint64_t)m->cap < 2 * worst && (rc = rehash(m, next_pow2(4 * worst))) != LOM_OK) return rc;
    if (worst > m->slab_cap && (rc = ensure_slabs(m, worst + worst / 2)) != LOM_OK) return rc;
    if (n > kOnePassMax && n <= kBiMaxPoints && !m->opt_no_bulk &&
        (rc = bulk_scratch(m, (uint32_t)n, bulk_shape((uint32_t)n, m->cap, m->bulk_ppt))) != LOM_OK)
        return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    LOM_HIP(m, hipEventCreate(&e0));
    LOM_HIP(m, hipEventCreate(&e1));
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    hipError_t e = hipEventRecord(e0, m->stream);
    rc = add_points_device(m, (const char *)d_xyz, (const char *)d_nrm, n, stride, false, false, false);
    if (e == hipSuccess) e = hipEventRecord(e1, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != LOM_OK) return rc;
    if (e != hipSuccess) return set_error(m, LOM_ERR_HIP, "lom_profile_insert", e);
    *total_us_out = (double)ms * 1e3;
    if ((rc = map_status(m)) != LOM_OK) return rc;
    // what lom_map_add_points_device does after a bulk insert: ~16 slots per voxel
    const uint32_t target = std::max(m->min_cap, next_pow2((uint64_t)m->table_slots_per_voxel * m->n_vox));
    if (m->cap > target) return rehash(m, target);
    return LOM_OK;
}

int lom_map_status(lom_map *m)
{
    if (!m) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    return map_status(m);
}

int lom_map_add_points(lom_map *m, const float *xyz, const float *nrm, size_t n, size_t stride)
{
    if (!m || (n && !xyz) || stride < 12 || (stride & 3)) return LOM_ERR_ARG;
    if (n == 0) return LOM_OK;
    LOM_HIP(m, hipSetDevice(m->device));
    // the points are in host memory: range-check them here (same f32 division and bounds as the
    // device's voxel_index) instead of paying a kernel and a synchronisation
    {
        const float vs = m->voxel_size;
        bool bad = false;
        for (size_t i = 0; i < n; i++) {
            const float *p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) + i * stride);
            const float fx = p[0] / vs, fy = p[1] / vs, fz = p[2] / vs;
            bad |= !(fx > -kIdxLimit && fx < kIdxLimit) || !(fy > -kIdxLimit && fy < kIdxLimit) ||
                   !(fz > -kIdxLimit && fz < kIdxLimit);
        }
        if (bad) return set_error(m, LOM_ERR_RANGE, "coordinate / voxel_size out of range or not finite");
    }
    const char *dx = nullptr, *dn = nullptr;
    int rc = resolve_pending(m);  // before the staging buffers (a pending insert's input) are overwritten
    if (rc != LOM_OK) return rc;
    if ((rc = stage_host_points(m, xyz, nrm, n, stride, &dx, &dn)) != LOM_OK) return rc;
    // the caller's buffer has been copied into the pinned bounce buffer: no need to wait for the GPU
    return add_points_device(m, dx, dn, n, stride, true, false);
}

// the single-pass scan of a radius cleanup: flags, new slab numbers, number of voxels kept (words[4]); `from` as k_cleanup_scan's
static bool launch_cleanup_scan(lom_map *m, uint32_t nv, const float center[3], float r2, uint32_t seq, const AlignState *from)
{
    uint32_t *keep = (uint32_t *)m->scr[S_FLAG].p, *newid = (uint32_t *)m->scr[S_RANK].p;
    auto launch_scan = [&](auto items) {
        constexpr int kItems = decltype(items)::value;
        hipLaunchKernelGGL((k_cleanup_scan<kItems>), dim3(blocks_for((nv + kItems - 1) / kItems)), dim3(kThreads), 0,
                           m->stream, m->d_pts, m->d_slab_count, m->K, nv, center[0], center[1], center[2], r2, keep, newid, d_agg(m), seq,
                           d_word(m, 0), take_test_fail_from(m), from);
    };
    if (nv <= kOnePassMax)
        launch_scan(std::integral_constant<int, 1>());
    else if (nv <= 4 * kOnePassMax)
        launch_scan(std::integral_constant<int, 4>());
    else if (nv <= 16 * kOnePassMax)
        launch_scan(std::integral_constant<int, 16>());
    else
        return false;
    return true;
}

// lidar_odometry.cpp:65-67 calls radiusCleanup with the translation the align has just produced: the scan of that cleanup
// only reads the map and writes scratch, so it can run right behind the align's last solve -- with the centre taken from
// the align's state in HBM -- instead of a host round trip, a thread hand-off and a launch later.  The caller arms it
// (lom_map_radius_cleanup_after_align), the next device-resident align on the handle enqueues scan and read-back behind
// its first five (k_match, k_lm) pairs (match.hip), and lom_map_radius_cleanup takes the result if, and only if, it was
// made for exactly its arguments on exactly this state of the map; everything else is the plain path below.
constexpr size_t kSpecWordsOffset = 768;  // of h_report / d_report: the words of a scan enqueued behind an align
constexpr int kSpecWords = 6;             // words 4 (kept), 7 (scan gave up), 12 (made for this call), 13..15 (centre used)

int lom_map_radius_cleanup_after_align(lom_map *m, float radius)
{
    if (!m) return LOM_ERR_ARG;
    m->spec_radius = (radius > 0.f && !m->parent) ? radius : 0.f;
    return LOM_OK;
}

}  // extern "C"
namespace lom {
void cleanup_scan_behind_align(lom_map *m)
{
    const float radius = m->spec_radius;
    m->spec_radius = 0.f;  // armed for one align
    // (a scan nobody has asked for since -- the caller did something else with the map -- is simply superseded: the
    // read-back of this one follows it on the stream and carries the next tag)
    m->spec_inflight = false;
    if (!(radius > 0.f) || m->parent || m->n_vox_stale || m->pending_n.load() || m->n_vox == 0 || !m->align_state.p) return;
    const uint32_t nv = m->n_vox;
    // (scratch that has to grow: the plain path does that; an allocation here would wait for the align)
    if (m->scr[S_FLAG].bytes < (size_t)nv * 4 || m->scr[S_RANK].bytes < (size_t)nv * 4 || nv > 16 * kOnePassMax) return;
    const float zero[3] = {0.f, 0.f, 0.f};
    const uint32_t seq = ++m->call_seq;
    if (!launch_cleanup_scan(m, nv, zero, radius * radius, seq, (const AlignState *)m->align_state.p)) return;
    WordPtrs w;
    const int idx[kSpecWords] = {4, 7, 12, 13, 14, 15};
    for (int i = 0; i < 32; i++) w.p[i] = i < kSpecWords ? d_word(m, idx[i]) : nullptr;
    if (++m->spec_tag == 0) m->spec_tag = 1;
    hipLaunchKernelGGL(k_gather_words, dim3(1), dim3(64), 0, m->stream, w, kSpecWords,
                       reinterpret_cast<unsigned long long *>((char *)m->d_report + kSpecWordsOffset), m->spec_tag);
    if (hipGetLastError() != hipSuccess) return;  // (nothing in flight that anybody will wait for)
    m->spec_inflight = true;
    m->spec_seq = seq;
    m->spec_nv = nv;
    m->spec_r = radius;
    m->spec_mutations = m->mutations.load();
}
}  // namespace lom
extern "C" {

// the result of a scan enqueued behind an align, if it was made for this call: 1 = h_flags[0] (kept) and h_flags[3]
// (give-up word) are set as read_words(m, 4, 4) would have, *seq_out = the scan's sequence number; 0 = not usable
static int take_cleanup_behind_align(lom_map *m, const float center[3], float radius, uint32_t *seq_out)
{
    if (!m->spec_inflight) return 0;
    m->spec_inflight = false;
    volatile unsigned long long *hw = reinterpret_cast<volatile unsigned long long *>((char *)m->h_report + kSpecWordsOffset);
    uint32_t got[kSpecWords];
    uint64_t spins = 0;
    for (int i = 0; i < kSpecWords; i++) {
        while ((uint32_t)(hw[i] >> 32) != m->spec_tag) {
            __builtin_ia32_pause();
            if ((++spins & 0x3FFF) != 0) continue;
            const hipError_t e = hipStreamQuery(m->stream);
            if (e == hipSuccess) {
                if ((uint32_t)(hw[i] >> 32) == m->spec_tag) break;
                return 0;
            }
            if (e != hipErrorNotReady) return 0;  // (the plain path meets the same stream and reports it)
        }
        got[i] = (uint32_t)hw[i];
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    uint32_t cb[3];
    std::memcpy(cb, center, 12);
    const bool usable = got[2] == m->spec_seq && got[3] == cb[0] && got[4] == cb[1] && got[5] == cb[2] &&
                        std::memcmp(&radius, &m->spec_r, 4) == 0 && m->call_seq == m->spec_seq &&
                        m->mutations.load() == m->spec_mutations && m->n_vox == m->spec_nv && !m->n_vox_stale;
    if (!usable) return 0;
    m->h_flags[0] = got[0];
    m->h_flags[3] = got[1];
    *seq_out = m->spec_seq;
    return 1;
}

int lom_map_radius_cleanup(lom_map *m, const float center[3], float radius)
{
    if (!m || !center) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    int rc;
    if ((rc = refresh_nvox(m)) != LOM_OK) return rc;
    uint32_t seq = 0;
    const bool taken = take_cleanup_behind_align(m, center, radius, &seq) == 1;
    if (m->n_vox == 0) return LOM_OK;
    const uint32_t nv = m->n_vox;
    // (twice what this call needs: a scan enqueued behind an align does not allocate, a growing keyframe should not
    // outgrow the scratch every few frames)
    // (a scan that has been taken left its flags and new slab numbers in these two: they stay where they are)
    if (!taken) {
        if ((rc = ensure(m, m->scr[S_FLAG], (size_t)nv * 8)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_RANK], (size_t)nv * 8)) != LOM_OK) return rc;
    }
    if ((rc = ensure(m, m->scr[S_SCAN], scan_tmp_words(nv) * 4)) != LOM_OK) return rc;
    uint32_t *keep = (uint32_t *)m->scr[S_FLAG].p, *newid = (uint32_t *)m->scr[S_RANK].p;
    const float r2 = radius * radius;  // voxel_grid.h:238
    if (!taken) seq = ++m->call_seq;
    m->mutations++;
    bool one_pass = true;
    if (!taken) {
        one_pass = launch_cleanup_scan(m, nv, center, r2, seq, nullptr);
        if (!one_pass) {
            hipLaunchKernelGGL(k_cleanup_flag, dim3(blocks_for(nv)), dim3(kThreads), 0, m->stream, m->d_pts, m->d_slab_count, m->K, nv,
                               center[0], center[1], center[2], r2, keep);
            LOM_HIP(m, hipGetLastError());
            if ((rc = scan_exclusive(m, keep, newid, nv, d_word(m, 4), (uint32_t *)m->scr[S_SCAN].p)) != LOM_OK) return rc;
        }
        LOM_HIP(m, hipGetLastError());
        if ((rc = read_words(m, 4, 4)) != LOM_OK) return rc;
    } else {
        m->cleanups_taken++;
    }
    if (one_pass && m->h_flags[3] == seq) {
        // the in-kernel scan gave up (it has written scratch only): flags + multi-launch scan instead
        m->grid_redos++;
        m->status_seq = std::max(m->status_seq, seq);
        hipLaunchKernelGGL(k_cleanup_flag, dim3(blocks_for(nv)), dim3(kThreads), 0, m->stream, m->d_pts, m->d_slab_count, m->K, nv,
                           center[0], center[1], center[2], r2, keep);
        LOM_HIP(m, hipGetLastError());
        if ((rc = scan_exclusive(m, keep, newid, nv, d_word(m, 4), (uint32_t *)m->scr[S_SCAN].p)) != LOM_OK) return rc;
        if ((rc = read_words(m, 4, 1)) != LOM_OK) return rc;
    }
    const uint32_t n_keep = m->h_flags[0], n_live = nv - m->n_dead;
    if (n_keep == n_live) return LOM_OK;
    if (!m->opt_dense_cleanup && (uint64_t)(nv - n_keep) * 4u <= (uint64_t)nv) {
        // few holes: the erased voxels' slabs stay where they are, empty (see k_cleanup_mark)
        const MapView v = view_of(m);
        hipLaunchKernelGGL(k_cleanup_mark, dim3(blocks_for(nv)), dim3(kThreads), 0, m->stream, m->d_table, v.mask, v.shift, keep, nv,
                           m->d_slab_key, m->d_slab_count);
        LOM_HIP(m, hipGetLastError());
        m->n_dead = nv - n_keep;
        m->dead_below = nv;
        return LOM_OK;
    }
    // stable compaction into the second (persistent) set of slab arrays, swap, rebuild the table
    if (m->alt_cap != m->slab_cap) {
        Slabs stale{m->alt_key, m->alt_count, m->alt_pts, m->alt_nrm};
        LOM_HIP(m, hipStreamSynchronize(m->stream));
        slabs_free(stale);
        m->alt_key = nullptr;
        m->alt_count = nullptr;
        m->alt_pts = m->alt_nrm = nullptr;
        m->alt_cap = 0;
        Slabs fresh;
        if ((rc = slabs_alloc(m, m->slab_cap, fresh)) != LOM_OK) return rc;
        m->alt_key = fresh.key;
        m->alt_count = fresh.count;
        m->alt_pts = fresh.pts;
        m->alt_nrm = fresh.nrm;
        m->alt_cap = m->slab_cap;
    }
    const size_t work = (size_t)nv * m->K;
    hipLaunchKernelGGL(k_compact, dim3(blocks_for(work)), dim3(kThreads), 0, m->stream, keep, newid, nv, m->K,
                       m->d_slab_key, m->d_slab_count, m->d_pts, m->d_nrm, m->alt_key, m->alt_count, m->alt_pts,
                       m->alt_nrm, d_nvox(m), n_keep);
    LOM_HIP(m, hipGetLastError());
    std::swap(m->d_slab_key, m->alt_key);
    std::swap(m->d_slab_count, m->alt_count);
    std::swap(m->d_pts, m->alt_pts);
    std::swap(m->d_nrm, m->alt_nrm);
    m->n_vox = n_keep;
    m->n_vox_ub = n_keep;
    m->n_dead = 0;  // (the holes are closed)
    const MapView v = view_of(m);
    hipLaunchKernelGGL(k_table_init, dim3(blocks_for(m->cap)), dim3(kThreads), 0, m->stream, m->d_table, m->cap);
    if (n_keep) {
        hipLaunchKernelGGL(k_rebuild, dim3(blocks_for(n_keep)), dim3(kThreads), 0, m->stream, m->d_table, v.mask,
                           v.shift, m->d_slab_key, m->d_slab_count, n_keep);
    }
    LOM_HIP(m, hipGetLastError());
    return LOM_OK;
}

int64_t lom_map_size(const lom_map *cm)
{
    lom_map *m = const_cast<lom_map *>(cm);
    if (!m) return LOM_ERR_ARG;
    if (m->n_vox_stale || m->pending_n) {
        if (hipSetDevice(m->device) != hipSuccess) return LOM_ERR_HIP;
        const int rc = refresh_nvox(m);
        if (rc != LOM_OK) return rc;
    }
    return (int64_t)(m->n_vox - m->n_dead);  // (slabs in use minus those whose voxel a cleanup erased)
}

int64_t lom_map_point_count(const lom_map *cm)
{
    // the stored points are exactly what the full export would return
    return lom_map_export(const_cast<lom_map *>(cm), LOM_EXPORT_FULL_NO_NORMALS, nullptr, nullptr, 0);
}

// shared body of the down-samplers: device input, results left in the workspace's scratch
// (S_ITEMS: xyz, S_PT_POS: normals), voxel count in word 4, range flag in word 5 of S_MISC.
// wait: read the count and the verdict back (one synchronisation); otherwise the call only enqueues and
// the count stays on the device (d_word(m, 4)) for the kernels that consume the result.
static int downsample_core(lom_map *m, float voxel_size, const char *dx, const char *dn, uint32_t N, size_t stride,
                           bool want_normals, bool wait, const uint32_t *n_dev = nullptr, bool multi_launch = false)
{
    int rc;
    if ((uint64_t)m->cap < 2ull * N) {
        if ((rc = rehash(m, next_pow2(4ull * N))) != LOM_OK) return rc;
    }
    const bool one_pass = N <= 4 * kOnePassMax && !multi_launch;
    if ((rc = ensure(m, m->scr[S_PT_SLOT], (size_t)N * 4)) != LOM_OK) return rc;
    if ((rc = ensure_rest(m, m->scr[S_DS_HEAD], (size_t)m->cap * 4, 0xFF)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_ITEMS], (size_t)N * 12)) != LOM_OK) return rc;     // compacted xyz
    if ((rc = ensure(m, m->scr[S_PT_POS], (size_t)N * 12)) != LOM_OK) return rc;    // compacted normals
    uint32_t *pt_slot = (uint32_t *)m->scr[S_PT_SLOT].p, *head = (uint32_t *)m->scr[S_DS_HEAD].p;
    float *oxyz = (float *)m->scr[S_ITEMS].p, *onrm = want_normals ? (float *)m->scr[S_PT_POS].p : nullptr;
    const uint32_t seq = ++m->call_seq;
    const MapView v = view_of(m);
    const dim3 g(blocks_for(N)), b(kThreads);
    if (n_dev && !one_pass) return set_error(m, LOM_ERR_ARG, "device-side point count: at most 262144 points");
    hipLaunchKernelGGL(k_ds_claim, g, b, 0, m->stream, m->d_table, v.mask, v.shift, dx, stride, N, n_dev, voxel_size,
                       pt_slot, head, seq, d_word(m, 5));
    if (one_pass) {
        if (N <= kOnePassMax)
            hipLaunchKernelGGL(k_ds_emit<1>, g, b, 0, m->stream, m->d_table, N, n_dev, pt_slot, head, dx, dn, stride, oxyz,
                               onrm, d_agg(m), seq, d_word(m, 0), take_test_fail_from(m));
        else
            hipLaunchKernelGGL(k_ds_emit<4>, dim3(blocks_for((N + 3) / 4)), b, 0, m->stream, m->d_table, N, n_dev, pt_slot,
                               head, dx, dn, stride, oxyz, onrm, d_agg(m), seq, d_word(m, 0), take_test_fail_from(m));
        LOM_HIP(m, hipGetLastError());
    } else {
        if ((rc = ensure(m, m->scr[S_FLAG], (size_t)N * 4)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_RANK], (size_t)N * 4)) != LOM_OK) return rc;
        if ((rc = ensure(m, m->scr[S_SCAN], scan_tmp_words(N) * 4)) != LOM_OK) return rc;
        uint32_t *flag = (uint32_t *)m->scr[S_FLAG].p, *rank = (uint32_t *)m->scr[S_RANK].p;
        hipLaunchKernelGGL(k_ds_flag, g, b, 0, m->stream, N, pt_slot, head, flag);
        LOM_HIP(m, hipGetLastError());
        if ((rc = scan_exclusive(m, flag, rank, N, d_word(m, 4), (uint32_t *)m->scr[S_SCAN].p)) != LOM_OK) return rc;
        // the kept point of every voxel also frees its table slot and head word: the workspace is at rest again
        hipLaunchKernelGGL(k_ds_write, g, b, 0, m->stream, N, flag, rank, dx, dn, stride, oxyz, onrm, m->d_table, pt_slot,
                           head);
        LOM_HIP(m, hipGetLastError());
    }
    if (!wait) return LOM_OK;
    if ((rc = read_words(m, 4, 4)) != LOM_OK) return rc;  // [0] voxels, [1] range flag, [3] grid error
    m->status_seq = seq;
    if (m->h_flags[3] == seq) {
        // the in-kernel scan gave up: every kept point has still put its slot and head word back to rest, so the
        // workspace is empty again; same call through the flag / scan / write kernels, which wait for nobody
        if (n_dev) return set_error(m, LOM_ERR_HIP, "a workgroup timed out waiting for the others of its grid");
        m->grid_redos++;
        return downsample_core(m, voxel_size, dx, dn, N, stride, want_normals, true, nullptr, true);
    }
    if (m->h_flags[1] == seq) return set_error(m, LOM_ERR_RANGE, "coordinate / voxel_size out of range or not finite");
    return LOM_OK;
}

int64_t lom_voxel_downsample(lom_map *ws, float voxel_size, const float *xyz, const float *nrm, size_t n,
                             size_t stride, float *xyz_out, float *nrm_out, size_t cap)
{
    lom_map *m = ws;
    if (!m || !(voxel_size > 0.f) || (n && !xyz) || stride < 12 || (stride & 3) || (n && !xyz_out)) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    int rc = lom_map_clear(m, voxel_size);  // the workspace grid ends up cleared, like a fresh VoxelGrid(voxel, 1)
    if (rc != LOM_OK || n == 0) return rc;
    const char *dx = nullptr, *dn = nullptr;
    if ((rc = stage_host_points(m, xyz, nrm, n, stride, &dx, &dn)) != LOM_OK) return rc;
    // the range rule of addCloud is checked by the claim kernel (flag read back with the count)
    if ((rc = downsample_core(m, voxel_size, dx, dn, (uint32_t)n, stride, nrm_out != nullptr, true)) != LOM_OK) return rc;
    const size_t total = m->h_flags[0];
    const size_t take = std::min(total, cap);
    if (take) {
        LOM_HIP(m, hipMemcpyAsync(xyz_out, m->scr[S_ITEMS].p, take * 12, hipMemcpyDeviceToHost, m->stream));
        if (nrm_out)
            LOM_HIP(m, hipMemcpyAsync(nrm_out, m->scr[S_PT_POS].p, take * 12, hipMemcpyDeviceToHost, m->stream));
        LOM_HIP(m, hipStreamSynchronize(m->stream));
    }
    return (int64_t)total;
}

int64_t lom_voxel_downsample_device(lom_map *ws, float voxel_size, const float *d_xyz, const float *d_nrm, size_t n,
                                    size_t stride, const float **d_xyz_out, const float **d_nrm_out)
{
    lom_map *m = ws;
    if (!m || !(voxel_size > 0.f) || (n && !d_xyz) || stride < 12 || (stride & 3) || !d_xyz_out) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    *d_xyz_out = nullptr;
    if (d_nrm_out) *d_nrm_out = nullptr;
    int rc = lom_map_clear(m, voxel_size);
    if (rc != LOM_OK || n == 0) return rc;
    if ((rc = downsample_core(m, voxel_size, (const char *)d_xyz, (const char *)d_nrm, (uint32_t)n, stride,
                              d_nrm_out != nullptr, true)) != LOM_OK)
        return rc;
    *d_xyz_out = (const float *)m->scr[S_ITEMS].p;
    if (d_nrm_out) *d_nrm_out = (const float *)m->scr[S_PT_POS].p;
    return (int64_t)m->h_flags[0];
}

int lom_voxel_downsample_device_nowait(lom_map *ws, float voxel_size, const float *d_xyz, const float *d_nrm,
                                       size_t n_bound, const uint32_t *d_n, size_t stride, const float **d_xyz_out,
                                       const float **d_nrm_out, const uint32_t **d_count_out)
{
    lom_map *m = ws;
    if (!m || !(voxel_size > 0.f) || (n_bound && !d_xyz) || stride < 12 || (stride & 3) || !d_xyz_out || !d_count_out)
        return LOM_ERR_ARG;
    if (n_bound >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    *d_xyz_out = nullptr;
    if (d_nrm_out) *d_nrm_out = nullptr;
    int rc = lom_map_clear(m, voxel_size);
    if (rc != LOM_OK) return rc;
    *d_count_out = d_word(m, 4);
    if (n_bound == 0) {
        LOM_HIP(m, hipMemsetAsync(d_word(m, 4), 0, 4, m->stream));
        return LOM_OK;
    }
    if ((rc = downsample_core(m, voxel_size, (const char *)d_xyz, (const char *)d_nrm, (uint32_t)n_bound, stride,
                              d_nrm_out != nullptr, false, d_n)) != LOM_OK)
        return rc;
    *d_xyz_out = (const float *)m->scr[S_ITEMS].p;
    if (d_nrm_out) *d_nrm_out = (const float *)m->scr[S_PT_POS].p;
    return LOM_OK;
}

int lom_map_wait_event(lom_map *m, void *hip_event)
{
    if (!m || !hip_event) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    LOM_HIP(m, hipStreamWaitEvent(m->stream, (hipEvent_t)hip_event, 0));
    return LOM_OK;
}

int lom_map_status_words(lom_map *m, const uint32_t **d_range, const uint32_t **d_grid, uint32_t *seq)
{
    if (!m || !d_range || !d_grid || !seq) return LOM_ERR_ARG;
    *d_range = d_word(m, 5);
    *d_grid = d_word(m, 7);
    *seq = m->call_seq;
    m->status_seq = m->call_seq;  // the caller looks at the words itself
    return LOM_OK;
}

int lom_map_read_device_words(lom_map *m, const uint32_t *const *d_ptrs, int n, uint32_t *out)
{
    if (!m || !d_ptrs || !out || n < 0 || n > 32) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    for (int i = 0; i < n; i++)
        if (!d_ptrs[i]) return LOM_ERR_ARG;
    return gather_words(m, d_ptrs, n, out);
}

int lom_map_read_device_words_begin(lom_map *m, const uint32_t *const *d_ptrs, int n)
{
    if (!m || !d_ptrs || n < 0 || n > 32) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    for (int i = 0; i < n; i++)
        if (!d_ptrs[i]) return LOM_ERR_ARG;
    return gather_words_begin(m, d_ptrs, n);
}

int lom_map_read_device_words_end(lom_map *m, uint32_t *out)
{
    if (!m || !out) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    return gather_words_end(m, out);
}

int lom_upload_points(lom_map *m, const float *xyz, const float *nrm, size_t n, size_t stride, const float **d_xyz_out,
                      const float **d_nrm_out)
{
    if (!m || (n && !xyz) || stride < 12 || (stride & 3) || !d_xyz_out) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    *d_xyz_out = nullptr;
    if (d_nrm_out) *d_nrm_out = nullptr;
    if (n == 0) return LOM_OK;
    const char *dx = nullptr, *dn = nullptr;
    int rc = resolve_pending(m);
    if (rc != LOM_OK) return rc;
    if ((rc = stage_host_points(m, xyz, nrm, n, stride, &dx, &dn)) != LOM_OK) return rc;
    *d_xyz_out = (const float *)dx;
    if (d_nrm_out) *d_nrm_out = (const float *)dn;
    return LOM_OK;
}

int lom_transform_points_device(lom_map *m, const lom_pose *pose, const float *d_xyz, const float *d_nrm, size_t n,
                                size_t stride, const float **d_xyz_out, const float **d_nrm_out)
{
    if (!m || !pose || (n && !d_xyz) || stride < 12 || (stride & 3) || !d_xyz_out) return LOM_ERR_ARG;
    if (n >= 0x7FFFFFFFull) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    *d_xyz_out = nullptr;
    if (d_nrm_out) *d_nrm_out = nullptr;
    if (n == 0) return LOM_OK;
    int rc;
    if ((rc = resolve_pending(m)) != LOM_OK) return rc;
    const bool with_n = d_nrm && d_nrm_out;
    if ((rc = ensure(m, m->scr[S_IN_XYZ], n * 12)) != LOM_OK) return rc;
    if (with_n && (rc = ensure(m, m->scr[S_IN_NRM], n * 12)) != LOM_OK) return rc;
    RigidArgs A;
    rotation_matrix(pose->q, A.R);
    for (int i = 0; i < 3; i++) A.t[i] = pose->t[i];
    hipLaunchKernelGGL(k_transform, dim3(blocks_for((uint32_t)n)), dim3(kThreads), 0, m->stream, (const char *)d_xyz,
                       (const char *)d_nrm, stride, (uint32_t)n, A, (float *)m->scr[S_IN_XYZ].p,
                       with_n ? (float *)m->scr[S_IN_NRM].p : (float *)nullptr);
    LOM_HIP(m, hipGetLastError());
    *d_xyz_out = (const float *)m->scr[S_IN_XYZ].p;
    if (with_n) *d_nrm_out = (const float *)m->scr[S_IN_NRM].p;
    return LOM_OK;
}

void *lom_map_get_stream(lom_map *m) { return m ? (void *)m->stream : nullptr; }

int64_t lom_map_export(lom_map *m, int mode, float *xyz_out, float *nrm_out, size_t cap)
{
    if (!m || mode < 0 || mode > 2) return LOM_ERR_ARG;
    LOM_HIP(m, hipSetDevice(m->device));
    int rc;
    if ((rc = refresh_nvox(m)) != LOM_OK) return rc;
    if (m->n_vox == 0) return 0;
    const uint32_t nv = m->n_vox;
    if ((rc = ensure(m, m->scr[S_FLAG], (size_t)nv * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_RANK], (size_t)nv * 4)) != LOM_OK) return rc;
    if ((rc = ensure(m, m->scr[S_SCAN], scan_tmp_words(nv) * 4)) != LOM_OK) return rc;
    uint32_t *cnt = (uint32_t *)m->scr[S_FLAG].p, *off = (uint32_t *)m->scr[S_RANK].p;
    hipLaunchKernelGGL(k_export_counts, dim3(blocks_for(nv)), dim3(kThreads), 0, m->stream, m->d_slab_count, nv, mode, cnt);
    LOM_HIP(m, hipGetLastError());
    if ((rc = scan_exclusive(m, cnt, off, nv, d_word(m, 4), (uint32_t *)m->scr[S_SCAN].p)) != LOM_OK) return rc;
    if ((rc = read_words(m, 4, 1)) != LOM_OK) return rc;
    const size_t total = m->h_flags[0];
    if (!xyz_out || cap == 0) return (int64_t)total;
    const bool want_n = nrm_out && mode == LOM_EXPORT_FULL;
    if ((rc = ensure(m, m->scr[S_IN_XYZ], total * 12)) != LOM_OK) return rc;
    if (want_n && (rc = ensure(m, m->scr[S_IN_NRM], total * 12)) != LOM_OK) return rc;
    const size_t work = (size_t)nv * m->K;
    hipLaunchKernelGGL(k_export_write, dim3(blocks_for(work)), dim3(kThreads), 0, m->stream, off, m->d_slab_count, nv,
                       m->K, mode, m->d_pts, m->d_nrm, (float *)m->scr[S_IN_XYZ].p,
                       want_n ? (float *)m->scr[S_IN_NRM].p : (float *)nullptr);
    LOM_HIP(m, hipGetLastError());
    const size_t take = std::min(total, cap);
    LOM_HIP(m, hipMemcpyAsync(xyz_out, m->scr[S_IN_XYZ].p, take * 12, hipMemcpyDeviceToHost, m->stream));
    if (want_n) LOM_HIP(m, hipMemcpyAsync(nrm_out, m->scr[S_IN_NRM].p, take * 12, hipMemcpyDeviceToHost, m->stream));
    LOM_HIP(m, hipStreamSynchronize(m->stream));
    return (int64_t)total;
}

}  // extern "C"
