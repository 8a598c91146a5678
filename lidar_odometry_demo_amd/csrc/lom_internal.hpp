// Internal declarations shared by the HIP translation units.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lidar_odometry_amd.h"

namespace lom {

// ---- open-addressed voxel hash in HBM ---------------------------------------
// One 16-byte slot per voxel: a single dwordx4 load yields key, point count
// and payload slab (reference: robin_map<Indices, VoxelWithPlanes>,
// src/voxel_grid.h:256).  Payload lives in slabs [slab][K] of packed 12-byte
// points and, separately, 12-byte normals (normals are read once per query,
// for the winner only, so they stay out of the candidate stream).
struct __attribute__((aligned(16))) Slot {
    unsigned long long key;  // packed (ix,iy,iz), 21 bits each, biased; kEmptyKey = free
    uint32_t count;          // stored points (<= K)
    uint32_t slab;           // creation index of the voxel; kNoSlab until assigned
};
static_assert(sizeof(Slot) == 16, "slot must be one dwordx4");

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kNoSlab = 0xFFFFFFFFu;
constexpr int kIdxBias = 1 << 20;          // indices in (-2^20, 2^20)
constexpr float kIdxLimit = 1048576.0f;    // |x / voxel_size| must stay below

__host__ __device__ inline unsigned long long pack_key(int ix, int iy, int iz)
{
    return ((unsigned long long)(uint32_t)(ix + kIdxBias) << 42) |
           ((unsigned long long)(uint32_t)(iy + kIdxBias) << 21) |
           (unsigned long long)(uint32_t)(iz + kIdxBias);
}

// Fibonacci hashing: top log2(cap) bits of key * 2^64/phi.  The hash never
// influences results (reference IndicesHash, voxel_grid.h:31-38, likewise).
// (Round 3 tried a brick-local hash -- the 2x2x2 block of voxels sharing (ix >> 1, iy >> 1, iz >> 1) in ONE 128-byte
// bucket, so that a query's 27 slots lie in eight lines: half the L2 requests, but the bricks of one surface collide
// in whole parity patterns (chains 2.4 slots against 1.4 at equal load), the three per-axis hash parts have to travel
// through LDS, and at equal table size the probe phase of C2 took 2.4 us against 1.5: DESIGN.md section 5.)
__host__ __device__ inline uint32_t hash_key(unsigned long long key, uint32_t shift)
{
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift);
}

// voxel_grid.h:70-72 / :166-168: static_cast<int64_t>(x / voxel_size_), f32
// division (correctly rounded), truncation toward zero.
__device__ inline bool voxel_index(float x, float voxel_size, int &out)
{
    const float f = __fdiv_rn(x, voxel_size);
    if (!(f > -kIdxLimit && f < kIdxLimit)) return false;  // also NaN
    out = (int)f;
    return true;
}

// Same value, cheaper when the voxel size is a power of two (the reference's 0.5 / 1.0 / 0.25):
// x * 2^-k IS the correctly rounded quotient x / 2^k (an exact scaling; where it would go
// subnormal both truncate to 0).  inv == 0 selects the IEEE division.
__device__ inline bool voxel_index_fast(float x, float voxel_size, float inv, int &out)
{
    const float f = (inv != 0.f) ? x * inv : __fdiv_rn(x, voxel_size);
    if (!(f > -kIdxLimit && f < kIdxLimit)) return false;  // also NaN
    out = (int)f;
    return true;
}

struct MapView {
    const Slot *table;
    uint32_t mask;   // capacity - 1
    uint32_t shift;  // 64 - log2(capacity)
    const float *pts;  // [slab][K][3]
    const float *nrm;  // [slab][K][3]
    uint32_t K;
    float voxel_size;
    float inv_voxel_size;  // 1 / voxel_size when that is exact (power of two), else 0
    float prune_slack;     // 1e-4f * voxel_size (f32 product): the absolute slack of k_match's face-distance bounds, as a
                           // kernel argument so that it lives in a scalar register
};

// pose as the kernels consume it
struct PoseArgs {
    double R[9];  // f32 rotationMatrix() widened (voxel_grid.h:212)
    double t[3];  // f32 translation widened (voxel_grid.h:213)
    float max_sq; // max_correspondence_distance^2 in f32 (voxel_grid.h:215)
};

struct EvalArgs {
    double q[4];  // w,x,y,z
    double t[3];
};

// Single-GPU align: the outer loop's state lives in HBM between the kernels of one align.  k_lm's
// workgroup 0 writes it at the end of an outer iteration; the next k_match takes its pose from P,
// the next k_lm continues from pose_t / pose_q -- the host enqueues the kernels of several outer
// iterations without waiting for any of them.
struct AlignState {
    PoseArgs P;        // pose of the next correspondence search
    float pose_t[3];   // f32 pose after the write-back (cloud_matcher.cpp:161-167)
    float pose_q[4];
    int32_t finished;  // converged (:169-172) or 35 iterations done: later kernels of the chain do nothing
    int32_t error;     // a workgroup gave up waiting for the others
    int32_t outer_done;
    int32_t lm_iterations, evaluations;
    int32_t pad;
    double valid_last, valid_total, cand_total, occ_total, queries_total;
    double final_cost, last_step_norm;
};

// what the host reads back (pinned host memory): a copy of the scalar part of AlignState, written
// by k_lm's workgroup 0 after every outer iteration, sequence word last
struct AlignReport {
    unsigned long long seq;
    int32_t finished, error, outer_done, lm_iterations, evaluations, pad;
    float pose_t[3], pose_q[4];
    float pad2;
    double valid_last, valid_total, cand_total, occ_total, queries_total;
    double final_cost, last_step_norm;
};

// ---- host-side handle ---------------------------------------------------------
struct DeviceBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace lom

struct lom_map {
    int device = 0;
    // a scan context (lom_scan_create): no table or slabs of its own, its kernels read the parent's
    lom_map *parent = nullptr;
    // map side: calls that changed the map (their kernels run on THIS handle's stream); context side: the count its
    // stream has been ordered behind, and the event it uses for that
    // (atomic: a context's fast path reads the parent's count while another context, or the map's own caller, settles)
    std::atomic<uint64_t> mutations{0};
    uint64_t seen_mutations = ~0ull;
    hipEvent_t parent_ev = nullptr;
    // Settling the map -- redoing an insert nobody has looked at yet, reading its voxel count back -- uses the MAP's
    // stream, pinned words and scratch: whoever does it (the map's own caller, a context's first call after a change,
    // lom_scan_create) holds this lock, so that several threads arriving together settle once.
    std::mutex settle_mutex;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // compute units this handle's own stream runs on: 0 = all of the device; a scan context on a partition
    // (lom_scan_create_on_partition) owns a stream with a CU mask, and its grids are sized for that many CUs
    uint32_t partition_cus = 0;
    float voxel_size = 0.5f;
    uint32_t K = 10;          // row stride of the slabs: the largest max_points_ the stored voxels have seen
    uint32_t max_points = 10; // max_points_ (voxel_grid.h:253): what an insert fills a voxel up to; <= K

    // hash table
    lom::Slot *d_table = nullptr;
    uint32_t cap = 0;  // power of two
    uint32_t min_cap = 0;

    // slabs (creation order).  n_vox is the host's copy of the device-side voxel counter: an insert
    // whose worst case fits the allocated slabs does not wait for the GPU, it only raises the upper
    // bound n_vox_ub and marks n_vox stale; whoever needs the exact value calls refresh_nvox().
    uint32_t n_vox = 0;
    uint32_t n_vox_ub = 0;
    bool n_vox_stale = false;
    // pinned bounce buffer for host-resident inputs (truly asynchronous H2D) + "last copy done" event
    void *h_stage = nullptr;
    size_t h_stage_bytes = 0;
    hipEvent_t stage_ev = nullptr;
    uint32_t slab_cap = 0;
    unsigned long long *d_slab_key = nullptr;  // [slab_cap]
    uint32_t *d_slab_count = nullptr;          // [slab_cap]
    float *d_pts = nullptr;                    // [slab_cap][K][3]
    float *d_nrm = nullptr;                    // [slab_cap][K][3]
    uint64_t n_points = 0;
    // second slab set: radiusCleanup compacts into it and swaps (no allocation per frame)
    unsigned long long *alt_key = nullptr;
    uint32_t *alt_count = nullptr;
    float *alt_pts = nullptr, *alt_nrm = nullptr;
    uint32_t alt_cap = 0;

    // scratch (grown on demand, never shrunk)
    lom::DeviceBuf scr[24];
    // sequence number of the last map-maintenance call (tags block aggregates and error words: no resets),
    // the one up to which lom_map_status has looked, and whether the table is known to be all-empty
    uint32_t call_seq = 0, status_seq = 0;
    bool table_clean = true;
    // the handle's last single-pass insert, for lom_map_status() to redo should its in-kernel scan have given up
    const char *pending_xyz = nullptr, *pending_nrm = nullptr;
    std::atomic<size_t> pending_n{0};  // != 0: an unverified single-pass insert (read without the lock by the fast paths)
    size_t pending_stride = 0;
    uint32_t pending_seq = 0, grid_resolved_seq = 0;
    uint32_t grid_redos = 0;  // calls redone with the multi-launch scan after a give-up (lom_map_debug_counter)
    // per-scan buffers of align/find_pairs
    lom::DeviceBuf scan_src, scan_idx, scan_on, scan_stats, partials, results;

    // pinned host result buffer
    double *h_results = nullptr;  // 1024 doubles (rank-ordered gather of up to 32 ranks)
    uint32_t *h_flags = nullptr;  // 64 words
    // mailbox: every k_eval workgroup stores one 32-double record (sums, counters,
    // sequence word) directly into coherent (fine-grained) pinned host memory; the host
    // polls the sequence words instead of paying a copy kernel plus
    // hipStreamSynchronize per residual evaluation.
    double *h_mail = nullptr;             // host view: 64 records x 32 doubles
    double *d_mail = nullptr;             // device view of the same allocation
    unsigned long long mail_seq = 0;
    // command word of the resident evaluation server (pinned host memory, host writes, device polls)
    void *h_cmd = nullptr, *d_cmd = nullptr;
    bool server_alive = false;
    bool eval_attr_set = false;
    // device-resident outer loop (single GPU): state in HBM, exchange records of k_lm's workgroups,
    // report in pinned host memory
    lom::DeviceBuf align_state, xrec, dbg_trace, dbg_stamps;
    bool align_state_dirty = true;  // AlignState must be zeroed before the next chain (fresh, or left with its error flag set)
    void *h_report = nullptr, *d_report = nullptr;  // 1 KiB: AlignReport, and at 512 the words of gather_words()
    uint32_t words_tag = 0;
    int words_pending = 0;  // words of a lom_map_read_device_words_begin not yet collected
    uint32_t n_dead = 0;  // slabs below n_vox whose voxel a radius cleanup erased without moving the others (k_cleanup_mark)
    uint32_t dead_below = 0;  // ... all of them below this slab number (the slab count at the last such cleanup)
    bool opt_dense_cleanup = false;  // LOM_DENSE_CLEANUP=1 at create: every radius cleanup closes its holes at once (as until round 4)
    void (*idle_hook)(void *) = nullptr;  // lom_map_set_align_idle_hook: one shot
    void *idle_user = nullptr;
    // the scan of a radius cleanup enqueued behind an align (lom_map_radius_cleanup_after_align, voxel_map.hip)
    float spec_radius = 0.f;     // > 0: the next device-resident align on this handle enqueues it
    bool spec_inflight = false;  // scan and read-back are on the stream; what they were made for:
    float spec_r = 0.f;
    uint32_t spec_seq = 0, spec_tag = 0, spec_nv = 0;
    uint64_t spec_mutations = 0;
    uint32_t cleanups_taken = 0;  // radius cleanups that used such a scan (lom_map_debug_counter)
    unsigned long long report_seq = 0, lm_seq = 0, lm_launches = 0;
    uint32_t lm_max_blocks[4] = {0, 0, 0, 0};  // co-resident k_lm workgroups this device admits, per variant (occupancy query, cached)
    double last_counters[4] = {0, 0, 0, 0};  // valid, cand, occ, queries of the last k_match

    bool profiling = false;       // this align carries event pairs
    int profile_period = 0;       // 0 off, N: every N-th align
    unsigned long long align_count = 0;
    std::vector<hipEvent_t> prof_events;  // pairs around each k_match launch of one align

    // device-to-device exchange (lom_comm_attach_p2p): this rank's buffer and the peers' IPC mappings
    bool p2p = false;
    void *p2p_local = nullptr;          // [4 sets][kP2pMaxRanks][32] exchange words in this GPU's HBM
    void *p2p_peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long p2p_epoch = 0;   // device-to-device aligns since the attach: the same number on every rank
    // RCCL
    void *comm = nullptr;       // RCCL communicator (device-side all-gather)
    void *host_comm = nullptr;  // host shared-memory exchange between the ranks of one node
    int rank = 0, nranks = 1;
    lom::DeviceBuf gather;

    // run-time switches: environment read ONCE at lom_map_create, afterwards lom_map_set_option only
    uint32_t table_slots_per_voxel = 16;  // after a bulk insert (LOM_TABLE_SLOTS_PER_VOXEL at create)
    bool opt_host_lm = false;       // LOM_OPT_HOST_LM / LOM_HOST_LM=1
    bool opt_debug_lm = false;      // LOM_OPT_DEBUG_LM_STAMPS / LOM_DEBUG_LM=1
    bool opt_debug_lm_twice = false;  // LOM_DEBUG_LM_TWICE=1 at create: k_lm<256, 64, 2> runs every policy step twice, stamps time the second
    bool opt_debug_timing = false;  // LOM_OPT_DEBUG_TIMING / LOM_DEBUG_TIMING=1
    bool opt_count = false;         // LOM_OPT_COUNT_CANDIDATES / LOM_COUNT_CANDIDATES=1: the searches also produce the reference-
                                    // algorithm counts (occupied voxels, stored points of all 27 neighbours): 27 slot loads per query
    bool opt_no_temporal = false;   // LOM_OPT_NO_TEMPORAL_BOUND / LOM_NO_TEMPORAL=1: every search at the plain max_dist bound
    unsigned long long patience_ticks = 5000000ull;  // bounded in-kernel waits: 50 ms of s_memrealtime (100 MHz)
    bool opt_no_bulk = false;       // LOM_OPT_NO_BULK_INSERT / LOM_NO_BULK_INSERT=1: batches above 65,536 points take the four-kernel path
    uint32_t bulk_ppt = 0;          // LOM_BULK_PPT at create: points per thread of k_bi_claim / k_bi_scatter (0: by batch size)
    uint32_t test_bulk_part_max = 0;  // LOM_OPT_TEST_BULK_PARTITION_MAX: points a partition of the bulk insert may hold (0: the LDS limit)
    int test_grid_give_up = -1;     // LOM_OPT_TEST_GRID_GIVE_UP: first workgroup that gives up in the next in-kernel scan
    int test_give_up_outer = -1;    // LOM_OPT_TEST_GIVE_UP_AT_OUTER: k_lm of that outer iteration of the next align gives up

    std::string last_error;
};

namespace lom {

int set_error(lom_map *m, int code, const char *what, hipError_t e = hipSuccess);
int ensure(lom_map *m, DeviceBuf &b, size_t bytes);  // grow-only device buffer
MapView view_of(const lom_map *m);
int resolve_pending(lom_map *m);  // voxel_map.hip: redo the last single-pass insert if its in-kernel scan gave up
void cleanup_scan_behind_align(lom_map *m);  // voxel_map.hip: see lom_map_radius_cleanup_after_align

#define LOM_HIP(m, expr)                                                        \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return lom::set_error((m), LOM_ERR_HIP, #expr, _e); \
    } while (0)

// RCCL (comm.cpp), loaded lazily with dlopen
int comm_allgather_sums(lom_map *m, const double *d_send, double *d_recv, int count);
// host shared-memory exchange: out = sum over ranks (rank order) of `mine`
int host_exchange_sums(lom_map *m, const double *mine, double *out);
int host_comm_rank(void *host_comm, int *rank, int *nranks, unsigned long long *seq = nullptr);
// lom_host_comm_allreduce with a deadline of the caller's choosing (the exchange that follows a device-to-device
// align must outlast the device-side patience of the slowest rank)
int host_comm_allreduce_deadline(void *host_comm, double *buf, int count, double timeout_s);
const char *host_comm_error(void *host_comm);
// device-to-device exchange (match.hip)
constexpr int kP2pMaxRanks = 8;
void p2p_detach(lom_map *m);

}  // namespace lom
