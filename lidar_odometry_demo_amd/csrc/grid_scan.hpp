// Workgroup scan + in-kernel prefix over the workgroups of a grid, shared by the map-maintenance kernels
// (voxel_map.hip) and the per-frame front end (frontend.hip).
//
// Single-pass kernels for per-frame sizes: at most 256 workgroups of 256 threads, all resident at once.
// What would be {flag kernel, 1-3 scan launches, consumer kernel} is one kernel: block-local scan, then
// every workgroup publishes its total as a tagged 8-byte word {call sequence number, value} (one store;
// no reset between calls, the sequence number tells fresh from stale) and adds up the totals of the
// workgroups before it -- <= 255 words, one per thread, fixed order, so the prefix is deterministic.
// Every wait is bounded (s_memrealtime).  A workgroup that gives up writes the call's sequence number into
// the error word and tells its caller (gave_up): it has NO valid prefix and must not write results that depend
// on one -- only put per-slot scratch back to rest.  Kernels launched behind it in the same call see the error
// word and do likewise, so a call that gave up changes nothing the next call could trip over; the host then
// redoes it with the multi-launch scan (voxel_map.hip), or reports it where the inputs are gone.
// (A workgroup waits for its PREDECESSORS only, and the dispatcher starts workgroups in index order, so this
// cannot deadlock; it gives up only when something else keeps the GPU from running the grid for 20 ms.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lom {

constexpr int kThreads = 256;
static inline uint32_t blocks_for(size_t n, int t = kThreads) { return (uint32_t)((n + t - 1) / t); }

constexpr uint32_t kOnePassMax = 256u * kThreads;
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
constexpr unsigned long long kGridWaitTicks = 2000000ull;  // 20 ms of s_memrealtime (100 MHz)
constexpr uint32_t kGridFailOnlyOne = 0x40000000u;

struct __attribute__((aligned(8))) Granule {
    uint32_t seq, val;
};

__device__ __forceinline__ void granule_store(Granule *g, uint32_t seq, uint32_t val)
{
    const unsigned long long w = ((unsigned long long)val << 32) | seq;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(g), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t granule_wait(const Granule *g, uint32_t seq, uint32_t *err_word, bool &timed_out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(g), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)w == seq) return (uint32_t)(w >> 32);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kGridWaitTicks) {
            __hip_atomic_store(err_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = true;
            return 0u;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// exclusive scan of one packed u64 per thread over the workgroup (two u32 quantities, no carry while
// the low sums stay below 2^32); s_w: 8 words of LDS; total = the workgroup's sum
__device__ __forceinline__ unsigned long long block_scan64(unsigned long long v, unsigned long long *s_w,
                                                           unsigned long long &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned long long off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
        if (w < wave) off += s_w[w];
        total += s_w[w];
    }
    __syncthreads();
    return off + inc - v;
}

// sum of the (lo, hi) totals of all workgroups before this one; agg: [gridDim.x][2] granules.
// gave_up (uniform over the workgroup): a wait timed out, the returned prefix is not valid.
// test_fail_from: workgroups of that index and beyond behave as if their waits had timed out
// (LOM_OPT_TEST_GRID_GIVE_UP; 0xFFFFFFFF in production); with kGridFailOnlyOne set, that workgroup alone -- a give-up
// in the middle of a grid whose later workgroups still get their prefix.
__device__ __forceinline__ unsigned long long grid_prefix64(unsigned long long my_total, Granule *agg, uint32_t seq,
                                                            uint32_t *err_word, unsigned long long *s_w, bool &gave_up,
                                                            uint32_t test_fail_from = 0xFFFFFFFFu)
{
    if (threadIdx.x == 0) {
        granule_store(agg + 2 * blockIdx.x, seq, (uint32_t)my_total);
        granule_store(agg + 2 * blockIdx.x + 1, seq, (uint32_t)(my_total >> 32));
    }
    unsigned long long v = 0;
    bool timed_out = false;
    if (threadIdx.x < blockIdx.x) {
        const bool forced = (test_fail_from != 0xFFFFFFFFu && (test_fail_from & kGridFailOnlyOne))
                                ? blockIdx.x == (test_fail_from & ~kGridFailOnlyOne)
                                : blockIdx.x >= test_fail_from;
        if (forced) {
            __hip_atomic_store(err_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = true;
        } else {
            const uint32_t lo = granule_wait(agg + 2 * threadIdx.x, seq, err_word, timed_out);
            const uint32_t hi = granule_wait(agg + 2 * threadIdx.x + 1, seq, err_word, timed_out);
            v = ((unsigned long long)hi << 32) | lo;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[4 + wave] = v;
    gave_up = __syncthreads_or(timed_out ? 1 : 0) != 0;
    unsigned long long sum = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) sum += s_w[4 + w];
    __syncthreads();
    return sum;
}


}  // namespace lom
