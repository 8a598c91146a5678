// What does one hand-off between two workgroups cost, by cache policy of the store / the polling load and by placement
// (same XCD or not)?  Two workgroups of a 512-workgroup grid play ping-pong with self-validating 16-byte words
// {bits, seq ^ bits} (a stale or torn read shows a wrong tag and is polled again; every poll loop is bounded).
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/xcd_pingpong.hip -o tools/microbench/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
struct Out { u64 cycles; unsigned xcc_a, xcc_b, timeouts, polls; };
template <int kStore> __device__ __forceinline__ void put(u64x2 *p, u64 v, u64 seq)
{
    u64x2 w; w.x = v; w.y = seq ^ v;
    if constexpr (kStore == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    if constexpr (kStore == 1) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
    if constexpr (kStore == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
    if constexpr (kStore == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(w) : "memory");
}
template <int kLoad> __device__ __forceinline__ u64x2 get(const u64x2 *p)
{
    u64x2 r;
    if constexpr (kLoad == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    if constexpr (kLoad == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    if constexpr (kLoad == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    if constexpr (kLoad == 3) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    if constexpr (kLoad == 4) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    if constexpr (kLoad == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return r;
}
template <int kStore, int kLoad>
__global__ __launch_bounds__(64) void k_pp(u64x2 *slots, int partner, int rounds, u64 seq0, Out *out)
{
    const int b = blockIdx.x;
    if (b != 0 && b != partner) return;
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    u64x2 *mine = slots + (b == 0 ? 0 : 8), *theirs = slots + (b == 0 ? 8 : 0);  // 128 bytes apart
    unsigned timeouts = 0, polls = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int k = 1; k <= rounds; k++) {
        const u64 seq = seq0 + (u64)k;
        if (b == 0) put<kStore>(mine, (u64)k, seq);
        int tries = 0;
        for (;;) {
            const u64x2 r = get<kLoad>(theirs);
            polls++;
            if ((r.x ^ r.y) == seq) break;
            if (++tries > 20000) { timeouts++; break; }
        }
        if (b != 0) put<kStore>(mine, (u64)k, seq);
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (b == 0) { out->cycles = t1 - t0; out->xcc_a = xcc; out->timeouts = timeouts; out->polls = polls; }
    else { out->xcc_b = xcc; out->timeouts += 1000 * timeouts; }
}
static u64 g_seq = 1000;
template <int kStore, int kLoad>
static void run(u64x2 *slots, Out *d_out, int partner, const char *name)
{
    const int rounds = 300;
    Out h{};
    hipMemset(d_out, 0, sizeof(Out));
    hipLaunchKernelGGL((k_pp<kStore, kLoad>), dim3(512), dim3(64), 0, 0, slots, partner, rounds, g_seq, d_out);
    hipDeviceSynchronize();
    g_seq += 100000;
    hipMemcpy(&h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s partner block %d (XCC %u / %u): %7.0f cycles per round trip (two hand-offs), %5.1f polls per round, timeouts %u\n", name, partner,
           h.xcc_a, h.xcc_b, (double)h.cycles / rounds, (double)h.polls / rounds, h.timeouts);
}
int main()
{
    u64x2 *slots; Out *d_out;
    hipMalloc(&slots, 4096); hipMemset(slots, 0, 4096);
    hipMalloc(&d_out, sizeof(Out));
    for (int partner : {8, 1, 4}) {
        run<0, 0>(slots, d_out, partner, "store sc1, load sc1 (k_lm today)");
        run<1, 0>(slots, d_out, partner, "store plain, load sc1");
        run<0, 1>(slots, d_out, partner, "store sc1, load sc0");
        run<1, 1>(slots, d_out, partner, "store plain, load sc0");
        run<3, 1>(slots, d_out, partner, "store sc0, load sc0");
        run<1, 3>(slots, d_out, partner, "store plain, buffer_inv sc0 + plain load");
        run<1, 5>(slots, d_out, partner, "store plain, buffer_inv sc1 + plain load");
        run<1, 4>(slots, d_out, partner, "store plain, load nt");
        run<2, 2>(slots, d_out, partner, "store sc0 sc1, load sc0 sc1");
    }
    return 0;
}
