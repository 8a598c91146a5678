// What does the vector L1 (TCP) count and charge per load instruction?  One wave per workgroup loads 16 bytes
// per lane from a 64 KiB cache-resident region with a given lane stride; run under
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
// and read accesses / wavefront per stride; the program itself prints cycles per load instruction.
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/tcp_lines.hip -o tools/microbench/tcp_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int kStride, int kBytes>
__global__ __launch_bounds__(64) void k_probe(const char *base, int iters, unsigned long long *out, uint32_t *sink)
{
    const char *p = base + (size_t)threadIdx.x * kStride;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        const char *a = p + ((i * 4096) & 0xFFFF);
        if constexpr (kBytes == 16) {
            u32x4 r;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
            acc += r.x;
        } else {
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            u32x3 r;
            asm volatile("global_load_dwordx3 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
            acc += r.x;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int kStride, int kBytes>
static void run(const char *d, unsigned long long *d_out, uint32_t *d_sink, const char *label)
{
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k_probe<kStride, kBytes>), dim3(blocks), dim3(64), 0, 0, d, iters, d_out, d_sink);
    hipLaunchKernelGGL((k_probe<kStride, kBytes>), dim3(blocks), dim3(64), 0, 0, d, iters, d_out, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-40s stride %4d B  %2d B/lane: %7.1f memtime ticks per dependent load (one wave per CU)\n", label, kStride, kBytes,
           s / blocks / iters);
}
int main()
{
    char *d;
    unsigned long long *d_out;
    uint32_t *d_sink;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 1, 1 << 20);
    hipMalloc(&d_out, 256 * 8);
    hipMalloc(&d_sink, 64);
    run<16, 16>(d, d_out, d_sink, "k_probe<16,16>  8 x 128 B");
    run<32, 16>(d, d_out, d_sink, "k_probe<32,16>  16 x 128 B, 32 x 64 B");
    run<64, 16>(d, d_out, d_sink, "k_probe<64,16>  32 x 128 B, 64 x 64 B");
    run<128, 16>(d, d_out, d_sink, "k_probe<128,16> 64 x 128 B");
    run<12, 12>(d, d_out, d_sink, "k_probe<12,12>  6 x 128 B");
    run<48, 12>(d, d_out, d_sink, "k_probe<48,12>  24 x 128 B");
    run<0, 16>(d, d_out, d_sink, "k_probe<0,16>   one address");
    return 0;
}
