// How much does one dependent kernel boundary cost on this GPU?  Chains of trivial kernels on one
// stream, timed with a HIP event pair around the whole chain (tools/microbench, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_spin(int *p, long long ticks)  // ~fixed-duration kernel: 100 MHz realtime ticks
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

int main()
{
    int *d = nullptr;
    hipMalloc(&d, 4);
    hipMemset(d, 0, 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Case { const char *name; int grid, block; long long ticks; };
    const Case cases[] = {{"empty 1x64", 1, 64, 0}, {"empty 1666x256", 1666, 256, 0}, {"empty 52x512", 52, 512, 0},
                          {"spin 10us 52x512", 52, 512, 1000}, {"spin 10us 1666x256", 1666, 256, 1000}};
    for (const Case &c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            const int n = 2000;
            hipEventRecord(e0, s);
            for (int i = 0; i < n; i++) {
                if (c.ticks)
                    hipLaunchKernelGGL(k_spin, dim3(c.grid), dim3(c.block), 0, s, d, c.ticks);
                else
                    hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, d);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-22s %.2f us per kernel (chain of %d)\n", c.name, ms * 1e3 / n, n);
        }
    }
    return 0;
}
