// v_permlane32_swap / v_permlane16_swap on gfx950: which lanes end up where (build: hipcc -O2 --offload-arch=gfx950
// tools/microbench/permlane_swap.hip -o tools/microbench/permlane_swap).  k_lm's wave-level reduce-scatter relies on:
//   permlane32_swap(A, B): A' = [A.lanes0-31, B.lanes0-31], B' = [A.lanes32-63, B.lanes32-63]
//   permlane16_swap(A, B): A' = [A.row0, B.row0, A.row2, B.row2], B' = [A.row1, B.row1, A.row3, B.row3]
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o)
{
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    o[128 + threadIdx.x] = s[0];
    o[192 + threadIdx.x] = s[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[4] = {"swap32 A'", "swap32 B'", "swap16 A'", "swap16 B'"};
    int bad = 0;
    for (int v = 0; v < 4; v++) {
        printf("%s:", names[v]);
        for (int l = 0; l < 64; l += 8) printf(" [%u..]", h[v * 64 + l]);
        printf("\n");
        for (int l = 0; l < 64; l++) {
            unsigned want;
            if (v == 0) want = l < 32 ? (unsigned)l : 100u + (l - 32);
            else if (v == 1) want = l < 32 ? (unsigned)(l + 32) : 100u + l;
            else {
                const int row = l / 16, j = l % 16;
                const int src_row = (v == 2) ? (row & 2) : (row & 2) + 1;   // A' takes even rows, B' odd rows
                const bool fromB = row & 1;
                want = (fromB ? 100u : 0u) + src_row * 16 + j;
            }
            if (h[v * 64 + l] != want) bad++;
        }
    }
    printf("mismatches against the documented lane mapping: %d\n", bad);
    return bad != 0;
}
