// Does global_load_dwordx4 return the right 16 bytes from an address that is only 4-byte aligned?  (gfx950, ROCm 7.2)
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/unaligned_x4.hip -o tools/microbench/unaligned_x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t *base, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *a = base + i * 3 + 1;  // 12-byte stride, 4 bytes off
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
    out[i * 4 + 0] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}
int main()
{
    const int n = 1 << 16;
    std::vector<uint32_t> h(n * 3 + 8), o(n * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t *d, *d_o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&d_o, o.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, d_o, n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) for (int k2 = 0; k2 < 4; k2++) bad += o[i * 4 + k2] != h[i * 3 + 1 + k2];
    printf("sync: %s; mismatching words: %d of %d\n", hipGetErrorString(e), bad, n * 4);
    return bad != 0;
}
