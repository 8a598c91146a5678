// What does a single wave pay per f64 instruction on gfx950, and does the SIMD skip 16-lane passes whose EXEC bits
// are all zero?  (k_lm's policy step is ~1500 instructions on ONE wave of which 6 lanes do useful work.)
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/exec_skip.hip -o tools/microbench/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// kind 0: one dependent chain of 512 v_fma_f64; kind 1: four independent chains of 128 each (512 FMAs);
// kind 2: 512 dependent v_fma_f32; kind 3: 256 x (v_readlane pair + dependent fma from SGPRs)
template <int kKind>
__global__ __launch_bounds__(64) void k(double *io, unsigned long long *cycles, unsigned long long exec_mask)
{
    double a = io[threadIdx.x], b = io[64 + threadIdx.x];
    double c0 = io[128 + threadIdx.x], c1 = c0 + 1.0, c2 = c0 + 2.0, c3 = c0 + 3.0;
    float fa = (float)a, fb = (float)b, fc = (float)c0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long full = __builtin_amdgcn_read_exec();
    asm volatile("s_mov_b64 exec, %0" ::"s"(exec_mask));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (kKind == 0) {
        REP64(REP8(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));))
    } else if constexpr (kKind == 1) {
        REP64(asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
                           "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                           : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                           : "v"(a), "v"(b));)
    } else if constexpr (kKind == 2) {
        REP64(REP8(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fc) : "v"(fa), "v"(fb));))
    } else {
        REP64(asm volatile("v_readlane_b32 s20, %1, 3\n\tv_readlane_b32 s21, %2, 3\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %0\n\t"
                           "v_readlane_b32 s20, %1, 2\n\tv_readlane_b32 s21, %2, 2\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %0\n\t"
                           "v_readlane_b32 s20, %1, 1\n\tv_readlane_b32 s21, %2, 1\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %0\n\t"
                           "v_readlane_b32 s20, %1, 0\n\tv_readlane_b32 s21, %2, 0\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %0"
                           : "+v"(c0)
                           : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b)
                           : "s20", "s21");)
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, %0" ::"s"(full));
    io[threadIdx.x] = c0 + c1 + c2 + c3 + (double)fc;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int kKind>
static void run(const char *what, int n_ops, double *d, unsigned long long *dc)
{
    const unsigned long long masks[4] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x3Full};
    const char *names[4] = {"64 lanes", "32 lanes", "16 lanes", " 6 lanes"};
    for (int i = 0; i < 4; i++) {
        unsigned long long c = 0;
        for (int rep = 0; rep < 3; rep++) {  // the last of three launches is reported (warm instruction cache)
            hipLaunchKernelGGL(k<kKind>, dim3(1), dim3(64), 0, 0, d, dc, masks[i]);
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-44s exec = %s: %6llu cycles, %.2f per instruction\n", what, names[i], c, (double)c / n_ops);
    }
}

int main()
{
    double *d, h[192];
    unsigned long long *dc;
    for (int i = 0; i < 192; i++) h[i] = 1.0 + 1e-9 * i;
    hipMalloc(&d, sizeof h);
    hipMalloc(&dc, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("512 dependent v_fma_f64", 512, d, dc);
    run<1>("512 v_fma_f64 in four independent chains", 512, d, dc);
    run<2>("512 dependent v_fma_f32", 512, d, dc);
    run<3>("256 x {2 v_readlane, v_fma_f64 from SGPRs}", 256, d, dc);
    return 0;
}
