// Host cost of issuing a chain of six small dependent kernels: six hipLaunchKernelGGL against one hipGraphLaunch of a
// six-node graph whose kernel parameters are set anew before every launch (hipGraphExecKernelNodeSetParams), as a per-frame
// chain with changing arguments would need.  Build: hipcc -O2 --offload-arch=gfx950 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

struct Args {
    float r[12];
};
__global__ void k_small(float *p, unsigned n, Args a, unsigned seq, const unsigned *words, float *q, size_t stride)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * a.r[i % 12] + (float)seq + (words ? (float)words[0] : 0.f) + (q ? q[i] : 0.f) + (float)stride;
}

// the same 22 values as 22 kernel arguments and as one struct argument: what a launch costs the host per argument
struct Many {
    float *a0;
    unsigned a1;
    float *a2;
    const unsigned *a3;
    unsigned a4, a5, a6, a7;
    const char *a8, *a9;
    size_t a10;
    unsigned a11, a12;
    float *a13, *a14;
    unsigned *a15, *a16;
    unsigned a17;
    const unsigned *a18;
    unsigned *a19, *a20;
    float a21;
};
__global__ void k_many(float *a0, unsigned a1, float *a2, const unsigned *a3, unsigned a4, unsigned a5, unsigned a6, unsigned a7,
                       const char *a8, const char *a9, size_t a10, unsigned a11, unsigned a12, float *a13, float *a14, unsigned *a15,
                       unsigned *a16, unsigned a17, const unsigned *a18, unsigned *a19, unsigned *a20, float a21)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a1) a0[i] += a21 + (float)(a4 + a5 + a6 + a7 + a11 + a12 + a17) + (float)a10 + (a3 ? (float)a3[0] : 0.f) +
                         (float)((size_t)a2 + (size_t)a8 + (size_t)a9 + (size_t)a13 + (size_t)a14 + (size_t)a15 + (size_t)a16 +
                                 (size_t)a18 + (size_t)a19 + (size_t)a20 == 1);
}
__global__ void k_one(Many m)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m.a1) m.a0[i] += m.a21 + (float)(m.a4 + m.a5 + m.a6 + m.a7 + m.a11 + m.a12 + m.a17) + (float)m.a10 +
                             (m.a3 ? (float)m.a3[0] : 0.f) +
                             (float)((size_t)m.a2 + (size_t)m.a8 + (size_t)m.a9 + (size_t)m.a13 + (size_t)m.a14 + (size_t)m.a15 +
                                     (size_t)m.a16 + (size_t)m.a18 + (size_t)m.a19 + (size_t)m.a20 == 1);
}

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const unsigned n = 10000;
    float *d = nullptr, *q = nullptr;
    unsigned *w = nullptr;
    CK(hipMalloc(&d, n * 4));
    CK(hipMalloc(&q, n * 4));
    CK(hipMalloc(&w, 64));
    CK(hipMemset(d, 0, n * 4));
    CK(hipMemset(q, 0, n * 4));
    CK(hipMemset(w, 0, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Args a{};
    const int kNodes = 6, kIters = 2000;
    // (a) direct launches
    double t_direct = 0;
    for (int it = 0; it < kIters + 100; it++) {
        const double t0 = now_us();
        for (int k = 0; k < kNodes; k++)
            hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, s, d, n, a, (unsigned)it, w, q, (size_t)12);
        const double t1 = now_us();
        if (it >= 100) t_direct += t1 - t0;
        CK(hipStreamSynchronize(s));
    }
    // (b) graph with per-launch parameter updates
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(kNodes);
    unsigned seq = 0;
    size_t stride = 12;
    unsigned nn = n;
    void *params[] = {&d, &nn, &a, &seq, &w, &q, &stride};
    hipKernelNodeParams kp{};
    kp.func = reinterpret_cast<void *>(k_small);
    kp.gridDim = dim3((n + 255) / 256);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = params;
    kp.extra = nullptr;
    for (int k = 0; k < kNodes; k++)
        CK(hipGraphAddKernelNode(&nodes[k], g, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &kp));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double t_graph = 0, t_set = 0;
    for (int it = 0; it < kIters + 100; it++) {
        seq = (unsigned)it;
        const double t0 = now_us();
        for (int k = 0; k < kNodes; k++) CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp));
        const double t1 = now_us();
        CK(hipGraphLaunch(ge, s));
        const double t2 = now_us();
        if (it >= 100) {
            t_set += t1 - t0;
            t_graph += t2 - t1;
        }
        CK(hipStreamSynchronize(s));
    }
    // (c) graph launch without parameter updates
    double t_plain = 0;
    for (int it = 0; it < kIters + 100; it++) {
        const double t0 = now_us();
        CK(hipGraphLaunch(ge, s));
        const double t1 = now_us();
        if (it >= 100) t_plain += t1 - t0;
        CK(hipStreamSynchronize(s));
    }
    // device time of the chain either way (events around it)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_direct = 0, ms_graph = 0;
    CK(hipEventRecord(e0, s));
    for (int it = 0; it < 200; it++)
        for (int k = 0; k < kNodes; k++)
            hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, s, d, n, a, (unsigned)it, w, q, (size_t)12);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_direct, e0, e1));
    CK(hipEventRecord(e0, s));
    for (int it = 0; it < 200; it++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    {   // 22 arguments against one struct
        Many mm{};
        mm.a0 = d;
        mm.a1 = n;
        mm.a3 = w;
        double t22 = 0, t1s = 0;
        for (int it = 0; it < kIters + 100; it++) {
            const double t0 = now_us();
            for (int k = 0; k < kNodes; k++)
                hipLaunchKernelGGL(k_many, dim3((n + 255) / 256), dim3(256), 0, s, mm.a0, mm.a1, mm.a2, mm.a3, mm.a4, mm.a5, mm.a6, mm.a7,
                                   mm.a8, mm.a9, mm.a10, mm.a11, mm.a12, mm.a13, mm.a14, mm.a15, mm.a16, mm.a17, mm.a18, mm.a19,
                                   mm.a20, mm.a21);
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            for (int k = 0; k < kNodes; k++) hipLaunchKernelGGL(k_one, dim3((n + 255) / 256), dim3(256), 0, s, mm);
            const double t3 = now_us();
            CK(hipStreamSynchronize(s));
            if (it >= 100) {
                t22 += t1 - t0;
                t1s += t3 - t2;
            }
        }
        std::printf("host time per launch: 22 arguments %.2f us, the same values as one struct argument %.2f us\n", t22 / kIters / kNodes,
                    t1s / kIters / kNodes);
    }
    std::printf("six small dependent kernels, host time per chain: direct launches %.2f us (%.2f each); graph: set params %.2f us + launch %.2f us; "
                "graph launch alone %.2f us\n",
                t_direct / kIters, t_direct / kIters / kNodes, t_set / kIters, t_graph / kIters, t_plain / kIters);
    std::printf("device time per chain back to back: direct %.2f us, graph %.2f us\n", ms_direct * 1e3 / 200, ms_graph * 1e3 / 200);
    return 0;
}
