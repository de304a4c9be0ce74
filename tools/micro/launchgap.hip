// Micro-benchmark: what 59 DEPENDENT kernels cost on one stream when each does almost nothing -- the floor under the
// batch-1 forward (59 launches).  One wave per kernel reads and writes 256 B; back-to-back launches, and the same chain
// replayed as a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void tiny(const float* __restrict__ in, float* __restrict__ out) { out[threadIdx.x] = in[threadIdx.x] + 1.f; }
// a "layer-shaped" small kernel: blocks x 256 threads, each block stages 16 KB through LDS-free loads and stores
__global__ __launch_bounds__(256) void small(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { float4 v = in[i]; v.x += 1.f; out[i] = v; }
}

int main() {
    const int CH = 59, REP = 200;
    float *a, *b;
    CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22));
    CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int mode = 0; mode < 2; ++mode) {
        const int n4 = mode == 0 ? 0 : (150528 / 4);           // mode 1: 150 KB per kernel (one 112 x 112 x 3 fp32 image)
        auto chain = [&] {
            for (int i = 0; i < CH; ++i) {
                float* src = (i & 1) ? b : a; float* dst = (i & 1) ? a : b;
                if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, src, dst);
                else hipLaunchKernelGGL(small, dim3(37), dim3(256), 0, st, (const float4*)src, (float4*)dst, n4);
            }
        };
        for (int i = 0; i < 20; ++i) chain();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < REP; ++i) chain();
        CK(hipStreamSynchronize(st));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
        printf("%s: %d dependent launches  %.1f us per chain  %.2f us per kernel (eager)\n", mode ? "150 KB copy kernels" : "one-wave kernels", CH, us, us / CH);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        chain();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < REP; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
        printf("%s: %d dependent launches  %.1f us per chain  %.2f us per kernel (hipGraph replay)\n", mode ? "150 KB copy kernels" : "one-wave kernels", CH, us, us / CH);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
