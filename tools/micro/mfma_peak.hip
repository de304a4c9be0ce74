// Sustained v_mfma_f32_32x32x16_bf16 rate with nothing else in the loop: the practical ceiling the MFMA-bound kernels
// (search_rq_kernel, conv_halo_kernel) should be priced against besides the 2.5 PFLOP/s datasheet figure.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && tools/micro/mfma_peak [waves_per_simd] [ms] [chains]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int CH>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / CH; ++u)
#pragma unroll
            for (int t = 0; t < CH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);   // CH independent chains
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;            // waves per SIMD
    const double want_ms = argc > 2 ? atof(argv[2]) : 200.0;
    const int chains = argc > 3 ? atoi(argv[3]) : 4;          // independent accumulator chains per wave: 1, 2 or 4
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (chains == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (chains == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
        printf("CUs %d, %d wave(s)/SIMD, %d chain(s), %d iters: %.2f ms  %.1f TFLOP/s\n", cus, wps, chains, iters, ms, flops / ms / 1e9);
        iters = (int)(iters * want_ms / ms);
        if (iters < 1000) iters = 1000;
        if (iters > 50000000) iters = 50000000;
    }
    hipFree(out);
    return 0;
}
