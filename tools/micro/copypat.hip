// Micro-benchmark: what a pure streaming kernel reaches on MI355X when it READS and WRITES (the 1x1 convs and the depthwise
// convs do both): coalesced 16-byte loads and stores, R bytes read per W bytes written, no arithmetic to speak of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// each thread: RD loads of 16 B (unit-stride across the wave), then WR stores of 16 B
template <int RD, int WR>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, f4* __restrict__ out, long nunits) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x, nt = (long)gridDim.x * 256;
    for (long u = t; u < nunits; u += nt) {
        f4 v[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) v[i] = in[(long)i * nunits + u];
        f4 s = v[0];
#pragma unroll
        for (int i = 1; i < RD; ++i) s += v[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) out[(long)i * nunits + u] = s + (float)i;
    }
}

template <int RD, int WR>
void run(const f4* in, f4* out, long nunits, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<RD, WR>), dim3(blocks), dim3(256), 0, 0, in, out, nunits);
    CK(hipEventRecord(a));
    const int R = 10;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<RD, WR>), dim3(blocks), dim3(256), 0, 0, in, out, nunits);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    const double bytes = (double)nunits * 16.0 * (RD + WR);
    printf("read %d : write %d  blocks %6d  %7.3f ms  %5.2f TB/s  (%.0f MB)\n", RD, WR, blocks, ms, bytes / ms / 1e9, bytes / 1e6);
}

int main() {
    const long nunits = 16L << 20;                 // 256 MB per stream: well beyond the 256 MB of MALL in total
    f4 *in, *out;
    CK(hipMalloc(&in, nunits * 16 * 4)); CK(hipMalloc(&out, nunits * 16 * 4));
    CK(hipMemset(in, 0, nunits * 16 * 4));
    for (int blocks : {2048, 8192, 32768}) {
        run<4, 1>(in, out, nunits, blocks);
        run<2, 1>(in, out, nunits, blocks);
        run<1, 1>(in, out, nunits, blocks);
        run<1, 2>(in, out, nunits, blocks);
        run<1, 4>(in, out, nunits, blocks);
    }
    // launches of the size of one layer of a 512-image sub-batch (20-150 MB): ramp-up and tail are part of the time.
    // (back-to-back launches of the same buffers: what fits the 256 MB of MALL is partly served from it)
    printf("layer-sized launches, 2048 blocks (what the 1x1 kernels use):\n");
    for (long mb : {16L, 32L, 64L, 128L, 256L}) {
        const long nu = mb * (1L << 20) / 16 / 2;          // read 1 : write 1 -> mb MB in total
        run<1, 1>(in, out, nu, 2048);
        run<1, 4>(in, out, nu * 2 / 5, 2048);
    }
    return 0;
}
