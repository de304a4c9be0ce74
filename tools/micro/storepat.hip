// Micro-benchmark: does the per-lane store pattern of the pointwise epilogue limit HBM write rate?
// Every variant reads R bytes per pixel coalesced and writes W = 128 B per pixel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// pattern 0: lane (p = lane&31, h = lane>>5) writes 16 B at pixel*128 + nt*64 + h*32 + {0,16}  (the MFMA epilogue)
// pattern 1: fully coalesced: the wave's 32 pixels x 128 B = 4 KB written as 4 x (64 lanes x 16 B) contiguous
template <int PAT, int MT>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, f4* __restrict__ out, long ntiles) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * 256) >> 6;
    for (long t0 = wave * MT; t0 < ntiles; t0 += nw * MT) {
        f4 v[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { long t = t0 + m; if (t < ntiles) v[m] = in[t * 64 + lane]; }   // 1 KB per tile (K=16 bf16 x 32 px)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            long t = t0 + m; if (t >= ntiles) break;
            f4 x = v[m];
            char* base = reinterpret_cast<char*>(out) + t * 4096;
            if (PAT == 0) {
                const int p = lane & 31, h = lane >> 5;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        x.x += 1.f;
                        *reinterpret_cast<f4*>(base + p * 128 + nt * 64 + h * 32 + q * 16) = x;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { x.x += 1.f; *reinterpret_cast<f4*>(base + q * 1024 + lane * 16) = x; }
            }
        }
    }
}

template <int PAT, int MT>
void run(const f4* in, f4* out, long ntiles, int blocks, const char* name) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, MT>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles);
    CK(hipEventRecord(a));
    const int R = 10;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<PAT, MT>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    double bytes = (double)ntiles * (1024 + 4096);
    printf("%-28s blocks %6d  %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / ms / 1e9);
}

int main() {
    const long ntiles = 1024L * 112 * 112 / 32;       // features.2 expand: 12.8 M pixels
    f4 *in, *out;
    CK(hipMalloc(&in, ntiles * 1024)); CK(hipMalloc(&out, ntiles * 4096));
    CK(hipMemset(in, 0, ntiles * 1024));
    for (int blocks : {2048, 8192, 32768}) {
        run<0, 1>(in, out, ntiles, blocks, "mfma-pattern MT1");
        run<0, 4>(in, out, ntiles, blocks, "mfma-pattern MT4");
        run<1, 1>(in, out, ntiles, blocks, "coalesced MT1");
        run<1, 4>(in, out, ntiles, blocks, "coalesced MT4");
    }
    return 0;
}
