// Micro-benchmark: A-fragment-shaped global loads (16 B per lane at row*ROWB + s*32 + h*16) vs coalesced loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// ROWB bytes per pixel row (K*2).  A tile = 32 pixels x ROWB bytes, contiguous.  Output: 32 B per pixel, coalesced-ish.
template <int PAT, int ROWB, int MT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ in, f4* __restrict__ out, long ntiles) {
    constexpr int NL = ROWB / 32;                      // loads per lane per tile
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * 256) >> 6;
    const int p = lane & 31, h = lane >> 5;
    for (long t0 = wave * MT; t0 < ntiles; t0 += nw * MT) {
        f4 v[MT][NL];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            long t = t0 + m; if (t >= ntiles) t = ntiles - 1;
            const char* base = in + t * 32 * ROWB;
#pragma unroll
            for (int s = 0; s < NL; ++s)
                v[m][s] = PAT == 0 ? *reinterpret_cast<const f4*>(base + p * ROWB + s * 32 + h * 16)
                                   : *reinterpret_cast<const f4*>(base + s * 1024 + lane * 16);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            long t = t0 + m; if (t >= ntiles) break;
            f4 x = v[m][0];
#pragma unroll
            for (int s = 1; s < NL; ++s) x += v[m][s];
            out[t * 64 + lane] = x;                    // 1 KB per tile
        }
    }
}

template <int PAT, int ROWB, int MT>
void run(const char* in, f4* out, long ntiles, int blocks, const char* name) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, ROWB, MT>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles);
    CK(hipEventRecord(a));
    const int R = 10;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<PAT, ROWB, MT>), dim3(blocks), dim3(256), 0, 0, in, out, ntiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    double bytes = (double)ntiles * (32.0 * ROWB + 1024);
    printf("%-34s ROWB %4d MT %d blocks %6d  %.3f ms  %.2f TB/s\n", name, ROWB, MT, blocks, ms, bytes / ms / 1e9);
}

int main() {
    const long npix = 1024L * 56 * 56;
    const long ntiles = npix / 32;
    char* in; f4* out;
    CK(hipMalloc(&in, npix * 512)); CK(hipMalloc(&out, ntiles * 1024));
    CK(hipMemset(in, 0, npix * 512));
    for (int blocks : {4096, 16384}) {
        run<0, 128, 1>(in, out, ntiles, blocks, "frag-pattern");
        run<1, 128, 1>(in, out, ntiles, blocks, "coalesced");
        run<0, 128, 2>(in, out, ntiles, blocks, "frag-pattern");
        run<1, 128, 2>(in, out, ntiles, blocks, "coalesced");
        run<0, 256, 1>(in, out, ntiles, blocks, "frag-pattern");
        run<1, 256, 1>(in, out, ntiles, blocks, "coalesced");
        run<0, 512, 1>(in, out, ntiles, blocks, "frag-pattern");
        run<1, 512, 1>(in, out, ntiles, blocks, "coalesced");
    }
    return 0;
}
