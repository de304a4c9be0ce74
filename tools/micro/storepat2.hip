// Micro-benchmark: store rate vs the size of the contiguous piece each group of lanes writes.
// One store instruction = 64 lanes x 16 B; lanes are grouped in runs of PIECE/16 lanes writing PIECE contiguous
// bytes; consecutive runs are GAP pieces apart (the other pieces are written by the following instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PIECE, int GAP>
__global__ __launch_bounds__(256) void k(f4* __restrict__ out, long nchunks /* of 64*GAP*16 bytes */) {
    constexpr int LP = PIECE / 16;                    // lanes per piece
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * 256) >> 6;
    const int run = lane / LP, within = lane % LP;
    f4 x = {1.f, 2.f, 3.f, 4.f};
    for (long c = wave; c < nchunks; c += nw) {
        char* base = reinterpret_cast<char*>(out) + c * (64L * GAP * 16);
#pragma unroll
        for (int g = 0; g < GAP; ++g) {
            x.x += 1.f;
            *reinterpret_cast<f4*>(base + ((long)run * GAP + g) * PIECE + within * 16) = x;
        }
    }
}

template <int PIECE, int GAP>
void run(f4* out, long bytes) {
    const long nchunks = bytes / (64L * GAP * 16);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PIECE, GAP>), dim3(8192), dim3(256), 0, 0, out, nchunks);
    CK(hipEventRecord(a));
    const int R = 10;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<PIECE, GAP>), dim3(8192), dim3(256), 0, 0, out, nchunks);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    printf("piece %4d B  gap %d  %.3f ms  %.2f TB/s (write only)\n", PIECE, GAP, ms, (double)nchunks * 64 * GAP * 16 / ms / 1e9);
}

int main() {
    const long bytes = 2L << 30;
    f4* out; CK(hipMalloc(&out, bytes));
    run<16, 4>(out, bytes); run<32, 4>(out, bytes); run<64, 4>(out, bytes); run<128, 4>(out, bytes);
    run<256, 4>(out, bytes); run<1024, 4>(out, bytes); run<1024, 1>(out, bytes);
    run<16, 8>(out, bytes); run<32, 8>(out, bytes); run<64, 2>(out, bytes); run<128, 2>(out, bytes);
    return 0;
}
