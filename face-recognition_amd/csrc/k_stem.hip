// 3x3 stem convolution with Cin = 3 (stride 1 for the MobileNets, stride 2 for SphereFace layer1):
// fp32 NCHW network input -> NHWC activations, as an MFMA implicit GEMM with K = 27 (padded to 32),
// fused folded-BN bias + PReLU / Hardswish.  Replaces the first Conv2dNormActivation of every
// backbone (models/mobilenetv1.py:18-24, mobilenetv2.py:112-118, mobilenetv3.py:133-141,
// sphereface.py:107-115 for layer1).
//
// A block owns an 8x16 output tile: the 3 x (8S+2) x (16S+2) fp32 input patch is staged in LDS
// (zero padded), each wave builds the im2col "B" fragments of a 4x8-pixel sub-tile from LDS
// (k = c*9 + ky*3 + kx, the natural order of weight.reshape(cout, 27)) and multiplies by the
// register-resident weight fragments.  Output layout/epilogue are shared with the pointwise GEMM.
#include "kernels.h"

template <typename TOut, bool SPLIT>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
    constexpr int NS = SPLIT ? 2 : 1;
    constexpr int TH = 8, TW = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    const int S = a.stride;
    const int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int n = b;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = oy0 * S - 1, gx0 = ox0 * S - 1;

    for (int i = threadIdx.x; i < 3 * IH * IW; i += 256) {
        const int c = i / (IH * IW); const int rem = i - c * IH * IW;
        const int iy = rem / IW, ix = rem - iy * IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        float v = 0.f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.x[(((long)n * 3 + c) * a.H + gy) * a.W + gx];
        patch[i] = v;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int py = 4 * (wave >> 1) + (p >> 3), px = 8 * (wave & 1) + (p & 7);
    const int base = py * S * IW + px * S;

    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Vec8f v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            const int c = k / 9, r9 = k - 9 * c;
            const int ky = r9 / 3, kx = r9 - 3 * ky;
            v.v[j] = k < 27 ? patch[c * IH * IW + ky * IW + kx + base] : 0.f;
        }
        if constexpr (SPLIT) split8(v, xh[s], xl[s]);
        else xh[s] = to_frag(v);
    }

    const int gy = oy0 + py, gx = ox0 + px;
    const bool ok = gy < a.Ho && gx < a.Wo;
    const long m = ((long)n * a.Ho + gy) * a.Wo + gx;
    const bf16x8* W = reinterpret_cast<const bf16x8*>(a.w);
    for (int t = 0; t < a.NT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8* wf = W + ((t * 2 + s) * NS) * 64 + lane;
            bf16x8 wh = wf[0];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
            if constexpr (SPLIT) {
                bf16x8 wl = wf[64];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
            }
        }
        epilogue_store<TOut>(acc, a.e, m, 32 * t + 16 * h, ok);
    }
}

int launch_stem(bool out_f32, const StemArgs& a, hipStream_t st) {
    if (a.e.cout % 8 || a.NT != cdiv(a.e.cout, 32) || (a.stride != 1 && a.stride != 2))
        FR_FAIL(FR_ERR_ARG, "stem: inconsistent shapes");
    const int IH = 7 * a.stride + 3, IW = 15 * a.stride + 3;
    const size_t lds = (size_t)3 * IH * IW * sizeof(float);
    const long blocks = (long)a.N * cdiv(a.Ho, 8) * cdiv(a.Wo, 16);
    if (blocks > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "stem: grid too large");
    if (out_f32) hipLaunchKernelGGL((stem_kernel<float, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((stem_kernel<bf16, false>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    FR_LAUNCH_CHECK("stem_kernel");
    return FR_OK;
}
