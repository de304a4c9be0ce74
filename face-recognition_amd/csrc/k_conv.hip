// Dense KxK convolution (K = 3, stride 1|2, pad 1: SphereFace) and large-K Linear (K = 1 on a 1x1
// map: SphereFace fc 25088 -> 512) as a bf16 MFMA implicit GEMM with fused bias + PReLU + residual.
// Replaces Conv2dNormActivation(norm_layer=None) + PReLU and `x + block(x)` (models/sphereface.py:
// 18-48, 107-117) and nn.Linear(512*7*7, 512) (models/sphereface.py:84, 126-127).
//
// D[cout][pixel] = sum_{tap,c} W[cout][tap][c] * X[pixel shifted by tap][c]  (compute-bound, MFMA).
// The K loop walks (tap, 32-channel chunk) slices.  Per slice a block stages, through registers
// (issue the global loads for slice s+1, run the MFMAs of slice s, then write LDS: T14 in the CDNA
// guide), a [BN couts][32] weight tile (host-packed in exactly this order) and a [BM pixels][32]
// activation tile gathered from NHWC with zero fill at the borders.  LDS rows are 64 B + 16 B pad
// so every ds_read_b128 fragment read is bank-conflict free.  LDS use is 20-50 KB, so 3-4 blocks
// share a CU and hide each other's barriers.  Wave tile = TI x TJ MFMA 32x32x16 tiles; the weight
// rows are permuted (common.h) so each lane holds 16 consecutive output channels of a pixel.
// fp32 plans (TIn = float) split both operands into bf16 hi+lo planes at staging: 3 MFMAs per product.
#include "kernels.h"

#define CV_ROW 80     // bytes per LDS row: 32 bf16 + 16 B pad

template <typename TIn, typename TOut, int TI, int TJ, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    constexpr bool SPLIT = std::is_same<TIn, float>::value;
    constexpr int NS = SPLIT ? 2 : 1;
    constexpr int BN = 32 * TI * WM, BM = 32 * TJ * WN;
    static_assert(WM * WN == 4 && BM == 128, "4 waves, 128 pixels per block");
    constexpr int A_BYTES = BN * CV_ROW, B_BYTES = BM * CV_ROW;
    constexpr int BUF = NS * (A_BYTES + B_BYTES);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const long m0 = (long)blockIdx.x * BM;
    const int nt0 = blockIdx.y;                          // cout tile of BN
    const int chunks = a.Cin >> 5;
    const int KS = a.ksize * a.ksize * chunks;

    // staging roles: unit = (row, 32-byte half of the 64-byte row)
    const int srow = tid >> 1, shalf = tid & 1;
    // B operand: this thread's pixel
    const long spix = m0 + srow;
    const bool spix_ok = spix < a.M;
    int sn = 0, soy = 0, sox = 0;
    if (spix_ok) { const int hw = a.Ho * a.Wo; sn = (int)(spix / hw); const int r = (int)(spix - (long)sn * hw); soy = r / a.Wo; sox = r - soy * a.Wo; }
    const TIn* X = reinterpret_cast<const TIn*>(a.x);
    const bf16* Wp = a.w + (size_t)nt0 * KS * NS * BN * 32;       // [ks][NS][BN][32]

    u32x4 pa[NS][2], pb[SPLIT ? 4 : 2];
    auto prefetch = [&](int ks) {
        if (srow < BN) {
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                const bf16* src = Wp + ((size_t)(ks * NS + sp) * BN + srow) * 32 + shalf * 16;
                pa[sp][0] = *reinterpret_cast<const u32x4*>(src);
                pa[sp][1] = *reinterpret_cast<const u32x4*>(src + 8);
            }
        }
        const int tap = ks / chunks, c0 = (ks - tap * chunks) << 5;
        const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
        const int iy = soy * a.stride + ky - a.pad, ix = sox * a.stride + kx - a.pad;
        const bool ok = spix_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        if (ok) {
            const TIn* src = X + (((long)sn * a.H + iy) * a.W + ix) * a.Cin + c0 + shalf * 16;
#pragma unroll
            for (int i = 0; i < (SPLIT ? 4 : 2); ++i) pb[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * i);
        } else {
#pragma unroll
            for (int i = 0; i < (SPLIT ? 4 : 2); ++i) pb[i] = u32x4{0, 0, 0, 0};
        }
    };
    auto commit = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
        if (srow < BN) {
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                unsigned char* dst = base + sp * A_BYTES + srow * CV_ROW + shalf * 32;
                *reinterpret_cast<u32x4*>(dst) = pa[sp][0];
                *reinterpret_cast<u32x4*>(dst + 16) = pa[sp][1];
            }
        }
        unsigned char* dst = base + NS * A_BYTES + srow * CV_ROW + shalf * 32;
        if constexpr (SPLIT) {
            // 16 fp32 -> 16 bf16 hi + 16 bf16 lo
            u32x4 hi[2], lo[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float x0 = __uint_as_float(pb[i][2 * e]), x1 = __uint_as_float(pb[i][2 * e + 1]);
                    const bf16 h0 = (bf16)x0, h1 = (bf16)x1;
                    hi[i >> 1][(i & 1) * 2 + e] = pack2bf((float)h0, (float)h1);
                    lo[i >> 1][(i & 1) * 2 + e] = pack2bf(x0 - (float)h0, x1 - (float)h1);
                }
            }
            *reinterpret_cast<u32x4*>(dst) = hi[0];
            *reinterpret_cast<u32x4*>(dst + 16) = hi[1];
            *reinterpret_cast<u32x4*>(dst + B_BYTES) = lo[0];
            *reinterpret_cast<u32x4*>(dst + B_BYTES + 16) = lo[1];
        } else {
            *reinterpret_cast<u32x4*>(dst) = pb[0];
            *reinterpret_cast<u32x4*>(dst + 16) = pb[1];
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    prefetch(0);
    commit(0);
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) prefetch(ks + 1);
        const unsigned char* ab = smem + buf * BUF;
        const unsigned char* bb = ab + NS * A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[NS][TI], bfr[NS][TJ];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    af[sp][i] = *reinterpret_cast<const bf16x8*>(ab + sp * A_BYTES + ((wm * TI + i) * 32 + p) * CV_ROW + (s * 16 + 8 * h) * 2);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    bfr[sp][j] = *reinterpret_cast<const bf16x8*>(bb + sp * B_BYTES + ((wn * TJ + j) * 32 + p) * CV_ROW + (s * 16 + 8 * h) * 2);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    if constexpr (SPLIT) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (ks + 1 < KS) commit(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const long m = m0 + (wn * TJ + j) * 32 + p;
#pragma unroll
        for (int i = 0; i < TI; ++i)
            epilogue_store<TOut>(acc[i][j], a.e, m, nt0 * BN + (wm * TI + i) * 32 + 16 * h, m < a.M);
    }
}

int conv_block_couts(int cout) { return cout >= 128 ? 128 : 64; }

template <typename TIn, typename TOut>
static int launch_conv_t(const ConvArgs& a, hipStream_t st) {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    const int bn = conv_block_couts(a.e.cout);
    dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)cdiv(a.e.cout, bn));
    const size_t lds = 2 * (size_t)NS * (bn + 128) * CV_ROW;
    if (bn == 128) {
        auto k = conv_gemm_kernel<TIn, TOut, 2, 2, 2, 2>;
        if (lds > 65536) FR_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    } else {
        auto k = conv_gemm_kernel<TIn, TOut, 2, 1, 1, 4>;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    }
    FR_LAUNCH_CHECK("conv_gemm_kernel");
    return FR_OK;
}

int launch_conv(bool in_f32, bool out_f32, const ConvArgs& a, hipStream_t st) {
    if (a.Cin % 32 || a.e.cout % 8 || a.M <= 0 || (a.ksize != 1 && a.ksize != 3))
        FR_FAIL(FR_ERR_ARG, "conv: Cin must be a multiple of 32, Cout of 8, ksize 1 or 3");
    if ((a.M + 127) / 128 > 0x7FFFFFFFL) FR_FAIL(FR_ERR_ARG, "conv: grid too large");
    if (in_f32 && out_f32) return launch_conv_t<float, float>(a, st);
    if (!in_f32 && !out_f32) return launch_conv_t<bf16, bf16>(a, st);
    if (!in_f32 && out_f32) return launch_conv_t<bf16, float>(a, st);
    FR_FAIL(FR_ERR_UNSUPPORTED, "conv: fp32 input with bf16 output is not instantiated");
}
