// Pointwise (1x1) convolution / Linear as a bf16 MFMA GEMM with a fused epilogue.
//
// Replaces nn.Conv2d(k=1)+BatchNorm2d+activation (utils/layers.py:15-84), the SE scale multiply
// (utils/layers.py:122), the residual add (models/mobilenetv3.py:96-97, mobilenetv2.py:55) and
// nn.Linear+BatchNorm1d (utils/layers.py:183-186).
//
// Orientation: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k] on v_mfma_f32_32x32x16_bf16, i.e.
// the (tiny, LDS-resident, pre-packed) weights are the MFMA "A" operand and the streamed NHWC
// activations are the "B" operand, loaded straight from global memory: lane (p = lane&31,
// h = lane>>5) reads the 16 contiguous bytes X[pixel p][16s+8h .. +7].  With the row permutation
// of common.h each lane ends up with 16 consecutive output channels of one pixel, so the epilogue
// (bias, PReLU/Hardswish/ReLU/Hardsigmoid, residual) stores 16-byte NHWC vectors.
//
// The op is HBM-bound: every activation is read once and written once; weights
// are staged into LDS once per block when they fit (k-chunked otherwise).
#include "kernels.h"

template <typename TIn>
struct XFrag;
template <>
struct XFrag<bf16> {
    bf16x8 hi;
    FR_DEV void zero() { hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    FR_DEV void load(const bf16* p, const float* se) {
        if (se) {
            Vec8f x = load8(p), s = load8(se);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] *= s.v[i];
            hi = to_frag(x);
        } else {
            hi = *reinterpret_cast<const bf16x8*>(p);
        }
    }
};
template <>
struct XFrag<float> {
    bf16x8 hi, lo;
    FR_DEV void zero() { hi = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; lo = hi; }
    FR_DEV void load(const float* p, const float* se) {
        Vec8f x = load8(p);
        if (se) {
            Vec8f s = load8(se);
#pragma unroll
            for (int i = 0; i < 8; ++i) x.v[i] *= s.v[i];
        }
        split8(x, hi, lo);
    }
};

template <typename TIn, typename TOut, int NTB>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs a) {
    constexpr bool SPLIT = std::is_same<TIn, float>::value;
    constexpr int NS = SPLIT ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.y * NTB;
    const int nt = min(NTB, a.NT - t0);
    const int nchunks = (a.KS + a.chunk_ks - 1) / a.chunk_ks;
    const long n_super = (a.M + 127) / 128;
    const TIn* X = reinterpret_cast<const TIn*>(a.x);
    bool staged = false;

    for (long st = blockIdx.x; st < n_super; st += gridDim.x) {
        const long m = st * 128 + wave * 32 + p;
        const bool m_ok = m < a.M;
        const long mc = m_ok ? m : a.M - 1;
        const TIn* xrow = X + mc * (long)a.K;
        const float* serow = a.se ? a.se + (mc / a.HW) * (long)a.K : nullptr;

        f32x16 acc[NTB];
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            const int s0 = c * a.chunk_ks;
            const int ns = min(a.chunk_ks, a.KS - s0);
            if (nchunks > 1 || !staged) {
                if (nchunks > 1) __syncthreads();
                const int total = nt * ns * NS * 64;          // 16-byte units
                for (int i = threadIdx.x; i < total; i += 256) {
                    const int unit = i & 63, f = i >> 6;
                    const int sp = f % NS, ts = f / NS;
                    const int s = ts % ns, t = ts / ns;
                    const u32x4* src = reinterpret_cast<const u32x4*>(a.w) +
                        ((((long)(t0 + t) * a.KS + s0 + s) * NS + sp) * 64 + unit);
                    reinterpret_cast<u32x4*>(smem)[i] = *src;
                }
                __syncthreads();
                staged = true;
            }
            for (int sg = 0; sg < ns; sg += 4) {
                XFrag<TIn> xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = (s0 + sg + j) * 16 + 8 * h;
                    if (sg + j < ns && kk < a.K) xf[j].load(xrow + kk, serow ? serow + kk : nullptr);
                    else xf[j].zero();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sg + j < ns) {
#pragma unroll
                        for (int t = 0; t < NTB; ++t) {
                            if (t < nt) {
                                const bf16x8* wf = reinterpret_cast<const bf16x8*>(smem) +
                                    ((t * ns + sg + j) * NS) * 64 + lane;
                                bf16x8 wh = wf[0];
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[j].hi, acc[t], 0, 0, 0);
                                if constexpr (SPLIT) {
                                    bf16x8 wl = wf[64];
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xf[j].lo, acc[t], 0, 0, 0);
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xf[j].hi, acc[t], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTB; ++t)
            if (t < nt) epilogue_store<TOut>(acc[t], a.e, m, 32 * (t0 + t) + 16 * h, m_ok);
    }
}

void pw_pick_tiling(bool in_f32, int NT, int KS, int* ntb, int* chunk_ks) {
    const int ns = in_f32 ? 2 : 1;
    const int cap = 4;
    int b = 1;
    while (b < NT && b < cap) b *= 2;
    int ck = 48 / (b * ns);           // <= 48 KB of LDS
    if (ck > KS) ck = KS;
    if (ck < 1) ck = 1;
    *ntb = b;
    *chunk_ks = ck;
}

template <typename TIn, typename TOut>
static int launch_pw_t(const PwArgs& a, int ntb, hipStream_t st) {
    constexpr int NS = std::is_same<TIn, float>::value ? 2 : 1;
    const long n_super = (a.M + 127) / 128;
    dim3 grid((unsigned)std::min<long>(n_super, 2048), (unsigned)cdiv(a.NT, ntb));
    const size_t lds = (size_t)std::min(ntb, a.NT) * a.chunk_ks * NS * 1024;
    switch (ntb) {
        case 1: hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 1>), grid, dim3(256), lds, st, a); break;
        case 2: hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 2>), grid, dim3(256), lds, st, a); break;
        case 4: hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 4>), grid, dim3(256), lds, st, a); break;
        case 8: hipLaunchKernelGGL((pw_gemm_kernel<TIn, TOut, 8>), grid, dim3(256), lds, st, a); break;
        default: FR_FAIL(FR_ERR_ARG, "pw: bad n-tile blocking");
    }
    FR_LAUNCH_CHECK("pw_gemm_kernel");
    return FR_OK;
}

int launch_pw(bool in_f32, bool out_f32, const PwArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.K <= 0 || a.K % 8 || a.e.cout % 8 || a.KS != cdiv(a.K, 16) || a.NT != cdiv(a.e.cout, 32))
        FR_FAIL(FR_ERR_ARG, "pw: inconsistent shapes (K and Cout must be multiples of 8)");
    int ntb, ck;
    pw_pick_tiling(in_f32, a.NT, a.KS, &ntb, &ck);
    if (ck != a.chunk_ks) FR_FAIL(FR_ERR_ARG, "pw: chunk_ks does not match pw_pick_tiling");
    if (in_f32 && out_f32) return launch_pw_t<float, float>(a, ntb, st);
    if (!in_f32 && !out_f32) return launch_pw_t<bf16, bf16>(a, ntb, st);
    if (!in_f32 && out_f32) return launch_pw_t<bf16, float>(a, ntb, st);
    FR_FAIL(FR_ERR_UNSUPPORTED, "pw: fp32 input with bf16 output is not instantiated");
}
